#!/bin/bash
# Developer profile: where do the waves of each kernel spend their cycles?  SQ_WAIT_ANY = parked at s_waitcnt / s_barrier,
# SQ_WAIT_INST_ANY = issue stalls, SQ_ACTIVE_INST_ANY = issuing; all in quad-cycles like SQ_WAVE_CYCLES
# (MI355X_MICROARCH.md, rocprofv3 PMC slots).  Usage: tools/pmc_waits.sh <tag> [bench.py args...]
set -u
TAG=${1:-run}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/waits_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d "$OUT/pmc" -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --also "" "$@" > "$OUT/bench.json" 2> "$OUT/pmc.err"
python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc" "$OUT/summary.json" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*kernel_trace.csv" -delete
find "$OUT" -name "*counter_collection.csv" -delete
find "$OUT" -name "*.db" -delete
python3 - "$OUT/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, c in sorted(d.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", {}).get("sum", 0))[:8]:
    w = c.get("SQ_WAVE_CYCLES", {}).get("sum", 0)
    if not w:
        continue
    f = lambda n: c.get(n, {}).get("sum", 0) / w
    print(f"{k[-60:]:60s} parked {f('SQ_WAIT_ANY'):.2f}  issue-stall {f('SQ_WAIT_INST_ANY'):.2f}  issuing {f('SQ_ACTIVE_INST_ANY'):.2f}  "
          f"(valu {f('SQ_ACTIVE_INST_VALU'):.2f} lds {f('SQ_ACTIVE_INST_LDS'):.2f})  valu insts/wave {c.get('SQ_INSTS_VALU', {}).get('sum', 0) / max(1, c.get('SQ_WAVES', {}).get('sum', 1)):.0f}")
PY
