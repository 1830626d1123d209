#!/bin/bash
# round 6, run 5: LDS waits of the GEMM K loops through the builtin (the compiler no longer re-waits per fragment between the MFMAs) --
# round-5 library / this tree with the LayerNorm fold / this tree without it, same box; then the fold's per-kernel durations
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
G="32:10 8:60 16:10 4:10"
rm -f $O/r06_ln_fold_ab4.log
for rep in 1 2; do
(AMX_ABI_OVERRIDE=5 AMX_LIB_PATH=$PWD/build/ab/r05.so timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/round-5 lib : /') >> $O/r06_ln_fold_ab4.log
(timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/fold        : /') >> $O/r06_ln_fold_ab4.log
(AMX_NO_LN_FOLD=1 AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/fold off    : /') >> $O/r06_ln_fold_ab4.log
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fold -- python3 $ROOT/tools/step_trace.py f16x3 32 10 10 > $O/step_fold.log 2> $O/step_fold.err
find $O/trace_fold -name "*kernel_stats.csv" -exec cp {} $O/r06_kernel_stats_32x10_fold3.csv \;
rm -rf $O/trace_fold
cd $ROOT
python - <<'PY'
import re,csv
for l in open('gpurun_out/r06_ln_fold_ab4.log'):
    m=re.match(r"(\S.*?): f16x3 (\d+ x \d+) s:\s+([\d.]+) ms/step.*kernels\s+([\d.]+) ms.*?gemm_pp=([\d.]+).*?attention=([\d.]+) rownorm=([\d.]+).*?gemm_ln=([\d.]+)",l)
    if m: print(f"{m.group(1):12s} {m.group(2):8s} step {m.group(3):>7s}  gemm_pp {m.group(5):>6s} attn {m.group(6)} rownorm {m.group(7)} gemm_ln {m.group(8)}")
rows=list(csv.DictReader(open("gpurun_out/r06_kernel_stats_32x10_fold3.csv")))
for r in rows[:9]:
    print(r["Name"][:90], r["Calls"], round(int(r["TotalDurationNs"])/13e6,3), "ms/step avg", round(float(r["AverageNs"])/1e3,1))
PY
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or xlsr or oracle" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -5)
