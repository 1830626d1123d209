#!/bin/bash
# round 6, run 4: consumer coefficients fetched a round ahead -- step times fold on / off and the fold's per-kernel durations
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
G="32:10 8:60 16:10"
rm -f $O/r06_ln_fold_ab3.log
for rep in 1 2; do
(timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/fold        : /') >> $O/r06_ln_fold_ab3.log
(AMX_NO_LN_FOLD=1 AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/fold off    : /') >> $O/r06_ln_fold_ab3.log
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fold -- python3 $ROOT/tools/step_trace.py f16x3 32 10 10 > $O/step_fold.log 2> $O/step_fold.err
find $O/trace_fold -name "*kernel_stats.csv" -exec cp {} $O/r06_kernel_stats_32x10_fold2.csv \;
rm -rf $O/trace_fold
cd $ROOT
python - <<'PY'
import re,csv
for l in open('gpurun_out/r06_ln_fold_ab3.log'):
    m=re.match(r"(\S.*?): f16x3 (\d+ x \d+) s:\s+([\d.]+) ms/step.*kernels\s+([\d.]+) ms.*?gemm_pp=([\d.]+).*?attention=([\d.]+) rownorm=([\d.]+)",l)
    if m: print(f"{m.group(1):12s} {m.group(2):8s} step {m.group(3):>7s}  gemm_pp {m.group(5):>6s} attn {m.group(6)} rownorm {m.group(7)}")
rows=list(csv.DictReader(open("gpurun_out/r06_kernel_stats_32x10_fold2.csv")))
for r in rows[:9]:
    print(r["Name"][:90], r["Calls"], round(int(r["TotalDurationNs"])/13e6,3), "ms/step avg", round(float(r["AverageNs"])/1e3,1))
PY
