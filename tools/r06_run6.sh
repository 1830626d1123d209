#!/bin/bash
# round 6, run 6: where the fold's 1.1 ms of extra GEMM time sits -- developer library, producers / consumers run as plain products
# one side at a time (timing only: the outputs of those runs are wrong); then the head-dimension tests and the new graph tests
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
rm -f $O/r06_fold_sides.log
export AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so
for rep in 1 2; do
for V in "" "AMX_FOLD_CONSUMER_PLAIN=1" "AMX_FOLD_PRODUCER_PLAIN=1" "AMX_FOLD_CONSUMER_PLAIN=1 AMX_FOLD_PRODUCER_PLAIN=1" "AMX_NO_LN_FOLD=1"; do
(env $V timeout 300 python tools/geometry_sweep.py f16x3 32:10 2>&1 | grep -v amdgpu.ids | sed "s/^/${V:-fold as built} : /") >> $O/r06_fold_sides.log
done
done
unset AMX_LIB_PATH
python - <<'PY'
import re
for l in open('gpurun_out/r06_fold_sides.log'):
    m=re.match(r"(.*?) : f16x3 (\d+ x \d+) s:\s+([\d.]+) ms/step.*kernels\s+([\d.]+) ms.*?gemm_pp=([\d.]+).*?attention=([\d.]+) rownorm=([\d.]+).*?gemm_ln=([\d.]+)",l)
    if m: print(f"{m.group(1):60s} step {m.group(3):>7s}  gemm_pp {m.group(5):>6s} rownorm {m.group(7)}")
    elif "Error" in l or "error" in l: print(l[:200])
PY
(timeout 1200 python -m pytest tests/test_gpu_head_dim.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -25)
