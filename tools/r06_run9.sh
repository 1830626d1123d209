#!/bin/bash
# round 6, run 9: the stream in planes (producers read their residual from the planes, fp32 rows only where read) -- parity first
# (XLS-R-shape oracle tests, range families, head dimensions, graphs), then step times against the fp32-stream form and no fold
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
(timeout 1500 python -m pytest tests/test_gpu_head_dim.py tests/test_gpu_graph.py tests/test_gpu_range.py tests/test_gpu_timed_path.py tests/test_gpu_parity.py -q -m gpu -x 2>&1 | grep -v "version\|Hostname\|Librccl") > $O/r06_run9_tests.log
grep -n "^E  \|^FAILED\|passed\|failed" $O/r06_run9_tests.log | cut -c1-250 | head -40
rm -f $O/r06_stream_in_planes_ab.log
export AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so
for rep in 1 2; do
for V in "" "AMX_FOLD_F32_STREAM=1" "AMX_NO_LN_FOLD=1"; do
(env $V timeout 400 python tools/geometry_sweep.py f16x3 32:10 8:60 16:10 2>&1 | grep -v amdgpu.ids | sed "s/^/${V:-stream in planes} : /") >> $O/r06_stream_in_planes_ab.log
done
done
unset AMX_LIB_PATH
python - <<'PY'
import re
for l in open('gpurun_out/r06_stream_in_planes_ab.log'):
    m=re.match(r"(.*?) : f16x3 (\d+ x \d+) s:\s+([\d.]+) ms/step.*kernels\s+([\d.]+) ms.*?gemm_pp=([\d.]+).*?attention=([\d.]+) rownorm=([\d.]+).*?gemm_ln=([\d.]+)",l)
    if m: print(f"{m.group(1):28s} {m.group(2):8s} step {m.group(3):>7s}  gemm_pp {m.group(5):>6s} rownorm {m.group(7)}")
    elif "Error" in l: print(l[:200])
PY
