#!/bin/bash
# (record of a measured and NOT adopted experiment: the kernel variant / developer switch it drives was removed again; the result is under profiles/r05_*)
# round 5, fourth box: LayerNorm rows with whole-line stores (DPP pair exchange) -- parity tests and same-box A/B
mkdir -p gpurun_out
O=gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_variant.py -x -q -m gpu 2>&1 | tail -6) > $O/r05_rownorm_parity.log
D=$PWD/build/liballophant_amx_dev.so
rm -f $O/r05_rownorm_ab.log
for i in 1 2; do
(AMX_LIB_PATH=$D AMX_ROWNORM_HALF_LINES=1 timeout 300 python tools/geometry_sweep.py f16x3 32:10 4:10 2>&1 | grep -v amdgpu.ids | sed 's/^/half-line stores : /') >> $O/r05_rownorm_ab.log
(AMX_LIB_PATH=$D timeout 300 python tools/geometry_sweep.py f16x3 32:10 4:10 2>&1 | grep -v amdgpu.ids | sed 's/^/whole-line stores: /') >> $O/r05_rownorm_ab.log
done
tail -n 30 $O/r05_rownorm_parity.log $O/r05_rownorm_ab.log
