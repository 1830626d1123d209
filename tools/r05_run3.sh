#!/bin/bash
# round 5, third box: conv layer 0 on the matrix pipe -- its tests, the parity tests that gate the conv extractor, and the
# same-box A/B against the VALU kernel (developer build: AMX_NO_CONV0_MFMA=1)
mkdir -p gpurun_out
O=gpurun_out
(timeout 900 python -m pytest tests/test_gpu_conv0.py -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25) > $O/r05_conv0_tests.log
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_path.py tests/test_gpu_range.py -x -q -m gpu 2>&1 | tail -8) > $O/r05_conv0_parity.log
D=$PWD/build/liballophant_amx_dev.so
for i in 1 2; do
(AMX_LIB_PATH=$D AMX_NO_CONV0_MFMA=1 timeout 300 python tools/geometry_sweep.py f16x3 32:10 4:10 2>&1 | grep -v amdgpu.ids | sed 's/^/VALU kernel : /') >> $O/r05_conv0_ab.log
(AMX_LIB_PATH=$D timeout 300 python tools/geometry_sweep.py f16x3 32:10 4:10 2>&1 | grep -v amdgpu.ids | sed 's/^/MFMA kernel : /') >> $O/r05_conv0_ab.log
done
tail -n 30 $O/r05_conv0_tests.log $O/r05_conv0_parity.log $O/r05_conv0_ab.log
