#!/bin/bash
mkdir -p gpurun_out
O=gpurun_out
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5) > $O/r05_gpu_suite.log
(timeout 300 build/gemm_bench small 2>&1 | grep "prec=3" | grep "qkv") > $O/r05_gemm_small2.log
(AMX_NO_NARROW_TILES=1 timeout 300 build/gemm_bench small 2>&1 | grep "prec=3" | grep "qkv") > $O/r05_gemm_small2_no_narrow.log
D=$PWD/build/liballophant_amx_dev.so
rm -f $O/r05_narrow_ab.log
for i in 1 2; do
(AMX_LIB_PATH=$D AMX_NO_NARROW_TILES=1 timeout 400 python tools/geometry_sweep.py f16x3 4:10 8:10 16:10 32:10 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/256-column tiles only : /') >> $O/r05_narrow_ab.log
(AMX_LIB_PATH=$D timeout 400 python tools/geometry_sweep.py f16x3 4:10 8:10 16:10 32:10 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/192 or 256, planned   : /') >> $O/r05_narrow_ab.log
done
tail -5 $O/r05_gpu_suite.log; cat $O/r05_gemm_small2.log $O/r05_gemm_small2_no_narrow.log; cut -c1-250 $O/r05_narrow_ab.log
