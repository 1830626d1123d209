#!/bin/bash
# round 5: the branch-free scatter / plane epilogue on 192-column tiles -- GEMM self-check (plan + forced NI = 3), short products,
# model-level tests, same-box A/B against 256-column tiles only (developer build, AMX_NO_NARROW_TILES=1)
mkdir -p gpurun_out
O=gpurun_out
(timeout 600 build/gemm_bench check 2>&1 | grep -v "ok$" | grep -v "fp64 truth" | tail -10) > $O/r05_gemm_check2.log
(AMX_PP_FORCE_NI=3 timeout 600 build/gemm_bench check 2>&1 | grep -v "ok$" | grep -v "fp64 truth" | tail -10) > $O/r05_gemm_check2_ni3.log
(timeout 300 build/gemm_bench small 2>&1 | grep "prec=3" | grep "qkv\|ffn1") > $O/r05_gemm_small2.log
(AMX_NO_NARROW_TILES=1 timeout 300 build/gemm_bench small 2>&1 | grep "prec=3" | grep "qkv\|ffn1") > $O/r05_gemm_small2_no_narrow.log
(timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_timed_path.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4) > $O/r05_narrow_tests.log
D=$PWD/build/liballophant_amx_dev.so
rm -f $O/r05_narrow_ab.log
for i in 1 2; do
(AMX_LIB_PATH=$D AMX_NO_NARROW_TILES=1 timeout 400 python tools/geometry_sweep.py f16x3 4:10 8:10 16:10 32:10 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/256-column tiles only : /') >> $O/r05_narrow_ab.log
(AMX_LIB_PATH=$D timeout 400 python tools/geometry_sweep.py f16x3 4:10 8:10 16:10 32:10 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/192 / 256 planned      : /') >> $O/r05_narrow_ab.log
done
tail -n 40 $O/r05_gemm_check2.log $O/r05_gemm_check2_ni3.log $O/r05_gemm_small2.log $O/r05_gemm_small2_no_narrow.log $O/r05_narrow_tests.log; cut -c1-240 $O/r05_narrow_ab.log
