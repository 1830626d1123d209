#!/bin/bash
# round 5, first box: the new GPU tests, the GEMM self-check incl. the 192-column tiles, the short-product timings, and the
# geometry sweep of this tree (graph replay vs eager) against the round-4 library on the same box
mkdir -p gpurun_out
O=gpurun_out
(timeout 1200 python -m pytest tests/test_gpu_graph.py tests/test_gpu_range.py tests/test_gpu_dist_fake.py -x -q -m gpu 2>&1 | tail -30) > $O/r05_new_tests.log
(timeout 600 build/gemm_bench check 2>&1 | grep -v "ok$" | tail -40) > $O/r05_gemm_check.log
(AMX_PP_FORCE_NI=3 timeout 600 build/gemm_bench check 2>&1 | grep -v "ok$" | tail -40) > $O/r05_gemm_check_ni3.log
(timeout 300 build/gemm_bench small 2>&1 | grep "prec=3") > $O/r05_gemm_small.log
(AMX_NO_NARROW_TILES=1 timeout 300 build/gemm_bench small 2>&1 | grep "prec=3") > $O/r05_gemm_small_no_narrow.log
G="1:3 4:10 8:10 16:10 32:10"
(timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids) > $O/r05_geometry_sweep.log
(AMX_ABI_OVERRIDE=4 AMX_LIB_PATH=$PWD/build/ab/r04.so timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids) > $O/r05_geometry_sweep_r04lib.log
(timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids) > $O/r05_geometry_sweep_again.log
tail -n 40 $O/r05_new_tests.log $O/r05_gemm_check.log $O/r05_gemm_check_ni3.log $O/r05_gemm_small.log $O/r05_gemm_small_no_narrow.log $O/r05_geometry_sweep.log $O/r05_geometry_sweep_r04lib.log $O/r05_geometry_sweep_again.log
