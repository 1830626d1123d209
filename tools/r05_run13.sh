#!/bin/bash
# round 5: attn2 output stores widened to 16 bytes with v_permlane32_swap -- long-utterance tests, then same-box A/B against HEAD's library
mkdir -p gpurun_out
O=gpurun_out
(timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_path.py tests/test_gpu_variant.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -4) > $O/r05_attn2_swap_tests.log
rm -f $O/r05_attn2_swap_ab.log
for i in 1 2 3; do
(AMX_LIB_PATH=$PWD/build/ab/head.so timeout 400 python tools/geometry_sweep.py f16x3 8:60 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/8-byte stores            : /') >> $O/r05_attn2_swap_ab.log
(AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 400 python tools/geometry_sweep.py f16x3 8:60 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/16-byte stores (lane swap): /') >> $O/r05_attn2_swap_ab.log
done
cat $O/r05_attn2_swap_tests.log; cut -c1-260 $O/r05_attn2_swap_ab.log
