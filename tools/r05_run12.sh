#!/bin/bash
# (record of a measured and NOT adopted experiment: the kernel variant / developer switch it drives was removed again; the result is under profiles/r05_*)
# round 5: attention Q rows coalesced through an LDS patch -- tests, then same-box A/B against the previous commit's library
mkdir -p gpurun_out
O=gpurun_out
(timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_path.py tests/test_gpu_variant.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -4) > $O/r05_attn_q_tests.log
rm -f $O/r05_attn_q_ab.log
for i in 1 2; do
(AMX_LIB_PATH=$PWD/build/ab/pre_attn2.so timeout 400 python tools/geometry_sweep.py f16x3 4:10 32:10 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/Q fragments straight from HBM : /') >> $O/r05_attn_q_ab.log
(AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 400 python tools/geometry_sweep.py f16x3 4:10 32:10 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/Q rows coalesced through LDS  : /') >> $O/r05_attn_q_ab.log
done
cat $O/r05_attn_q_tests.log; cut -c1-250 $O/r05_attn_q_ab.log
