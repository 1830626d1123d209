#!/bin/bash
# round 5: attention outputs as whole lines through an LDS patch -- parity tests, then same-box A/B (developer build,
# AMX_ATTN_NARROW_STORES=1 = the direct 8-byte stores)
mkdir -p gpurun_out
O=gpurun_out
(timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_path.py tests/test_gpu_variant.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -4) > $O/r05_attn_store_tests.log
D=$PWD/build/liballophant_amx_dev.so
rm -f $O/r05_attn_store_ab.log
for i in 1 2; do
(AMX_LIB_PATH=$D AMX_ATTN_NARROW_STORES=1 timeout 400 python tools/geometry_sweep.py f16x3 4:10 32:10 1:10 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/8-byte stores from registers : /') >> $O/r05_attn_store_ab.log
(AMX_LIB_PATH=$D timeout 400 python tools/geometry_sweep.py f16x3 4:10 32:10 1:10 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/whole lines through LDS      : /') >> $O/r05_attn_store_ab.log
done
cat $O/r05_attn_store_tests.log; cut -c1-250 $O/r05_attn_store_ab.log
