#!/bin/bash
# (record of a measured and NOT adopted experiment: the kernel variant / developer switch it drives was removed again; the result is under profiles/r05_*)
# round 5: attn2_kernel (60 s utterances) outputs as whole lines -- tests, then same-box A/B against the previous commit's library
mkdir -p gpurun_out
O=gpurun_out
(timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_timed_path.py tests/test_gpu_variant.py -x -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -4) > $O/r05_attn2_store_tests.log
rm -f $O/r05_attn2_store_ab.log
for i in 1 2; do
(AMX_LIB_PATH=$PWD/build/ab/pre_attn2.so timeout 400 python tools/geometry_sweep.py f16x3 8:60 2:30 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/8-byte stores from registers : /') >> $O/r05_attn2_store_ab.log
(AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 400 python tools/geometry_sweep.py f16x3 8:60 2:30 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/whole lines through LDS      : /') >> $O/r05_attn2_store_ab.log
done
(STRESS_N=4 STRESS_SECONDS=60 STRESS_ITERS=12 timeout 600 python tools/stress_repro.py 2>&1 | grep -v amdgpu.ids | tail -2) > $O/r05_attn2_race.log
(STRESS_N=4 STRESS_SECONDS=60 STRESS_ITERS=12 STRESS_PACKED=1 timeout 600 python tools/stress_repro.py 2>&1 | grep -v amdgpu.ids | tail -2) >> $O/r05_attn2_race.log
cat $O/r05_attn2_store_tests.log $O/r05_attn2_race.log; cut -c1-250 $O/r05_attn2_store_ab.log
