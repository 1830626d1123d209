#!/bin/bash
# round 6, run 2: the fold's kernels as instances of their own (the plain kernels keep their round-5 code): A/B again, then the new tests
mkdir -p gpurun_out
O=gpurun_out
G="32:10 8:60 16:10 4:10"
rm -f $O/r06_ln_fold_ab2.log
for rep in 1 2; do
(AMX_ABI_OVERRIDE=5 AMX_LIB_PATH=$PWD/build/ab/r05.so timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/round-5 lib : /') >> $O/r06_ln_fold_ab2.log
(timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/fold        : /') >> $O/r06_ln_fold_ab2.log
(AMX_NO_LN_FOLD=1 AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/fold off    : /') >> $O/r06_ln_fold_ab2.log
done
cut -c1-30,300-420 $O/r06_ln_fold_ab2.log
(timeout 1500 python -m pytest tests/test_c_host.py tests/test_gpu_long.py tests/test_gpu_range.py tests/test_gpu_timed_path.py -x -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -15) > $O/r06_run2_tests.log
cat $O/r06_run2_tests.log
