#!/bin/bash
# round 6, run 1: the LayerNorm fold -- GPU suite on the product library, then same-box step times: round-5 library / this tree /
# this tree with the fold switched off (developer build)
mkdir -p gpurun_out
O=gpurun_out
(timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -15) > $O/r06_run1_suite.log
G="32:10 8:60 16:10 4:10"
rm -f $O/r06_ln_fold_ab.log
for rep in 1 2; do
(AMX_ABI_OVERRIDE=5 AMX_LIB_PATH=$PWD/build/ab/r05.so timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/round-5 lib : /') >> $O/r06_ln_fold_ab.log
(timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/fold        : /') >> $O/r06_ln_fold_ab.log
(AMX_NO_LN_FOLD=1 AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/fold off    : /') >> $O/r06_ln_fold_ab.log
done
cat $O/r06_run1_suite.log; cut -c1-420 $O/r06_ln_fold_ab.log
