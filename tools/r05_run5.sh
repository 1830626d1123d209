#!/bin/bash
# (record of a measured and NOT adopted experiment: the kernel variant / developer switch it drives was removed again; the result is under profiles/r05_*)
# round 5: attention -- static wave priority against the convoy of co-resident waves (developer build, AMX_ATTN_PRIO)
mkdir -p gpurun_out
O=gpurun_out/r05_attn_prio.log
rm -f $O
D=$PWD/build/liballophant_amx_dev.so
for mode in "0 8" "1 8" "2 8" "3 8" "1 9" "0 8"; do
  set -- $mode
  (AMX_LIB_PATH=$D AMX_ATTN_PRIO=$1 AMX_ATTN_PRIO_BIT=$2 timeout 300 python tools/geometry_sweep.py f16x3 32:10 8:60 2>&1 | grep -v amdgpu.ids | sed "s/^/prio=$1 bit=$2 : /" | sed 's/host-side.*kernels/kernels/') >> $O
done
cat $O
