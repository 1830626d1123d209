#!/bin/bash
# (record of a measured and NOT adopted experiment: the kernel variant / developer switch it drives was removed again; the result is under profiles/r05_*)
# round 5: attention at 60 s utterances (attn2_kernel) -- static wave priority against the convoy of the two co-resident workgroups;
# plus the range / graph tests on the tree with the carried range count
mkdir -p gpurun_out
O=gpurun_out/r05_attn_prio_long.log
rm -f $O
D=$PWD/build/liballophant_amx_dev.so
for mode in "0 8" "1 8" "1 9" "1 7" "0 8" "1 8"; do
  set -- $mode
  (AMX_LIB_PATH=$D AMX_ATTN_PRIO=$1 AMX_ATTN_PRIO_BIT=$2 timeout 300 python tools/geometry_sweep.py f16x3 8:60 2>&1 | grep -v amdgpu.ids | sed "s/^/prio=$1 bit=$2 : /" | sed 's/host-side.*kernels/kernels/') >> $O
done
cut -c1-260 $O
(timeout 900 python -m pytest tests/test_gpu_range.py tests/test_gpu_graph.py -x -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -3)
