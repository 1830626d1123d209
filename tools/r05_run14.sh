#!/bin/bash
# round 5: attention softmax with fewer vector instructions (lane-swap max / sum instead of ds_bpermute, v_fma_mix for the low
# plane of P, single-instruction max) -- bitwise against HEAD's library, attention tests, same-box A/B
mkdir -p gpurun_out
O=gpurun_out
for P in f16x3 bf16x3 f16; do
AMX_LIB_PATH=$PWD/build/ab/head.so timeout 600 python tools/ab_bitwise.py $P 1:3 4:10 32:10 8:60 2>/dev/null > $O/bits_head_$P.txt
AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 600 python tools/ab_bitwise.py $P 1:3 4:10 32:10 8:60 2>/dev/null > $O/bits_new_$P.txt
done
(for P in f16x3 bf16x3 f16; do cat $O/bits_new_$P.txt; if diff -q $O/bits_head_$P.txt $O/bits_new_$P.txt > /dev/null; then echo "$P: bitwise equal to the previous library on every geometry"; else echo "$P: DIFFERS"; diff $O/bits_head_$P.txt $O/bits_new_$P.txt; fi; done) > $O/r05_attn_valu_bitwise.log
rm -f $O/r05_attn_valu_ab.log
for i in 1 2 3; do
(AMX_LIB_PATH=$PWD/build/ab/head.so timeout 400 python tools/geometry_sweep.py f16x3 4:10 32:10 8:60 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/before: /') >> $O/r05_attn_valu_ab.log
(AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 400 python tools/geometry_sweep.py f16x3 4:10 32:10 8:60 2>&1 | grep -v amdgpu.ids | sed 's/host-side.*kernels/kernels/' | sed 's/^/after : /') >> $O/r05_attn_valu_ab.log
done
cat $O/r05_attn_valu_bitwise.log; cut -c1-250 $O/r05_attn_valu_ab.log
