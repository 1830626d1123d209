#!/bin/bash
# round 6, run 16: the wide attention instances that stop at the real head columns (dh = 80 / 96 in 128-wide rows): head-dim tests,
# (developer build: the A/B switch only exists there) bitwise against the whole-row form, XLS-R 1B step with and without
mkdir -p gpurun_out
O=gpurun_out
rm -f $O/r06_attention_real_columns_ab.log
(AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so AB_ENCODER=xlsr-1b timeout 600 python tools/ab_bitwise.py f16x3 8:10 2:30 2>&1 | grep -v amdgpu.ids) > $O/r06_run16_bits_real.txt
(AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so AB_ENCODER=xlsr-1b AMX_ATTN_WHOLE_ROW=1 timeout 600 python tools/ab_bitwise.py f16x3 8:10 2:30 2>&1 | grep -v amdgpu.ids) > $O/r06_run16_bits_whole.txt
(diff $O/r06_run16_bits_real.txt $O/r06_run16_bits_whole.txt && echo "xlsr-1b, real columns against whole rows: bitwise equal ($(wc -l < $O/r06_run16_bits_real.txt) digests)") > $O/r06_attention_real_columns_bitwise.log 2>&1
for i in 1 2; do
(AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 600 python bench.py --encoder xlsr-1b --also "" --no-cpu-baseline --no-ragged 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('real columns ', round(d['ms_per_step'],3), 'ms', d['kernels']['attention'], d['parity_spot_check'].get('max_abs'))") >> $O/r06_attention_real_columns_ab.log
(AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so AMX_ATTN_WHOLE_ROW=1 timeout 600 python bench.py --encoder xlsr-1b --also "" --no-cpu-baseline --no-ragged 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('whole rows   ', round(d['ms_per_step'],3), 'ms', d['kernels']['attention'], d['parity_spot_check'].get('max_abs'))") >> $O/r06_attention_real_columns_ab.log
done
cat $O/r06_attention_real_columns_bitwise.log $O/r06_attention_real_columns_ab.log
