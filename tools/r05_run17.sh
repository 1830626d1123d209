#!/bin/bash
# round 5: the attention kernel forms side by side again, now that no form has compiler-inserted vmcnt(0) waits in its key loop
# (tools/attn_bench.hip; f16x3, H = 16, all utterances full length)
mkdir -p gpurun_out
O=gpurun_out/r05_attention_forms.log
rm -f $O
for g in "8 2999" "12 1999" "16 999" "32 499"; do
  echo "== N T = $g" >> $O
  ATTN2_WAVES=4 timeout 120 build/attn_bench $g | tail -2 >> $O
  ATTN2_WAVES=8 timeout 120 build/attn_bench $g | tail -1 >> $O
  ATTN2_WAVES=4 AMX_ATTN2_PERSISTENT=1 timeout 120 build/attn_bench $g | tail -1 | sed 's/2 slots>/2 slots, persistent grid>/' >> $O
done
cat $O
