#!/bin/bash
# round 5: attention at one wave per SIMD with the sub-blocks of a wave pipelined (tools/experiments/attn5_one_wave_per_simd.inc)
# against the library's attn2_kernel<4 waves, 2 slots>: bitwise comparison, then time per launch (tools/attn_bench.hip)
mkdir -p gpurun_out
O=gpurun_out/r05_attn5.log
rm -f $O
for g in "8 2999" "16 999"; do
  echo "== N T = $g" >> $O
  ATTN_BASE_WAVES=2 ATTN2_WAVES=${A5V:-6} timeout 120 build/attn_bench $g 2>&1 | tail -3 >> $O
done
ATTN_BASE_WAVES=2 ATTN2_WAVES=${A5V:-6} timeout 120 build/attn_bench_stamp 8 2999 2>&1 | grep "cycles per 64-key" >> $O
cat $O
