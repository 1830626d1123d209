#!/bin/bash
# round 4: the attention kernels side by side (tools/attn_bench.hip; f16x3, H = 16, all utterances full length)
#   attn_kernel (32 queries per wave, 4 waves per SIMD) against attn2_kernel (64 queries per wave) in its forms, and the two
#   overlap experiments (tools/experiments/attn3_pingpong.inc, attn4_subblock_pipeline.inc)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for g in "32 499" "8 2999" "64 249" "16 999" "8 1499" "12 1999" "16 749" "1 2999"; do
  echo "== N T = $g"
  ATTN2_WAVES=4 timeout 120 build/attn_bench $g | tail -3
  ATTN2_WAVES=4 AMX_ATTN2_PERSISTENT=1 timeout 120 build/attn_bench $g | tail -1 | sed 's/2 slots>/2 slots, persistent grid>/'
  ATTN2_WAVES=8 timeout 120 build/attn_bench $g | tail -1
done
for g in "32 499" "8 2999"; do
  echo "== experiments, N T = $g"
  ATTN2_WAVES=5 timeout 120 build/attn_bench $g | tail -1
  ATTN2_WAVES=3 timeout 120 build/attn_bench $g | tail -1
done
echo "== anatomy (stamped build: cycles include ~40 per stamp)"
for w in 8 4; do
  for g in "32 499" "8 2999"; do ATTN2_WAVES=$w timeout 120 build/attn_bench_stamp $g | grep -A3 anatomy; done
done
ATTN2_WAVES=5 timeout 120 build/attn_bench_stamp 8 2999 | grep "attn3 group"
