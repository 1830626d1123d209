#!/bin/bash
# round 5, final tree: wide randomised parity sweep against the oracle (300 random small models / batches) and the rocprofv3 + PMC
# passes of config 5 (8 x 60 s: attn2_kernel's matrix-pipe occupancy after the late attention changes)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out
(AMX_RANDOM_SEEDS=300 timeout 1500 python3 -m pytest tests/test_gpu_parity.py -q -k random_models 2>&1 | grep -E "passed|failed|error" | tail -3) > gpurun_out/r05_random_sweep_300.log
bash tools/profile_bench.sh r05_c5 --config 5 > gpurun_out/r05_profile_c5.log 2>&1
cat gpurun_out/r05_random_sweep_300.log; head -8 gpurun_out/prof_r05_c5/kernel_stats.csv | cut -c1-160; grep -i "attn2\|gemm_pp_kernelIDF16_Li2ELi8ELi4" gpurun_out/prof_r05_c5/pmc_MFMA_summary.txt | head -6
