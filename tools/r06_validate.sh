#!/bin/bash
# round 6, final tree: wide randomised parity sweep against the oracle (300 random small models / batches; every third one with a head
# dimension other than 64, half of those with an adapter) and the rocprofv3 + PMC passes of config 5 (8 x 60 s: matrix-pipe
# occupancy of attn2_kernel and of the fold's 128-row producer instances)
mkdir -p gpurun_out
(AMX_RANDOM_SEEDS=300 timeout 1800 python3 -m pytest tests/test_gpu_parity.py -q -k random_models 2>&1 | grep -E "passed|failed|error|Error" | tail -5) > gpurun_out/r06_random_sweep_300.log
bash tools/profile_bench.sh r06_c5 --config 5 > gpurun_out/r06_profile_c5.log 2>&1
cat gpurun_out/r06_random_sweep_300.log; head -8 gpurun_out/prof_r06_c5/kernel_stats.csv | cut -c1-160; grep -i "attn2\|gemm_pp_kernelIDF16_Li2ELi[48]ELi4" gpurun_out/prof_r06_c5/pmc_MFMA_summary.txt | head -8
