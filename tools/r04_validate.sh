#!/bin/bash
# round 4, final tree: wide randomised sweep, race screens, config-5 PMC pass
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out/r04
AMX_RANDOM_SEEDS=300 python3 -m pytest tests/test_gpu_parity.py -q -k random_models 2>&1 | grep -E "passed|failed" > gpurun_out/r04/random_sweep_300.log
for G in "1 3 40" "4 10 40" "32 10 20" "8 60 12" "2 25 40"; do
  set -- $G
  STRESS_N=$1 STRESS_SECONDS=$2 STRESS_ITERS=$3 python3 tools/stress_repro.py >> gpurun_out/r04/race_screen.log 2>&1; STRESS_PACKED=1 STRESS_N=$1 STRESS_SECONDS=$2 STRESS_ITERS=$3 python3 tools/stress_repro.py >> gpurun_out/r04/race_screen.log 2>&1
done
tools/profile_bench.sh r04_c5 --config 5 > gpurun_out/r04/profile_c5.log 2>&1
tail -3 gpurun_out/r04/random_sweep_300.log gpurun_out/r04/race_screen.log
