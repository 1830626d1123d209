#!/bin/bash
# round 6, run 20: 24 drawn large geometries (10-40 utterances of 3-24 s, ragged or equal) at XLS-R shape against the CPU oracle
mkdir -p gpurun_out
(AMX_LARGE_GEOMETRY_SEEDS=24 timeout 2400 python -m pytest tests/test_gpu_timed_path.py -q -s -k random_large 2>&1 | grep "max |log-prob\|passed\|failed\|Error\|assert" | tail -40) > gpurun_out/r06_large_geometry_sweep.log
cat gpurun_out/r06_large_geometry_sweep.log
