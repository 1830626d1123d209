#!/bin/bash
# round 6, run 18: the GPU suite on the final tree after golden g15 (add_adapter) joined the tiny goldens
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -8) > gpurun_out/r06_gpu_suite.log
cat gpurun_out/r06_gpu_suite.log
