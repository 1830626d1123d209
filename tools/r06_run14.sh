#!/bin/bash
# round 6, run 14: hidden up to 2048 -- golden g14, the XLS-R 1B / 2B layer shapes, and the suites the row kernels sit under
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
(timeout 2400 python -m pytest tests/test_gpu_head_dim.py tests/test_gpu_variant.py tests/test_gpu_range.py tests/test_gpu_parity.py -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl") > $O/r06_run14_tests.log
grep -n "^E  \|^FAILED\|passed\|failed" $O/r06_run14_tests.log | cut -c1-250 | head -40
