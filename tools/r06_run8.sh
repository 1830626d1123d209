#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_head_dim.py tests/test_gpu_graph.py -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl") > $O/r06_run8_full.log
grep -n "^E  \|^FAILED\|passed\|failed" $O/r06_run8_full.log | cut -c1-250 | head -60
