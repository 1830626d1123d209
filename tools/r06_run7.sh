#!/bin/bash
# round 6, run 7: per-product durations with the fold on / off (rocprofv3 kernel trace), head-dimension tests, graph tests
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_fold -- python3 $ROOT/tools/step_trace.py f16x3 32 10 10 > $O/step_fold.log 2> $O/step_fold.err
export AMX_NO_LN_FOLD=1 AMX_LIB_PATH=$ROOT/build/liballophant_amx_dev.so
rocprofv3 --kernel-trace --output-format csv -d $O/trace_nofold -- python3 $ROOT/tools/step_trace.py f16x3 32 10 10 > $O/step_nofold.log 2> $O/step_nofold.err
unset AMX_NO_LN_FOLD AMX_LIB_PATH
cd $ROOT
(echo "== fold"; python tools/r06_dispatch_summary.py $O/trace_fold; echo "== no fold"; python tools/r06_dispatch_summary.py $O/trace_nofold) > $O/r06_per_product_durations.log 2>&1
rm -rf $O/trace_fold $O/trace_nofold
cat $O/r06_per_product_durations.log
(timeout 1200 python -m pytest tests/test_gpu_head_dim.py tests/test_gpu_graph.py tests/test_gpu_range.py -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -25)
