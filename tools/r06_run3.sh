#!/bin/bash
# round 6, run 3: (1) recordings keyed on geometry (tests/test_gpu_graph.py), (2) which of round 5's two changes removed the replay fault
# (tools/r06_graph_fault.sh), (3) rocprofv3 per-kernel durations of the 32 x 10 s step with the LayerNorm fold on and off
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_graph.py -x -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -15) > $O/r06_run3_graph_tests.log
cat $O/r06_run3_graph_tests.log
bash tools/r06_graph_fault.sh
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fold -- python3 $ROOT/tools/step_trace.py f16x3 32 10 10 > $O/step_fold.log 2> $O/step_fold.err
find $O/trace_fold -name "*kernel_stats.csv" -exec cp {} $O/r06_kernel_stats_32x10_fold.csv \;
rm -rf $O/trace_fold
export AMX_NO_LN_FOLD=1 AMX_LIB_PATH=$ROOT/build/liballophant_amx_dev.so
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_nofold -- python3 $ROOT/tools/step_trace.py f16x3 32 10 10 > $O/step_nofold.log 2> $O/step_nofold.err
find $O/trace_nofold -name "*kernel_stats.csv" -exec cp {} $O/r06_kernel_stats_32x10_nofold.csv \;
rm -rf $O/trace_nofold
cut -c1-150 $O/r06_kernel_stats_32x10_fold.csv | head -14; cut -c1-150 $O/r06_kernel_stats_32x10_nofold.csv | head -12
grep -v amdgpu $O/step_fold.log $O/step_nofold.log
