#!/bin/bash
# round 6, run 11: the fold test in every mode, the corpus through the batcher (plain and on a grid of geometries: replay share),
# per-product durations of the 32 x 10 s step (fold on / off)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_timed_path.py -q -m gpu -k "layer_norm_fold" 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -8) > $O/r06_run11_tests.log
cat $O/r06_run11_tests.log
(timeout 900 python tools/corpus_throughput.py f16x3 4096 2>&1 | grep -v amdgpu.ids) > $O/r06_corpus_throughput.log
cat $O/r06_corpus_throughput.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_fold -- python3 $ROOT/tools/step_trace.py f16x3 32 10 10 > $O/step_fold.log 2> $O/step_fold.err
export AMX_NO_LN_FOLD=1 AMX_LIB_PATH=$ROOT/build/liballophant_amx_dev.so
rocprofv3 --kernel-trace --output-format csv -d $O/trace_nofold -- python3 $ROOT/tools/step_trace.py f16x3 32 10 10 > $O/step_nofold.log 2> $O/step_nofold.err
unset AMX_NO_LN_FOLD AMX_LIB_PATH
cd $ROOT
(echo "== fold (stream in planes)"; python tools/r06_dispatch_summary.py $O/trace_fold; echo "== no fold"; python tools/r06_dispatch_summary.py $O/trace_nofold) > $O/r06_per_product_durations.log 2>&1
rm -rf $O/trace_fold $O/trace_nofold
cat $O/r06_per_product_durations.log
