#!/bin/bash
# round 5: rocprofv3 kernel stats of short and long steps (4 x 10 s = config 3's per-GPU share, 8 x 60 s = config 5's geometry); the
# steps replay HIP graphs (the default): the trace shows whether the profiler sees the kernels of a replayed graph
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for G in "4 10" "8 60"; do
  set -- $G
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$1x$2 -- python3 $ROOT/tools/step_trace.py f16x3 $1 $2 20 > $OUT/step_$1x$2.log 2> $OUT/step_$1x$2.err
  find $OUT/trace_$1x$2 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$1x$2.csv \;
  rm -rf $OUT/trace_$1x$2
done
head -12 $OUT/kernel_stats_4x10.csv | cut -c1-170; head -8 $OUT/kernel_stats_8x60.csv | cut -c1-170; cat $OUT/step_4x10.log $OUT/step_8x60.log | grep -v amdgpu
