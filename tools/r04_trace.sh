#!/bin/bash
# round 4: kernel traces of short-batch steps (4 x 10 s, 1 x 3 s) and bench lines of configs 4 and 5
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for G in "4 10" "1 3" "8 10"; do
  set -- $G
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$1x$2 -- python3 $ROOT/tools/step_trace.py f16x3 $1 $2 20 > $OUT/step_$1x$2.log 2> $OUT/step_$1x$2.err
  find $OUT/trace_$1x$2 -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_$1x$2.csv \;
  rm -rf $OUT/trace_$1x$2
  python3 $ROOT/tools/step_trace.py f16x3 $1 $2 40 >> $OUT/step_$1x$2.log 2>> $OUT/step_$1x$2.err
done
cd $ROOT
python3 bench.py --config 4 > $OUT/bench_config4.json 2> $OUT/bench_config4.err
python3 bench.py --config 5 --also f16 > $OUT/bench_config5.json 2> $OUT/bench_config5.err
ls -la $OUT
