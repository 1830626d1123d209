#!/bin/bash
# round 6, run 22: rocprofv3 kernel stats of the XLS-R 1B / 2B steps (32 x 10 s) -- which instances the wide shapes run and what each costs
mkdir -p gpurun_out
ROOT=$PWD
O=gpurun_out
cd /tmp && export TMPDIR=/tmp
for V in xlsr-1b xlsr-2b; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/trace_$V -- python3 $ROOT/tools/step_trace.py f16x3 32 10 10 $V > $ROOT/$O/step_$V.log 2> $ROOT/$O/step_$V.err
  find $ROOT/$O/trace_$V -name "*kernel_stats.csv" -exec cp {} $ROOT/$O/r06_kernel_stats_$V.csv \;
  rm -rf $ROOT/$O/trace_$V
  tail -2 $ROOT/$O/step_$V.log; head -9 $ROOT/$O/r06_kernel_stats_$V.csv | cut -c1-150
done
