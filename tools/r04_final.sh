#!/bin/bash
# round 4, final tree: bench lines of configs 2 / 4 / 5 and the rocprofv3 + PMC passes of both precision modes
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out/r04
python3 bench.py > gpurun_out/r04/bench_line.json 2> gpurun_out/r04/bench_line.err
python3 bench.py --config 4 > gpurun_out/r04/bench_config4.json 2> gpurun_out/r04/bench_config4.err
python3 bench.py --config 5 --also f16 > gpurun_out/r04/bench_config5.json 2> gpurun_out/r04/bench_config5.err
tools/profile_bench.sh r04_f16x3 --precision f16x3 > gpurun_out/r04/profile_f16x3.log 2>&1
tools/profile_bench.sh r04_bf16 --precision bf16 > gpurun_out/r04/profile_bf16.log 2>&1
# per-kernel stats of configs 4 and 5 (kernel trace only)
cd /tmp && export TMPDIR=/tmp
for C in 4 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r04/trace_c$C -- python3 $ROOT/bench.py --config $C --steps 5 --warmup 2 --no-cpu-baseline --no-ragged --also "" > $ROOT/gpurun_out/r04/bench_config${C}_traced.json 2> $ROOT/gpurun_out/r04/trace_c$C.err
  find $ROOT/gpurun_out/r04/trace_c$C -name "*kernel_stats.csv" -exec cp {} $ROOT/gpurun_out/r04/kernel_stats_config$C.csv \;
  rm -rf $ROOT/gpurun_out/r04/trace_c$C
done
cd $ROOT
python3 tools/geometry_sweep.py f16x3 1:3 1:10 4:10 8:10 16:10 32:10 1:60 > gpurun_out/r04/geometry_sweep.log 2>&1
ls -la gpurun_out/r04 | tail -30
