#!/bin/bash
# round 6: everything that describes the final tree, on ONE box -- the whole GPU suite, rocprofv3 kernel stats + FETCH / WRITE / MFMA PMC
# passes of bench.py (config 2, f16x3 and the bf16 throughput mode), the driver-style bench line, the lines of configs 1 / 4 / 5 and of
# the wav2vec2-base family, the geometry sweep of this tree against the round-5 library, kernel stats of the short and the long step,
# the race screen, the corpus through the batcher
mkdir -p gpurun_out
O=gpurun_out
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -8) > $O/r06_gpu_suite.log
bash tools/profile_bench.sh r06_f16x3 --precision f16x3 > $O/r06_profile_f16x3.log 2>&1
bash tools/profile_bench.sh r06_bf16 --precision bf16 > $O/r06_profile_bf16.log 2>&1
python tools/collect_profiles.py r06 f16x3:prof_r06_f16x3 bf16:prof_r06_bf16 > $O/r06_collect.log 2>&1
cp profiles/r06_traffic.json $O/r06_traffic.json
(timeout 900 python bench.py > $O/r06_bench_line.json 2> $O/r06_bench_stderr.log; echo "bench rc=$?" >> $O/r06_bench_stderr.log)
(timeout 600 python bench.py --config 4 --also "" > $O/r06_bench_config4.json 2>> $O/r06_bench_stderr.log)
(timeout 600 python bench.py --config 5 --also f16 > $O/r06_bench_config5.json 2>> $O/r06_bench_stderr.log)
(timeout 600 python bench.py --config 1 --also "" > $O/r06_bench_config1.json 2>> $O/r06_bench_stderr.log)
(timeout 600 python bench.py --encoder w2v2-base --also "" > $O/r06_bench_w2v2base.json 2>> $O/r06_bench_stderr.log)
(timeout 900 python bench.py --encoder xlsr-1b --also "" --cpu-sample 4 > $O/r06_bench_xlsr1b.json 2>> $O/r06_bench_stderr.log)
(timeout 1200 python bench.py --encoder xlsr-2b --also "" --cpu-sample 4 > $O/r06_bench_xlsr2b.json 2>> $O/r06_bench_stderr.log)
G="1:3 4:10 8:10 16:10 32:10 1:60 8:60"
rm -f $O/r06_geometry_sweep_final.log
(timeout 600 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/this tree    : /') >> $O/r06_geometry_sweep_final.log
(AMX_ABI_OVERRIDE=5 AMX_LIB_PATH=$PWD/build/ab/r05.so timeout 600 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/round-5 lib   : /') >> $O/r06_geometry_sweep_final.log
(timeout 600 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/this tree    : /') >> $O/r06_geometry_sweep_final.log
(AMX_ABI_OVERRIDE=5 AMX_LIB_PATH=$PWD/build/ab/r05.so timeout 600 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/round-5 lib   : /') >> $O/r06_geometry_sweep_final.log
rm -f $O/r06_race_screen.log
for g in "1 3" "4 10" "32 10" "8 60" "2 25"; do
  set -- $g
  (STRESS_N=$1 STRESS_SECONDS=$2 STRESS_ITERS=20 timeout 600 python tools/stress_repro.py 2>&1 | grep -v amdgpu.ids | tail -3) >> $O/r06_race_screen.log
  (STRESS_N=$1 STRESS_SECONDS=$2 STRESS_ITERS=20 STRESS_PACKED=1 timeout 600 python tools/stress_repro.py 2>&1 | grep -v amdgpu.ids | tail -3) >> $O/r06_race_screen.log
done
(timeout 900 python tools/corpus_throughput.py f16x3 4096 2>&1 | grep -v amdgpu.ids) > $O/r06_corpus_throughput.log
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for G2 in "4 10" "8 60"; do
  set -- $G2
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$O/trace_$1x$2 -- python3 $ROOT/tools/step_trace.py f16x3 $1 $2 20 > $ROOT/$O/step_$1x$2.log 2> $ROOT/$O/step_$1x$2.err
  find $ROOT/$O/trace_$1x$2 -name "*kernel_stats.csv" -exec cp {} $ROOT/$O/r06_kernel_stats_$1x$2s.csv \;
  rm -rf $ROOT/$O/trace_$1x$2
done
cd $ROOT
python - <<'PY'
import json
def r(x, n):
    return None if x is None else round(x, n)
for name in ("r06_bench_line", "r06_bench_config4", "r06_bench_config5", "r06_bench_config1", "r06_bench_w2v2base", "r06_bench_xlsr1b", "r06_bench_xlsr2b"):
    try:
        d = json.load(open(f"gpurun_out/{name}.json"))
        print(name, round(d["ms_per_step"], 3), "ms", round(d["value"]), "frames/s ok", d.get("ok"), "frac", r(d["roofline"]["frac"], 4),
              "whole_block", r(d["roofline"]["whole_block"]["frac"], 4), "conv0", r(d["roofline"]["conv_stage"]["conv0"]["frac"], 3),
              "spot", d.get("parity_spot_check", {}).get("max_abs"), "traffic", d["roofline"]["traffic"], "pass", d.get("pass"))
    except Exception as e:
        print(name, "ERROR", e)
PY
cat $O/r06_gpu_suite.log $O/r06_collect.log $O/r06_race_screen.log $O/r06_corpus_throughput.log; cut -c1-200 $O/r06_geometry_sweep_final.log; tail -3 $O/r06_bench_stderr.log
