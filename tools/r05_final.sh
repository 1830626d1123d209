#!/bin/bash
# round 5: everything that describes the final tree, on ONE box -- the whole GPU suite, rocprofv3 kernel stats + FETCH / WRITE /
# MFMA PMC passes of bench.py (config 2, f16x3 and the bf16 throughput mode), the driver-style bench line, the lines of configs
# 4 / 5 and of the wav2vec2-base family, and the geometry sweep of this tree against the round-4 library
mkdir -p gpurun_out
O=gpurun_out
# output bits of this tree against the library of an earlier commit of the round (build/ab/head.so, tools/ab_build.sh): the late attention
# changes (store shape, instruction selection, DMA form) claim "same values"
if [ -f build/ab/head.so ]; then
  for P in f16x3 bf16x3; do
    AMX_LIB_PATH=$PWD/build/ab/head.so timeout 600 python tools/ab_bitwise.py $P 1:3 4:10 32:10 8:60 2>/dev/null > $O/bits_head_$P.txt
    timeout 600 python tools/ab_bitwise.py $P 1:3 4:10 32:10 8:60 2>/dev/null > $O/bits_tree_$P.txt
  done
  (for P in f16x3 bf16x3; do if diff -q $O/bits_head_$P.txt $O/bits_tree_$P.txt > /dev/null; then echo "$P: this tree is bitwise the earlier library on every geometry ($(wc -l < $O/bits_tree_$P.txt) digests)"; else echo "$P: DIFFERS"; diff $O/bits_head_$P.txt $O/bits_tree_$P.txt; fi; done) > $O/r05_final_bitwise.log
fi
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -8) > $O/r05_gpu_suite.log
bash tools/profile_bench.sh r05_f16x3 --precision f16x3 > $O/r05_profile_f16x3.log 2>&1
bash tools/profile_bench.sh r05_bf16 --precision bf16 > $O/r05_profile_bf16.log 2>&1
python tools/collect_profiles.py r05 f16x3:prof_r05_f16x3 bf16:prof_r05_bf16 > $O/r05_collect.log 2>&1
cp profiles/r05_traffic.json $O/r05_traffic.json
(timeout 900 python bench.py > $O/r05_bench_line.json 2> $O/r05_bench_stderr.log; echo "bench rc=$?" >> $O/r05_bench_stderr.log)
(timeout 600 python bench.py --config 4 --also "" > $O/r05_bench_config4.json 2>> $O/r05_bench_stderr.log)
(timeout 600 python bench.py --config 5 --also f16 > $O/r05_bench_config5.json 2>> $O/r05_bench_stderr.log)
(timeout 600 python bench.py --encoder w2v2-base --also "" > $O/r05_bench_w2v2base.json 2>> $O/r05_bench_stderr.log)
G="1:3 4:10 8:10 16:10 32:10 1:60"
rm -f $O/r05_geometry_sweep_final.log
(timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/this tree    : /') >> $O/r05_geometry_sweep_final.log
(AMX_ABI_OVERRIDE=4 AMX_LIB_PATH=$PWD/build/ab/r04.so timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/round-4 lib   : /') >> $O/r05_geometry_sweep_final.log
(timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/this tree    : /') >> $O/r05_geometry_sweep_final.log
# bitwise reproducibility of repeated passes (graph replays included: the packed form goes through predict() as a caller would)
rm -f $O/r05_race_screen.log
for g in "1 3" "4 10" "32 10" "8 60" "2 25"; do
  set -- $g
  (STRESS_N=$1 STRESS_SECONDS=$2 STRESS_ITERS=20 timeout 600 python tools/stress_repro.py 2>&1 | grep -v amdgpu.ids | tail -3) >> $O/r05_race_screen.log
  (STRESS_N=$1 STRESS_SECONDS=$2 STRESS_ITERS=20 STRESS_PACKED=1 timeout 600 python tools/stress_repro.py 2>&1 | grep -v amdgpu.ids | tail -3) >> $O/r05_race_screen.log
done
python - <<'PY'
import json
for name in ("r05_bench_line", "r05_bench_config4", "r05_bench_config5", "r05_bench_w2v2base"):
    try:
        d = json.load(open(f"gpurun_out/{name}.json"))
        print(name, round(d["ms_per_step"], 3), "ms", round(d["value"]), "frames/s ok", d.get("ok"), "frac", round(d["roofline"]["frac"], 4),
              "whole_block", round(d["roofline"]["whole_block"]["frac"], 4), "conv0", round(d["roofline"]["conv_stage"]["conv0"]["frac"], 3),
              "spot", d.get("parity_spot_check", {}).get("max_abs"), "traffic", d["roofline"]["traffic"])
    except Exception as e:
        print(name, "ERROR", e)
PY
cat $O/r05_final_bitwise.log $O/r05_gpu_suite.log $O/r05_collect.log $O/r05_race_screen.log; cut -c1-200 $O/r05_geometry_sweep_final.log; tail -3 $O/r05_bench_stderr.log
