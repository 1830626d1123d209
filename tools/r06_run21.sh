#!/bin/bash
# round 6, run 21: long soak of the replay path (graphs are on by default): thousands of passes per geometry, every output bitwise the
# first run's (tools/stress_repro.py), padded and packed rows
mkdir -p gpurun_out
O=gpurun_out/r06_replay_soak.log
rm -f $O
for g in "1 3 3000" "4 10 2000" "8 10 1000" "32 10 400" "2 25 1000"; do
  set -- $g
  (STRESS_N=$1 STRESS_SECONDS=$2 STRESS_ITERS=$3 timeout 900 python tools/stress_repro.py 2>&1 | grep -v amdgpu.ids | tail -2) >> $O
  (STRESS_N=$1 STRESS_SECONDS=$2 STRESS_ITERS=$3 STRESS_PACKED=1 timeout 900 python tools/stress_repro.py 2>&1 | grep -v amdgpu.ids | tail -2) >> $O
done
cat $O
