#!/bin/bash
# Round-5 advisor finding: the replay fault (0x01010101 in the non-finite counter of every other replay) went away after TWO changes
# that were never separated -- (a) no memset / memcpy NODE in a recording (zero fills and device copies became kernels), (b) the graph
# template kept alive as long as its executable.  The developer build can put each one back; the regression test says which one
# brings the fault back.
mkdir -p gpurun_out
O=gpurun_out/r06_graph_fault.log
rm -f $O
T="tests/test_gpu_graph.py::test_replays_between_eager_bursts_stay_clean tests/test_gpu_graph.py::test_graph_replay_is_bitwise_the_eager_pass_padded"
for V in "" "AMX_GRAPH_MEMSET_NODES=1" "AMX_GRAPH_DROP_TEMPLATE=1" "AMX_GRAPH_MEMSET_NODES=1 AMX_GRAPH_DROP_TEMPLATE=1"; do
  echo "=== developer library, switches: ${V:-none}" >> $O
  (env $V AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 900 python -m pytest $T -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl\|amdgpu.ids" | tail -12) >> $O
done
cat $O
