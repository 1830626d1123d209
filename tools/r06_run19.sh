#!/bin/bash
# round 6, run 19: bench.py through the real RCCL communicator path on a one-rank group (the gather, the status word, the barrier /
# max-over-ranks timing -- everything of --gpus N that one GPU can execute), and the GPU suite on the final tree
mkdir -p gpurun_out
O=gpurun_out
(AMX_BENCH_FORCE_DIST=1 timeout 900 python bench.py --also "" --no-ragged > $O/r06_one_rank_rccl_bench.json 2> $O/r06_one_rank_rccl_bench.err; echo rc=$? >> $O/r06_one_rank_rccl_bench.err)
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -8) > $O/r06_gpu_suite.log
python -c "
import json; d=json.load(open('gpurun_out/r06_one_rank_rccl_bench.json')); print(d['ms_per_step'], d['value'], d['ok'], d.get('gather'), d['config'].get('parallelism'))"
tail -3 $O/r06_one_rank_rccl_bench.err; cat $O/r06_gpu_suite.log
