#!/bin/bash
# round 6, run 19: bench.py through the real RCCL communicator path on a one-rank group (the gather, the status word, the barrier /
# max-over-ranks timing -- everything of --gpus N that one GPU can execute)
mkdir -p gpurun_out
O=gpurun_out
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533
(AMX_BENCH_FORCE_DIST=1 timeout 900 python bench.py --also "" --no-ragged > $O/r06_one_rank_rccl_bench.json 2> $O/r06_one_rank_rccl_bench.err; echo rc=$? >> $O/r06_one_rank_rccl_bench.err)
python -c "
import json; d=json.load(open('gpurun_out/r06_one_rank_rccl_bench.json')); print(d['ms_per_step'], d['value'], d['ok'], d['config'].get('parallelism'), d.get('gather'))"
tail -3 $O/r06_one_rank_rccl_bench.err
