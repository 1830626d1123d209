#!/bin/bash
# round 6, run 12: after the plan / enqueue split -- output bits against the library before it, the corpus through reused buffers
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
for P in f16x3 bf16; do
  AMX_LIB_PATH=$PWD/build/ab/before_mix.so timeout 600 python tools/ab_bitwise.py $P 1:3 4:10 32:10 2>/dev/null > $O/bits_before_$P.txt
  timeout 600 python tools/ab_bitwise.py $P 1:3 4:10 32:10 2>/dev/null > $O/bits_tree_$P.txt
done
(for P in f16x3 bf16; do if diff -q $O/bits_before_$P.txt $O/bits_tree_$P.txt > /dev/null; then echo "$P: plan / enqueue split: bitwise the outputs of the library before it ($(wc -l < $O/bits_tree_$P.txt) digests)"; else echo "$P: DIFFERS"; diff $O/bits_before_$P.txt $O/bits_tree_$P.txt; fi; done) > $O/r06_plan_split_bitwise.log
cat $O/r06_plan_split_bitwise.log
(timeout 900 python tools/corpus_throughput.py f16x3 4096 2>&1 | grep -v amdgpu.ids) > $O/r06_corpus_throughput.log
cat $O/r06_corpus_throughput.log
(timeout 1200 python -m pytest tests/test_gpu_boundary.py tests/test_gpu_variant.py tests/test_c_host.py tests/test_gpu_long.py tests/test_gpu_range.py -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -6)
