#!/bin/bash
# round 6, run 10: lo planes by v_fma_mix in every epilogue ("same bits": SHA-256 of the outputs against the library of the commit before),
# the whole GPU suite on the product library, step times (this tree / the commit before / fold off), bitwise reproducibility of repeated passes
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
for P in f16x3 bf16x3; do
  AMX_LIB_PATH=$PWD/build/ab/before_mix.so timeout 600 python tools/ab_bitwise.py $P 1:3 4:10 32:10 8:60 2>/dev/null > $O/bits_before_$P.txt
  timeout 600 python tools/ab_bitwise.py $P 1:3 4:10 32:10 8:60 2>/dev/null > $O/bits_tree_$P.txt
done
(for P in f16x3 bf16x3; do if diff -q $O/bits_before_$P.txt $O/bits_tree_$P.txt > /dev/null; then echo "$P: lo planes by v_fma_mixlo/hi: bitwise the outputs of the commit before on every geometry ($(wc -l < $O/bits_tree_$P.txt) digests)"; else echo "$P: DIFFERS"; diff $O/bits_before_$P.txt $O/bits_tree_$P.txt; fi; done) > $O/r06_mix_split_bitwise.log
cat $O/r06_mix_split_bitwise.log
(timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "version\|Hostname\|Librccl" | tail -12) > $O/r06_run10_suite.log
cat $O/r06_run10_suite.log
rm -f $O/r06_mix_split_ab.log
for rep in 1 2; do
(timeout 400 python tools/geometry_sweep.py f16x3 32:10 8:60 4:10 2>&1 | grep -v amdgpu.ids | sed "s/^/this tree    : /") >> $O/r06_mix_split_ab.log
(AMX_LIB_PATH=$PWD/build/ab/before_mix.so timeout 400 python tools/geometry_sweep.py f16x3 32:10 8:60 4:10 2>&1 | grep -v amdgpu.ids | sed "s/^/commit before: /") >> $O/r06_mix_split_ab.log
(AMX_NO_LN_FOLD=1 AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 400 python tools/geometry_sweep.py f16x3 32:10 8:60 4:10 2>&1 | grep -v amdgpu.ids | sed "s/^/fold off     : /") >> $O/r06_mix_split_ab.log
(AMX_ABI_OVERRIDE=5 AMX_LIB_PATH=$PWD/build/ab/r05.so timeout 400 python tools/geometry_sweep.py f16x3 32:10 8:60 4:10 2>&1 | grep -v amdgpu.ids | sed "s/^/round-5 lib  : /") >> $O/r06_mix_split_ab.log
done
python - <<'PY'
import re
for l in open('gpurun_out/r06_mix_split_ab.log'):
    m=re.match(r"(.*?): f16x3 (\d+ x \d+) s:\s+([\d.]+) ms/step.*kernels\s+([\d.]+) ms.*?gemm_pp=([\d.]+).*?attention=([\d.]+) rownorm=([\d.]+).*?gemm_ln=([\d.]+)",l)
    if m: print(f"{m.group(1):14s} {m.group(2):8s} step {m.group(3):>7s}  gemm_pp {m.group(5):>6s} attn {m.group(6)} rownorm {m.group(7)} gemm_ln {m.group(8)}")
PY
rm -f $O/r06_race_screen.log
for g in "1 3" "4 10" "32 10" "8 60"; do
  set -- $g
  (STRESS_N=$1 STRESS_SECONDS=$2 STRESS_ITERS=20 timeout 600 python tools/stress_repro.py 2>&1 | grep -v amdgpu.ids | tail -3) >> $O/r06_race_screen.log
  (STRESS_N=$1 STRESS_SECONDS=$2 STRESS_ITERS=20 STRESS_PACKED=1 timeout 600 python tools/stress_repro.py 2>&1 | grep -v amdgpu.ids | tail -3) >> $O/r06_race_screen.log
done
cat $O/r06_race_screen.log
