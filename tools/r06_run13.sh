#!/bin/bash
# round 6, run 13: consumer coefficients fetched in front of the first round, producer pairs fetched two pairs ahead -- against the commit before
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOT/gpurun_out
mkdir -p $O
rm -f $O/r06_epilogue_waits_ab.log
for rep in 1 2 3; do
(timeout 400 python tools/geometry_sweep.py f16x3 32:10 8:60 16:10 2>&1 | grep -v amdgpu.ids | sed "s/^/this tree    : /") >> $O/r06_epilogue_waits_ab.log
(AMX_LIB_PATH=$PWD/build/ab/prev.so timeout 400 python tools/geometry_sweep.py f16x3 32:10 8:60 16:10 2>&1 | grep -v amdgpu.ids | sed "s/^/commit before: /") >> $O/r06_epilogue_waits_ab.log
done
python - <<'PY'
import re
for l in open('gpurun_out/r06_epilogue_waits_ab.log'):
    m=re.match(r"(.*?): f16x3 (\d+ x \d+) s:\s+([\d.]+) ms/step.*kernels\s+([\d.]+) ms.*?gemm_pp=([\d.]+).*?attention=([\d.]+) rownorm=([\d.]+).*?gemm_ln=([\d.]+)",l)
    if m: print(f"{m.group(1):14s} {m.group(2):8s} step {m.group(3):>7s}  gemm_pp {m.group(5):>6s} attn {m.group(6)} rownorm {m.group(7)}")
PY
for P in f16x3; do
  AMX_LIB_PATH=$PWD/build/ab/prev.so timeout 600 python tools/ab_bitwise.py $P 32:10 2>/dev/null > $O/bits_before_$P.txt
  timeout 600 python tools/ab_bitwise.py $P 32:10 2>/dev/null > $O/bits_tree_$P.txt
  diff $O/bits_before_$P.txt $O/bits_tree_$P.txt > /dev/null && echo "$P: bitwise the commit before" || echo "$P: DIFFERS"
done
