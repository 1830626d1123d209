#!/bin/bash
# round 5, second box: the whole GPU suite on this tree, the driver-style bench line, the geometry sweep (with the one-pass
# enqueue cost: eager against graph replay) and the conv0 tap ablation
mkdir -p gpurun_out
O=gpurun_out
(timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > $O/r05_gpu_suite.log
(timeout 900 python bench.py > $O/r05_bench_line.json 2> $O/r05_bench_stderr.log; echo "bench rc=$?" >> $O/r05_bench_stderr.log)
G="1:3 4:10 8:10 16:10 32:10"
(timeout 500 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids) > $O/r05_geometry_sweep2.log
(AMX_LIB_PATH=$PWD/build/ab/c0_notaps.so timeout 300 python tools/geometry_sweep.py f16x3 32:10 2>&1 | grep -v amdgpu.ids) > $O/r05_conv0_one_tap.log
(timeout 300 python tools/geometry_sweep.py f16x3 32:10 2>&1 | grep -v amdgpu.ids) >> $O/r05_conv0_one_tap.log
tail -n 30 $O/r05_gpu_suite.log $O/r05_bench_stderr.log $O/r05_geometry_sweep2.log $O/r05_conv0_one_tap.log; python -c "
import json; d=json.load(open('$O/r05_bench_line.json')); print({k: d[k] for k in ('value','ms_per_step','ok','launch_collapse')}); print(d['roofline']['frac'], d['roofline']['whole_block']); print(d['cpu_baseline']); print(d.get('pcie_inclusive'))"
