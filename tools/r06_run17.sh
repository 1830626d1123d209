#!/bin/bash
# round 6, run 17: upper bound of a weight prefetch -- the step with every layer on layer 0's weights (developer build, wrong results,
# timing only): the weights of a layer are then on the chip (L2 / MALL) when the layer starts
mkdir -p gpurun_out
O=gpurun_out/r06_shared_weights_bound.log
rm -f $O
G="4:10 8:10 16:10 32:10"
for i in 1 2; do
(AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 600 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/own weights    : /') >> $O
(AMX_DEV_SHARE_LAYER_WEIGHTS=1 AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so timeout 600 python tools/geometry_sweep.py f16x3 $G 2>&1 | grep -v amdgpu.ids | sed 's/^/layer 0 weights: /') >> $O
done
cut -c1-60 $O; grep "4 x 10" $O | cut -c200-420
