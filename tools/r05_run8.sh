#!/bin/bash
# round 5: the group-norm first layer on the matrix pipe with covariance statistics -- tests and same-box A/B (wav2vec2-base shape)
mkdir -p gpurun_out
O=gpurun_out
(timeout 900 python -m pytest tests/test_gpu_conv0.py tests/test_gpu_variant.py -x -q -m gpu -s 2>&1 | grep -v "^$" | grep "conv0 mfma\|passed\|failed\|Error\|assert" | tail -20) > $O/r05_gn_tests.log
D=$PWD/build/liballophant_amx_dev.so
rm -f $O/r05_gn_ab.log
for i in 1 2; do
(AMX_LIB_PATH=$D AMX_NO_CONV0_MFMA=1 timeout 300 python bench.py --encoder w2v2-base --also "" --no-cpu-baseline --no-ragged --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('VALU kernel + recomputed statistics :', round(d['ms_per_step'],3), 'ms', d['kernels']['conv0'], 'ok', d.get('ok'), d.get('parity_spot_check',{}).get('max_abs'))") >> $O/r05_gn_ab.log
(AMX_LIB_PATH=$D timeout 300 python bench.py --encoder w2v2-base --also "" --no-cpu-baseline --no-ragged --steps 10 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('MFMA kernel + covariance statistics :', round(d['ms_per_step'],3), 'ms', d['kernels']['conv0'], 'ok', d.get('ok'), d.get('parity_spot_check',{}).get('max_abs'))") >> $O/r05_gn_ab.log
done
cat $O/r05_gn_tests.log $O/r05_gn_ab.log
