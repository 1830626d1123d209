"""Developer check: step time of ONE ragged batch (device-resident, lengths ~ U[lo, hi] seconds) with the encoder layers on
packed rows (default for ragged batches) against the padded layout (AMX_FLAG_NO_PACK).

    python tools/packed_rows_probe.py [precision] [utterances] [lo seconds] [hi seconds]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from allophant_amd import synthetic
from allophant_amd.estimator import Batch, Estimator

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lo, hi = (float(sys.argv[3]), float(sys.argv[4])) if len(sys.argv) > 4 else (2.0, 10.0)
spec = bench.build_spec()
est = Estimator(spec, synthetic.make_state_dict(spec, seed=0), torch.device("cuda", 0), prec)
tfi = synthetic.make_inventory(spec, 27, seed=0)
g = torch.Generator().manual_seed(9)
lengths = torch.randint(int(lo * 16000), int(hi * 16000) + 1, (n,), generator=g)
lengths[0] = int(hi * 16000)
audio = torch.randn(n, int(hi * 16000), generator=g) * 0.1
for i in range(n):
    audio[i, int(lengths[i]):] = 0.0
batch = Batch(audio.cuda(), lengths, torch.zeros(n, dtype=torch.long))
for label, no_pack in (("packed rows", False), ("padded rows", True), ("packed rows", False), ("padded rows", True)):
    for _ in range(3):
        pred = est.predict(batch, tfi, _no_pack=no_pack)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        pred = est.predict(batch, tfi, _no_pack=no_pack)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    frames = int(pred.lengths.sum())
    est.timing_fetch()
    for _ in range(5):
        est.predict(batch, tfi, _no_pack=no_pack, _timing=True)
    torch.cuda.synchronize()
    classes = " ".join(f"{k}={v[0] / 5:.2f}" for k, v in est.timing_fetch().items())
    print(f"{prec} {n} utterances U[{lo:g}, {hi:g}] s, padding efficiency {frames / (n * int(pred.lengths.max())):.3f}: {label} "
          f"{dt * 1e3:7.3f} ms/step = {frames / dt:9.0f} valid frames/s   [{classes}]", flush=True)
est.close()
