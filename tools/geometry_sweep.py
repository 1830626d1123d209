"""Developer check: predict() step time, frames/s and per-class kernel time over batch geometries (small batches are
launch-bound: compare the wall time with the sum of kernel times).

    python tools/geometry_sweep.py [precision] [N:seconds ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from allophant_amd import synthetic
from allophant_amd.estimator import Batch, Estimator

args = sys.argv[1:]
prec = args[0] if args else "f16x3"
geometries = [tuple(a.split(":")) for a in args[1:]] or [("1", "3"), ("1", "10"), ("4", "10"), ("8", "10"), ("32", "10"), ("1", "60")]
spec = bench.build_spec()
state = synthetic.make_state_dict(spec, seed=0)
est = Estimator(spec, state, torch.device("cuda", 0), prec)
tfi = synthetic.make_inventory(spec, 27, seed=0)
for n, seconds in geometries:
    n, length = int(n), int(float(seconds) * 16000)
    audio, lengths = synthetic.make_audio(n, length, seed=1234)
    batch = Batch(audio.cuda(), lengths, torch.zeros(n, dtype=torch.long))
    for _ in range(3):
        pred = est.predict(batch, tfi, True)
    torch.cuda.synchronize()
    steps = 20
    t0 = time.perf_counter()
    for _ in range(steps):
        pred = est.predict(batch, tfi, True)
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / steps
    est.timing_fetch()
    for _ in range(steps):
        est.predict(batch, tfi, True, _timing=True)
    torch.cuda.synchronize()
    tm = est.timing_fetch()
    kernels = sum(v[0] for v in tm.values()) / steps
    launches = sum(v[1] for v in tm.values()) // steps
    frames = int(pred.lengths.sum())
    print(f"{prec} {n} x {seconds} s: {wall * 1e3:8.3f} ms/step  host-side {host / steps * 1e3:6.3f} ms  kernels {kernels:8.3f} ms "
          f"({launches} launches)  {frames / wall:10.0f} frames/s  "
          + " ".join(f"{k}={v[0] / steps:.2f}" for k, v in tm.items()))
est.close()
