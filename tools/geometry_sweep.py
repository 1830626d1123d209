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
    steps = 20
    walls = {}
    for mode, no_graph in (("eager", True), ("graph", False)):  # graph: the default -- the pass replays one HIP graph (ABI 5)
        for _ in range(4):
            pred = est.predict(batch, tfi, True, _no_graph=no_graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pred = est.predict(batch, tfi, True, _no_graph=no_graph)
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        walls[mode] = ((time.perf_counter() - t0) / steps, host / steps)
    wall, host = walls["graph"][0], walls["graph"][1] * steps
    # host cost of ONE pass with an idle GPU in front of it (the loops above fill the queue: their "host-side" time is the GPU's)
    enqueue = {}
    for mode, no_graph in (("eager", True), ("graph", False)):
        samples = []
        for _ in range(12):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            est.predict(batch, tfi, True, _no_graph=no_graph)
            samples.append(time.perf_counter() - t0)
        enqueue[mode] = sorted(samples)[len(samples) // 2] * 1e6
    torch.cuda.synchronize()
    est.timing_fetch()
    for _ in range(steps):
        est.predict(batch, tfi, True, _timing=True)
    torch.cuda.synchronize()
    tm = est.timing_fetch()
    kernels = sum(v[0] for v in tm.values()) / steps
    launches = sum(v[1] for v in tm.values()) // steps
    frames = int(pred.lengths.sum())
    print(f"{prec} {n} x {seconds} s: {wall * 1e3:8.3f} ms/step  host-side {host / steps * 1e3:6.3f} ms  "
          f"[eager: {walls['eager'][0] * 1e3:8.3f} ms/step  host-side {walls['eager'][1] * 1e3:6.3f} ms]  graphs {est.graph_info()}  one pass enqueued in {enqueue['graph']:.0f} us (eager {enqueue['eager']:.0f} us)  kernels {kernels:8.3f} ms "
          f"({launches} launches)  {frames / wall:10.0f} frames/s  "
          + " ".join(f"{k}={v[0] / steps:.2f}" for k, v in tm.items()))
est.close()
