"""Developer probe: one 32 x 10 s batch on one stream against two 16 x 10 s half batches on two streams (two handles), to see
whether independent launches de-phase the HBM-bound epilogues from the MFMA-bound main loops."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from allophant_amd import synthetic
from allophant_amd.estimator import Batch, Estimator

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
spec = bench.build_spec()
state = synthetic.make_state_dict(spec, seed=0)
tfi = synthetic.make_inventory(spec, 27, seed=0)
dev = torch.device("cuda", 0)
audio, lengths = synthetic.make_audio(32, 160000, seed=1234)
full = Batch(audio.cuda(), lengths, torch.zeros(32, dtype=torch.long))
halves = [Batch(audio[i * 16:(i + 1) * 16].cuda(), lengths[i * 16:(i + 1) * 16], torch.zeros(16, dtype=torch.long)) for i in range(2)]
ests = [Estimator(spec, state, dev, prec) for _ in range(2)]
streams = [torch.cuda.Stream(dev) for _ in range(2)]


def run_full():
    ests[0].predict(full, tfi, True)


def run_halves():
    for est, stream, half in zip(ests, streams, halves):
        with torch.cuda.stream(stream):
            est.predict(half, tfi, True)


for name, fn in (("one stream, 32 x 10 s", run_full), ("two streams, 2 x (16 x 10 s)", run_halves), ("one stream, 32 x 10 s", run_full)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"{prec} {name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per 32 utterances")
