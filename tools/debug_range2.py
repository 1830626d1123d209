"""Developer diagnostic: the burst pattern of tools/geometry_sweep.py across a geometry switch, range report caught and printed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from allophant_amd import synthetic
from allophant_amd.estimator import Batch, Estimator

spec = bench.build_spec()
state = synthetic.make_state_dict(spec, seed=0)
est = Estimator(spec, state, torch.device("cuda", 0), "f16x3")
tfi = synthetic.make_inventory(spec, 27, seed=0)
for n, seconds in ((32, 10), (8, 60), (4, 10), (8, 60)):
    audio, lengths = synthetic.make_audio(n, seconds * 16000, seed=1234)
    batch = Batch(audio.cuda(), lengths, torch.zeros(n, dtype=torch.long))
    for mode, no_graph in (("eager", True), ("graph", False), ("eager", True), ("graph", False)):
        try:
            for i in range(24):
                pred = est.predict(batch, tfi, True, _no_graph=no_graph)
            torch.cuda.synchronize()
            finite = bool(torch.isfinite(pred._flat).all())
            for i in range(6):
                torch.cuda.synchronize()
                est.predict(batch, tfi, True, _no_graph=no_graph)
            est.synchronize()
            print(f"{n} x {seconds} s {mode}: ok, last outputs finite {finite}, graphs {est.graph_info()}", flush=True)
        except FloatingPointError as exc:
            print(f"{n} x {seconds} s {mode} pass {i}: RAISED ... {str(exc)[-120:]}", flush=True)
