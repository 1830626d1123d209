"""Developer diagnostic: reproduces the geometry switch 32 x 10 s -> 8 x 60 s in one estimator and prints the range counters."""
import ctypes as C
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


def count():
    frames = C.c_int64(-1)
    stream = torch.cuda.current_stream().cuda_stream
    rc = est._lib.amx_check_finite(est._handle, C.c_void_p(stream), C.byref(frames))
    return rc, frames.value


for n, seconds in ((32, 10), (8, 60), (32, 10), (8, 60)):
    audio, lengths = synthetic.make_audio(n, seconds * 16000, seed=1234)
    batch = Batch(audio.cuda(), lengths, torch.zeros(n, dtype=torch.long))
    for step in range(6):
        for no_graph in (True, False):
            try:
                pred = est.predict(batch, tfi, True, _no_graph=no_graph)
                torch.cuda.synchronize()
                finite = bool(torch.isfinite(pred._flat).all())
                print(f"{n} x {seconds} s step {step} no_graph={no_graph}: outputs finite {finite}  check_finite {count()}  graphs {est.graph_info()}", flush=True)
            except FloatingPointError as exc:
                print(f"{n} x {seconds} s step {step} no_graph={no_graph}: RAISED {str(exc)[:60]}", flush=True)
    # a burst without synchronisation (the host runs ahead), then one synchronize
    try:
        for _ in range(24):
            est.predict(batch, tfi, True)
        est.synchronize()
        print(f"{n} x {seconds} s burst ok  graphs {est.graph_info()}", flush=True)
    except FloatingPointError as exc:
        print(f"{n} x {seconds} s burst RAISED {str(exc)[:60]}", flush=True)
