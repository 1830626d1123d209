"""Developer check: wall time per predict() step with and without per-kernel HIP-event timing."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allophant_amd import spec as S, synthetic
from allophant_amd.estimator import Batch, Estimator
import bench

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
spec = bench.build_spec()
state = synthetic.make_state_dict(spec, seed=0)
est = Estimator(spec, state, torch.device("cuda", 0), prec)
tfi = synthetic.make_inventory(spec, 27, seed=0)
audio, lengths = synthetic.make_audio(32, 160000, seed=1234)
batch = Batch(audio.cuda(), lengths, torch.zeros(32, dtype=torch.long))
import os
for fused in ("0", "1", "0", "1"):
    os.environ["AMX_NO_FUSED_CONV_LN"] = fused
    for _ in range(3):
        est.predict(batch, tfi, True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        est.predict(batch, tfi, True)
    torch.cuda.synchronize()
    print(f"{prec} AMX_NO_FUSED_CONV_LN={fused}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms/step")
os.environ["AMX_NO_FUSED_CONV_LN"] = "0"
for timing in (False, True):
    for _ in range(3):
        est.predict(batch, tfi, True, _timing=timing)
    est.timing_fetch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        est.predict(batch, tfi, True, _timing=timing)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    tm = est.timing_fetch()
    print(f"{prec} timing={timing}: {dt * 1e3:.3f} ms/step; sum of kernel events {sum(v[0] for v in tm.values()) / 10:.3f} ms")
