import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from allophant_amd import synthetic
from allophant_amd.estimator import Batch, Estimator
spec = bench.build_spec(); state = synthetic.make_state_dict(spec, seed=0)
est = Estimator(spec, state, torch.device("cuda", 0), "f16x3"); tfi = synthetic.make_inventory(spec, 27, seed=0)
audio, lengths = synthetic.make_audio(32, 160000, seed=1234)
dev_batch = Batch(audio.cuda(), lengths, torch.zeros(32, dtype=torch.long))
host_batch = Batch(audio.pin_memory(), lengths, torch.zeros(32, dtype=torch.long))
def run(batch, fetch):
    for _ in range(3):
        p = est.predict(batch, tfi)
        if fetch: p._flat.cpu()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        p = est.predict(batch, tfi)
        if fetch: out = p._flat.cpu()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 10 * 1e3
print(f"resident in HBM, outputs stay on device: {run(dev_batch, False):.3f} ms/step")
print(f"pinned host audio in, log-probs fetched to host (synchronous per step): {run(host_batch, True):.3f} ms/step")
