"""Developer diagnostic: bitwise reproducibility of repeated predict() calls on a long padded batch; on a mismatch, the first
encoder layer whose hidden state differs from the first run and the rows that differ."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from allophant_amd import synthetic
from allophant_amd.estimator import Batch, Estimator

n = int(os.environ.get("STRESS_N", "4"))
seconds = float(os.environ.get("STRESS_SECONDS", "60"))
iters = int(os.environ.get("STRESS_ITERS", "30"))
spec = bench.build_spec()
state = synthetic.make_state_dict(spec, seed=0)
tfi = synthetic.make_inventory(spec, 27, seed=0)
L = int(seconds * 16000)
audio, lengths = synthetic.make_audio(n, L, seed=17, ragged=True)
batch = Batch(audio.cuda(), lengths, torch.zeros(n, dtype=torch.long))
est = Estimator(spec, state, "cuda:0", "f16x3")
layers = spec["layers"]


packed = os.environ.get("STRESS_PACKED", "0") == "1"  # without hidden-state capture a ragged batch takes the packed-row path


def run():
    if packed:
        pred = est.predict(batch, tfi)
        torch.cuda.synchronize()
        return pred._flat.clone(), []
    pred = est.predict(batch, tfi, _keep_hidden=True)
    torch.cuda.synchronize()
    hidden = [est.debug_fetch("hidden", i) for i in (0, 1, 2, 6, 12, 18, 24)]
    return pred._flat.clone(), hidden


ref_flat, ref_hidden = run()
bad = 0
for it in range(iters):
    flat, hidden = run()
    if torch.equal(flat, ref_flat):
        continue
    bad += 1
    msg = [f"iteration {it}: outputs differ (max {float((flat - ref_flat).abs().max()):.2e})"]
    for idx, (a, b) in zip((0, 1, 2, 6, 12, 18, 24), zip(hidden, ref_hidden)):
        d = (a - b).abs().amax(-1)  # [N, T]
        if float(d.max()) > 0:
            rows = torch.nonzero(d > 0)
            ns = sorted(set(rows[:, 0].tolist()))
            ts = rows[:, 1]
            msg.append(f"  hidden[{idx}] differs: utterances {ns}, frames {int(ts.min())}..{int(ts.max())} ({len(rows)} rows), max {float(d.max()):.2e}")
            big = torch.nonzero(d > 0.1 * d.max())
            msg.append(f"    rows above 10% of the max: utterances {sorted(set(big[:, 0].tolist()))} frames {int(big[:, 1].min())}..{int(big[:, 1].max())}")
            break
    print("\n".join(msg), flush=True)
print(f"{bad} of {iters} repeats differ from the first run ({n} x {seconds:.0f} s{', packed rows' if packed else ''})")
est.close()
