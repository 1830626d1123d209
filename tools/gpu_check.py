"""Developer harness: runs golden fixtures through the HIP path on a GPU box and prints per-stage max-abs errors."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from allophant_amd.estimator import Batch, Estimator  # noqa: E402
from golden_util import Golden, max_abs_valid_bm, max_abs_valid_tm  # noqa: E402


def run(name, precisions):
    g = Golden(name)
    sd = g.state_dict()
    for prec in precisions:
        t0 = time.time()
        est = Estimator(g.spec, sd, "cuda:0", prec)
        t1 = time.time()
        batch = Batch(g.audio.cuda(), g.lengths, torch.zeros(len(g.lengths), dtype=torch.long))
        pred = est.predict(batch, g.tfi, True, _keep_hidden=True)
        torch.cuda.synchronize()
        t2 = time.time()
        assert list(pred.outputs.keys()) == g.output_names, (list(pred.outputs.keys()), g.output_names)
        assert torch.equal(pred.lengths.cpu(), g.frame_lengths), (pred.lengths, g.frame_lengths)
        step = 8 if g.subsampled else 1
        conv = est.debug_fetch("conv")[:, :, ::step]
        msg = [f"conv {max_abs_valid_bm(conv, g.conv_out(), g.frame_lengths):.2e}"]
        for i in g.hidden_indices():
            hd = est.debug_fetch("hidden", i)[:, :, ::step]
            msg.append(f"h{i} {max_abs_valid_bm(hd, g.hidden(i), g.frame_lengths):.2e}")
        worst = 0.0
        for k in g.output_names:
            e = max_abs_valid_tm(pred.outputs[k].cpu(), g.logprobs(k), g.frame_lengths)
            worst = max(worst, e)
        raw = est.predict(batch, g.tfi, False)
        worst_logit = max(max_abs_valid_tm(raw.outputs[k].cpu(), g.logits(k), g.frame_lengths) for k in g.output_names)
        dec = est.greedy_decode(pred)
        mism = 0
        total = 0
        for k in g.output_names:
            for i in range(len(g.lengths)):
                tok, ts, sc = g.tokens(k, i)
                h = dec[k][i][0]
                total += 1
                if not (torch.equal(h.tokens, tok) and torch.equal(h.timesteps, ts)):
                    mism += 1
        print(f"[{name}/{prec}] create {t1 - t0:.2f}s fwd {t2 - t1:.3f}s | " + " ".join(msg) +
              f" | logprobs {worst:.2e} logits {worst_logit:.2e} | greedy mismatches {mism}/{total}", flush=True)
        est.close()


if __name__ == "__main__":
    names = sys.argv[1].split(",") if len(sys.argv) > 1 else ["g1_tiny_multitask"]
    precs = sys.argv[2].split(",") if len(sys.argv) > 2 else ["f16x3", "bf16x3", "f16", "bf16"]
    for n in names:
        run(n, precs)
