"""Developer check on the GPU box: BASELINE configs 4 (hierarchical, 64 x 5 s, 48 phones) and 5 (long-form, 8 x 60 s,
fp16 single plane + f16x3, 200 phones) at full size -- finite, normalised, batch-independent outputs + throughput."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allophant_amd import spec as S, synthetic
from allophant_amd.estimator import Batch, Estimator


def run(name, spec, n, seconds, phones, precision):
    state = synthetic.make_state_dict(spec, seed=0)
    est = Estimator(spec, state, torch.device("cuda", 0), precision)
    tfi = synthetic.make_inventory(spec, phones, seed=0)
    length = int(seconds * 16000)
    audio, lengths = synthetic.make_audio(n, length, seed=99, ragged=True)
    batch = Batch(audio.cuda(), lengths, torch.zeros(n, dtype=torch.long))
    pred = est.predict(batch, tfi)
    T = pred.outputs["phoneme"].shape[0]
    valid = (torch.arange(T).unsqueeze(1) < pred.lengths.unsqueeze(0)).cuda()
    worst_norm = 0.0
    for k, out in pred.outputs.items():
        assert torch.isfinite(out[valid]).all(), (name, k)
        worst_norm = max(worst_norm, (out.exp().sum(-1)[valid] - 1).abs().max().item())
    i = n // 2
    ni = int(lengths[i])
    solo = est.predict(Batch(audio[i:i + 1, :ni].contiguous().cuda(), lengths[i:i + 1], torch.zeros(1, dtype=torch.long)), tfi)
    ti = int(pred.lengths[i])
    dev = max((pred.outputs[k][:ti, i] - solo.outputs[k][:ti, 0]).abs().max().item() for k in pred.outputs)
    for _ in range(2):
        est.predict(batch, tfi)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        est.predict(batch, tfi)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    frames = int(pred.lengths.sum())
    print(f"{name} [{precision}]: T={T} outputs={len(pred.outputs)} phoneme classes={pred.outputs['phoneme'].shape[-1]} "
          f"|sum p - 1| <= {worst_norm:.1e}; batch-vs-solo max dev {dev:.1e}; {dt * 1e3:.2f} ms/step, {frames / dt:.0f} valid frames/s "
          f"({n * T / dt:.0f} padded frames/s)", flush=True)
    est.close()


enc = S.xlsr_300m_encoder()
hier = S.hierarchical_spec(enc, allophone_layer=True)
hier["shared_phones"] = 80
run("config 4 hierarchical 64x5s ['es','it']-sized inventory", hier, 64, 5.0, 48, "f16x3")
multi = S.multitask_spec(enc, allophone_layer=True)
multi["shared_phones"] = 80
for prec in ("f16", "f16x3"):
    run("config 5 long-form 8x60s 200 phones", multi, 8, 60.0, 200, prec)
