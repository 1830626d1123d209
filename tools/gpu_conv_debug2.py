import os, sys, copy
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allophant_amd import spec as S, synthetic
from allophant_amd.estimator import Batch, Estimator
from oracle import allophant_oracle as O
torch.set_num_threads(32)
enc = S.xlsr_300m_encoder()
nconv = 2
e = copy.deepcopy(enc)
e["conv_kernel"] = enc["conv_kernel"][:nconv]; e["conv_stride"] = enc["conv_stride"][:nconv]; e["layers"] = 1
spec = S.baseline_spec(e, 10)
sd = synthetic.make_state_dict(spec, seed=0)
audio, lengths = synthetic.make_audio(2, 4000, seed=5, ragged=True)
sd64 = {k: v.double() for k, v in sd.items()}
with torch.inference_mode():
    mask = O.mask_sequence(lengths)
    x64 = O.zero_mean_unit_var_norm(audio.double(), lengths, mask)
    c64 = O.feature_encoder(x64, sd64, spec)
    # also the layer-0 output in fp64 (input of the GEMM layer)
    import torch.nn.functional as F
    p = "_acoustic_model._model.feature_extractor.conv_layers.0."
    h = F.conv1d(x64.unsqueeze(1), sd64[p + "conv.weight"], sd64[p + "conv.bias"], stride=5).transpose(1, 2)
    a0 = F.gelu(F.layer_norm(h, (512,), sd64[p + "layer_norm.weight"], sd64[p + "layer_norm.bias"], 1e-5))
fl = O.downsampled_lengths(lengths, spec["conv_kernel"], spec["conv_stride"])
m = (torch.arange(c64.shape[1]).unsqueeze(0) < fl.unsqueeze(1)).unsqueeze(-1)
est = Estimator(spec, sd, "cuda:0", "f16x3")
est.predict(Batch(audio.cuda(), lengths, torch.zeros(2, dtype=torch.long)), None, True, _keep_hidden=True)
conv = est.debug_fetch("conv").double()
err = ((conv - c64).abs() * m)
rowmax = err.amax(-1)
print("rows with max err > 2e-5:", (rowmax > 2e-5).sum().item(), "of", int(m.sum()))
vals, idx = err.flatten().topk(12)
for v, i in zip(vals, idx):
    n, t, c = np_ = (i // (err.shape[1] * 512)).item(), ((i // 512) % err.shape[1]).item(), (i % 512).item()
    # inputs of this output frame: layer-0 frames 2t..2t+2
    win = a0[n, 2 * t: 2 * t + 3]
    print(f"err {v.item():.2e} n={n} t={t} c={c} got {conv[n,t,c].item():.6f} ref {c64[n,t,c].item():.6f} row-mean-err {err[n,t].mean().item():.1e} "
          f"window |a0| max {win.abs().max().item():.3f} min-nonzero {win.abs()[win.abs()>0].min().item():.2e}")
# error by channel and by row
print("per-channel max err (top 5):", err.amax((0, 1)).topk(5))
print("per-row max err (top 5):", rowmax.flatten().topk(5))
est.close()
