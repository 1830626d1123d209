import os, sys, copy, ctypes as C
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allophant_amd import spec as S, synthetic
from allophant_amd.estimator import Batch, Estimator
from oracle import allophant_oracle as O
import torch.nn.functional as F
torch.set_num_threads(32)
enc = S.xlsr_300m_encoder()
e = copy.deepcopy(enc)
e["conv_kernel"] = enc["conv_kernel"][:2]; e["conv_stride"] = enc["conv_stride"][:2]; e["layers"] = 1
spec = S.baseline_spec(e, 10)
sd = synthetic.make_state_dict(spec, seed=0)
audio, lengths = synthetic.make_audio(2, 4000, seed=5, ragged=True)
sd64 = {k: v.double() for k, v in sd.items()}
with torch.inference_mode():
    mask = O.mask_sequence(lengths)
    x64 = O.zero_mean_unit_var_norm(audio.double(), lengths, mask)
    p = "_acoustic_model._model.feature_extractor.conv_layers.0."
    h = F.conv1d(x64.unsqueeze(1), sd64[p + "conv.weight"], sd64[p + "conv.bias"], stride=5).transpose(1, 2)
    a0 = F.gelu(F.layer_norm(h, (512,), sd64[p + "layer_norm.weight"], sd64[p + "layer_norm.bias"], 1e-5))
est = Estimator(spec, sd, "cuda:0", "f16x3")
est.predict(Batch(audio.cuda(), lengths, torch.zeros(2, dtype=torch.long)), None, True, _keep_hidden=True)
T1 = a0.shape[1]
rows = 2 * T1
nbytes = rows * 512 * 2 * 2
buf = torch.empty(nbytes // 4 + 1024, dtype=torch.float32)
ld = C.c_int64(0)
code = est._lib.amx_debug_fetch(est._handle, 3, 0, C.c_void_p(buf.data_ptr()), buf.numel(), C.byref(ld))
assert code == 0
raw = buf.view(torch.float16)
hi = raw[: rows * 512].view(2, T1, 512).double()
lo = raw[rows * 512: 2 * rows * 512].view(2, T1, 512).double()
valid = (torch.arange(T1).unsqueeze(0) < ((lengths - 10) // 5 + 1).unsqueeze(1)).unsqueeze(-1)
e_hi = ((hi - a0).abs() * valid)
e_sum = ((hi + lo - a0).abs() * valid)
print("layer0 planes: hi-only max err", e_hi.max().item(), " hi+lo max err", e_sum.max().item(), "mean", (e_sum.sum() / valid.sum() / 512).item())
vals, idx = e_sum.flatten().topk(8)
for v, i in zip(vals, idx):
    n, t, c = (i // (T1 * 512)).item(), ((i // 512) % T1).item(), (i % 512).item()
    print(f"  err {v.item():.2e} n={n} t={t} c={c} ref {a0[n,t,c].item():.7f} hi {hi[n,t,c].item():.7f} lo {lo[n,t,c].item():.3e} true-res {(a0[n,t,c]-hi[n,t,c]).item():.3e}")
est.close()
