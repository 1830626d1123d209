"""Developer harness: error of the conv feature extractor after 2..7 layers vs an fp64 CPU evaluation."""
import os, sys, copy
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from allophant_amd import spec as S, synthetic
from allophant_amd.estimator import Batch, Estimator
from oracle import allophant_oracle as O

torch.set_num_threads(32)
enc = S.xlsr_300m_encoder()
conv_dim = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for nconv in (2, 3, 5, 7):
    e = copy.deepcopy(enc)
    e["conv_kernel"] = enc["conv_kernel"][:nconv]
    e["conv_stride"] = enc["conv_stride"][:nconv]
    e["layers"] = 1
    e["conv_dim"] = conv_dim
    spec = S.baseline_spec(e, 10)
    sd = synthetic.make_state_dict(spec, seed=0)
    audio, lengths = synthetic.make_audio(2, 4000, seed=5, ragged=True)
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.inference_mode():
        mask = O.mask_sequence(lengths)
        x64 = O.zero_mean_unit_var_norm(audio.double(), lengths, mask)
        c64 = O.feature_encoder(x64, sd64, spec)
        c32 = O.feature_encoder(O.zero_mean_unit_var_norm(audio, lengths, mask), sd, spec)
    fl = O.downsampled_lengths(lengths, spec["conv_kernel"], spec["conv_stride"])
    m = (torch.arange(c64.shape[1]).unsqueeze(0) < fl.unsqueeze(1)).unsqueeze(-1)
    line = f"nconv={nconv} C={conv_dim} T={c64.shape[1]} cpu32 {((c32 - c64).abs() * m).max().item():.2e}"
    for prec in ("f16x3", "bf16x3", "f16"):
        est = Estimator(spec, sd, "cuda:0", prec)
        est.predict(Batch(audio.cuda(), lengths, torch.zeros(2, dtype=torch.long)), None, True, _keep_hidden=True)
        conv = est.debug_fetch("conv").double()
        err = ((conv - c64).abs() * m)
        idx = err.argmax()
        line += f" | {prec} {err.max().item():.2e} (mean {((err).sum() / m.sum() / conv_dim).item():.1e})"
        est.close()
    print(line, flush=True)
