"""Diagnostic (CPU only; runs the oracle, hence under tests/): how much of the 1e-3 logit gate do cheaper split-precision
schemes use?  Every nn.Linear-shaped product of the oracle (transformer QKV / out-proj / FFN, feature projection, heads --
87 % of the path's FLOPs; the conv stack and attention stay exact fp32 here, so the figures are LOWER bounds of the
error) is replaced by an emulation of a matrix-pipe scheme and the log-probs are compared with the exact fp32 oracle:

  f16x3      hi.hi + lo.hi + hi.lo on f16 planes (what the HIP path runs)
  f16        hi.hi only
  f16+mxfp8  hi.hi on f16 planes; the two cross terms with BOTH operands as MX-scaled fp8 (e4m3, one power-of-two scale
             per 32 elements along K): 16x16x128 scaled MFMAs run at twice the f16 rate, so the scheme would cost 2 f16-MFMA
             equivalents per product instead of 3
  f16+mxfp8b the same with only the LO operand of each cross term in fp8 and the HI operand kept in f16 (not a hardware
             mode: separates the two error sources)

    python tests/diagnostics/emulate_split_schemes.py [seconds]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F

from allophant_amd import spec as S, synthetic
from oracle import allophant_oracle as O

torch.set_num_threads(8)
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0


def split(x):
    hi = x.half().float()
    return hi, x - hi


def mxfp8(x):
    """Round to e4m3 with one shared power-of-two scale per 32 elements of the last dimension (OCP MX)."""
    shape = x.shape
    k = shape[-1]
    pad = (-k) % 32
    if pad:
        x = F.pad(x, (0, pad))
    b = x.reshape(*x.shape[:-1], -1, 32)
    amax = b.abs().amax(-1, keepdim=True).clamp_min(1e-38)
    scale = torch.exp2(torch.floor(torch.log2(amax)) - 8.0)  # block maximum lands in [256, 512): e4m3 tops out at 448
    q = (b / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * scale
    q = q.reshape(*x.shape)
    return q[..., :k] if pad else q


def product(a, w, scheme):
    ah, al = split(a)
    wh, wl = split(w)
    main = ah @ wh.t()
    if scheme == "f16":
        return main
    if scheme == "f16x3":
        return main + al.half().float() @ wh.t() + ah @ wl.half().float().t()
    if scheme == "f16+mxfp8":
        return main + mxfp8(al) @ mxfp8(wh).t() + mxfp8(ah) @ mxfp8(wl).t()
    if scheme == "f16+mxfp8b":
        return main + mxfp8(al) @ wh.t() + ah @ mxfp8(wl).t()
    raise ValueError(scheme)


spec = S.multitask_spec(S.xlsr_300m_encoder(), allophone_layer=True)
spec["shared_phones"] = 80
state = synthetic.make_state_dict(spec, seed=0)
tfi = synthetic.make_inventory(spec, 27, seed=0)
audio, lengths = synthetic.make_audio(1, int(seconds * 16000), seed=1234)
offsets = synthetic.category_offsets(spec)
exact, flen = O.predict(audio, lengths, state, spec, tfi, offsets)
real_linear = F.linear
for scheme in ("f16x3", "f16", "f16+mxfp8b", "f16+mxfp8"):
    def emulated(x, weight, bias=None, _scheme=scheme):
        out = product(x.reshape(-1, x.shape[-1]), weight, _scheme).reshape(*x.shape[:-1], weight.shape[0])
        return out if bias is None else out + bias
    F.linear = emulated
    try:
        got, _ = O.predict(audio, lengths, state, spec, tfi, offsets)
    finally:
        F.linear = real_linear
    worst = max((got[k] - exact[k]).abs().max().item() for k in exact)
    print(f"{scheme:12s} max |log-prob - exact fp32| = {worst:.2e}   (gate 1e-3)", flush=True)
