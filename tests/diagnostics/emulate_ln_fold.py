"""Diagnostic (CPU only; runs the oracle, hence under tests/): does folding the pre-LN encoder's LayerNorm passes into the
products around them keep the error budget of the f16x3 planes?  (round-5 review, item 1: "emulate first".)

Three arithmetic schemes for the 24 encoder layers of the XLS-R-shape model, everything outside the layers exact fp32, every
Linear-shaped product of a layer as hi.hi + lo.hi + hi.lo on f16 planes with fp32 accumulation:

  current   LayerNorm in fp32 (two-pass), its output split onto planes, products on W                     (the round-5 path)
  fold      the residual stream stays fp32; the out-projection / FFN2 epilogues also write planes of u = (x - p) * s, with p
            (pivot) and s (an exact power of two) the mean and the scale the PREVIOUS statistics of that row gave, plus
            per-row partial sums of (x - p) and (x - p)^2 per 64-column block; QKV / FFN1 multiply u by W' = gamma (.) W and
            apply  y = alpha_m * acc + beta_m * c_n + d_n  with alpha = rstd / s, beta = -rstd * (mu - p),
            c = W gamma, d = W beta + b (fp64 at create)
  fold_nopivot   the same with p = 0 and s = 1 (what a pivot-free form would do; DC offsets then cancel in fp32 after the product)
  fold_planes    fold, with the residual stream kept ONLY as those planes between the products (no fp32 copy: the next residual is
                 (hi + lo) / s + p): what the two-plane modes run
  fold_planes_fixed   the same with the pivot and the scale of a row FIXED for the whole stack at what the first norm gave (one pair
                 of row parameters per pass instead of one per product)

Weight families: those of tests/test_gpu_range.py (plain, scales, student_t, ln_gain, outlier) plus `dc30`: every row of the
residual stream carries a DC offset of ~ 30 sigma.

    python tests/diagnostics/emulate_ln_fold.py [seconds] [family ...]
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F

from allophant_amd import spec as S, synthetic
from oracle import allophant_oracle as O
from tests.test_gpu_range import _variant

torch.set_num_threads(8)
AM = O._AM


def split(x):
    hi = x.half().float()
    lo = (x - hi).half().float()
    return hi, lo


def pack_scale(w):
    """the per-tensor power of two amx_create packs weights under: largest element into [4096, 8192)"""
    top = float(w.abs().max())
    return 2.0 ** (12 - math.floor(math.log2(top))) if top > 0 else 1.0


def x3(a, w):
    """a [M,K] fp32 (already what the planes hold), w [N,K] fp32 -> a.w^T as the three plane products, fp32 accumulate"""
    r = pack_scale(w)
    ah, al = split(a)
    wh, wl = split(w * r)
    return (ah @ wh.t() + al @ wh.t() + ah @ wl.t()) / r


def pow2_floor(x):
    return torch.exp2(torch.floor(torch.log2(x)))


def row_partials(e, width=64):
    """per-row mean and variance the way the epilogue + the finalize kernel form them: per 64-column block (one wave's 16 lanes x 4
    columns) the sum and the sum of squares about the BLOCK's own mean (fp32), merged over the blocks in ascending order with the
    pairwise update of Chan et al. -- no E[x^2] - E[x]^2 cancellation however far the pivot is from the row's mean"""
    M, D = e.shape
    b = e.reshape(M, D // width, width)
    s1 = b.reshape(M, D // width, width // 4, 4).sum(-1).sum(-1)            # [M, blocks]
    mb = s1 / width
    m2 = ((b - mb[..., None]) ** 2).reshape(M, D // width, width // 4, 4).sum(-1).sum(-1)
    mean = torch.zeros(M)
    for j in range(s1.shape[1]):
        mean = mean + s1[:, j]
    mean = mean / D
    M2 = torch.zeros(M)
    for j in range(s1.shape[1]):
        M2 = M2 + (m2[:, j] + width * (mb[:, j] - mean) ** 2)
    return mean, M2 / D


def encoder_layers(h, bias, state, spec, scheme):
    """h [N,T,D] after the positional convolution -> hidden_states list (as oracle.wav2vec2_hidden_states)"""
    eps = spec["eps"]
    H = spec["heads"]
    N, T, D = h.shape
    dh = D // H
    hidden = []
    fold = scheme.startswith("fold")
    pivot = scheme in ("fold", "fold_planes", "fold_planes_fixed")
    planes_only = scheme in ("fold_planes", "fold_planes_fixed")
    fixed = scheme == "fold_planes_fixed"  # pivot and scale of a row stay what the first norm of the stack gave
    x = h.reshape(N * T, D).clone()
    M = x.shape[0]
    if fold:
        # "rowprep": exact statistics of the stream as the positional convolution left it
        mu = x.mean(-1)
        var = ((x - mu[:, None]) ** 2).mean(-1)
        rstd = 1.0 / torch.sqrt(var + eps)
        p = mu if pivot else torch.zeros(M)
        s = pow2_floor(16.0 * rstd) if pivot else torch.ones(M)
        u = (x - p[:, None]) * s[:, None]
        alpha = rstd / s
        beta = -rstd * (mu - p)

    def consumer(prefixes, g, b):
        """LN(x) . W^T + bias for the stacked weights of `prefixes`"""
        W = torch.cat([state[q + ".weight"] for q in prefixes])
        bb = torch.cat([state[q + ".bias"] for q in prefixes])
        if not fold:
            a = F.layer_norm(x, (D,), g, b, eps)
            return x3(a, W) + bb
        Wg = (W.double() * g.double()[None, :])
        c = Wg.sum(-1).float()
        d = (W.double() @ b.double() + bb.double()).float()
        acc = x3(u, Wg.float())
        return alpha[:, None] * acc + beta[:, None] * c[None, :] + d[None, :]

    def producer(a, prefix):
        """x += a . W^T + bias; fold: also the planes of the new stream and its row statistics"""
        nonlocal x, u, p, s, alpha, beta
        v = x3(a, state[prefix + ".weight"]) + state[prefix + ".bias"] + x
        x = v
        if planes_only:
            pass  # (x is replaced by what the planes hold once they are formed below)
        if fold:
            e = v - p[:, None]
            u = e * s[:, None]
            m1, var = row_partials(e)
            rstd = 1.0 / torch.sqrt(var + eps)
            alpha = rstd / s
            beta = -rstd * m1
            if planes_only:
                # the stream exists as planes only: the next residual is what they hold, under the pivot / scale they were written with
                uh, ul = split(u)
                x = (uh + ul) / s[:, None] + p[:, None]
            if pivot and not fixed:
                # what the NEXT producer writes its planes under
                p = p + m1
                s = pow2_floor(16.0 * rstd)

    for i in range(spec["layers"]):
        hidden.append(x.reshape(N, T, D).clone())
        q = f"{AM}encoder.layers.{i}."
        qkv = consumer([q + "attention.q_proj", q + "attention.k_proj", q + "attention.v_proj"], state[q + "layer_norm.weight"],
                       state[q + "layer_norm.bias"])
        qq, kk, vv = (t.reshape(N, T, H, dh).transpose(1, 2) for t in qkv.split(D, -1))
        scores = torch.matmul(qq, kk.transpose(2, 3)) * (dh ** -0.5) + bias
        attn = torch.matmul(torch.softmax(scores, -1), vv).transpose(1, 2).reshape(N * T, D)
        # (the producer's planes are those of the NEW stream under the pivot / scale known before the product ran, and the
        # coefficients the next consumer applies refer to that pair: producer() forms them before it moves p and s on)
        producer(attn, q + "attention.out_proj")
        f1 = F.gelu(consumer([q + "feed_forward.intermediate_dense"], state[q + "final_layer_norm.weight"],
                             state[q + "final_layer_norm.bias"]))
        producer(f1, q + "feed_forward.output_dense")
    xf = F.layer_norm(x, (D,), state[AM + "encoder.layer_norm.weight"], state[AM + "encoder.layer_norm.bias"], eps)
    hidden.append(xf.reshape(N, T, D))
    return hidden


def run(audio, lengths, state, spec, tfi, offsets, scheme):
    with torch.inference_mode():
        eps = spec["eps"]
        mask = O.mask_sequence(lengths, None)
        xa = O.zero_mean_unit_var_norm(audio, lengths, mask)
        feats = O.feature_encoder(xa, state, spec)
        fl = O.downsampled_lengths(lengths, spec["conv_kernel"], spec["conv_stride"])
        T = feats.shape[1]
        fm = torch.arange(T).unsqueeze(0) < fl.unsqueeze(1)
        pp = AM + "feature_projection."
        h = F.layer_norm(feats, (feats.shape[-1],), state[pp + "layer_norm.weight"], state[pp + "layer_norm.bias"], eps)
        h = F.linear(h, state[pp + "projection.weight"], state[pp + "projection.bias"]) * fm.unsqueeze(-1)
        k = spec["pos_kernel"]
        pos = F.conv1d(h.transpose(1, 2), O._pos_conv_weight(state), state[AM + "encoder.pos_conv_embed.conv.bias"], padding=k // 2,
                       groups=spec["pos_groups"])
        if k % 2 == 0:
            pos = pos[:, :, :-1]
        h = h + F.gelu(pos).transpose(1, 2)
        bias = torch.zeros(h.shape[0], 1, 1, T)
        bias.masked_fill_(~fm[:, None, None, :], torch.finfo(torch.float32).min)
        if scheme == "exact":
            hidden, _, _ = O.wav2vec2_hidden_states(audio, lengths, state, spec)
        else:
            hidden = encoder_layers(h, bias, state, spec, scheme)
        logits = O.projection_forward([t.transpose(0, 1) for t in hidden], state, spec, tfi, offsets, fl)
        return {k_: F.log_softmax(v, -1) for k_, v in logits.items()}, fl, hidden


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    families = sys.argv[2:] or ["plain", "dc30", "scales", "student_t", "ln_gain", "outlier"]
    spec = S.multitask_spec(S.xlsr_300m_encoder(), allophone_layer=True)
    spec["shared_phones"] = 80
    tfi = synthetic.make_inventory(spec, 27, seed=0)
    offsets = synthetic.category_offsets(spec)
    audio, lengths = synthetic.make_audio(2, int(seconds * 16000), seed=1234, ragged=True)
    for fam in families:
        if fam in ("plain", "dc30"):
            state = synthetic.make_state_dict(spec, seed=0)
        else:
            state = _variant(spec, 0, fam)
        if fam == "dc30":
            # a DC offset on every channel of the stream: ~ 30 x the row's standard deviation (measured below)
            key = AM + "encoder.layers.0.attention.out_proj.bias"
            state[key] = state[key] + 80.0
        exact, fl, hid = run(audio, lengths, state, spec, tfi, offsets, "exact")
        row = hid[2].reshape(-1, hid[2].shape[-1])
        ratio = (row.mean(-1).abs() / row.std(-1)).median().item()
        line = f"{fam:10s} |mean|/sigma of stream rows (layer 2) = {ratio:6.2f}  "
        for scheme in ("current", "fold", "fold_planes", "fold_planes_fixed"):
            got, _, _ = run(audio, lengths, state, spec, tfi, offsets, scheme)
            worst = 0.0
            for k_ in exact:
                valid = (torch.arange(got[k_].shape[0]).unsqueeze(1) < fl.unsqueeze(0)).unsqueeze(-1)
                worst = max(worst, ((got[k_] - exact[k_]).abs() * valid).max().item())
            line += f" {scheme} {worst:.2e} "
        print(line, flush=True)


if __name__ == "__main__":
    main()
