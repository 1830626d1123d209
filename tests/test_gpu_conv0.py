"""conv layer 0 on the matrix pipe (round 5: ``conv0_mfma_kernel``, csrc/amx_rowops.hip) against the CPU oracle.

The kernel replaces the fp32 FMA taps and the two LayerNorm reductions of ``conv0_kernel`` by one split-precision MFMA per
(16 channels x 16 frames) and by per-frame statistics taken from fp64 tables (mean and covariance of the conv rows over the
channels).  Its fp16 planes are range-safe only through the per-frame power-of-two scale, and its statistics only as exact as
those identities: so the audio here is what would break either -- digital silence, a quiet passage (1e-4 of full scale) next to
a loud one, a DC offset larger than the signal, single-sample clicks, a frame whose samples span 2^20 -- at wav2vec 2.0's conv
shape (k = 10, s = 5, C = 512; reference call site acoustic_model.py:839, transformers' Wav2Vec2LayerNormConvLayer).

Gate: the conv extractor's OUTPUT (seven layers, LayerNorm + GELU each; `conv_out` of the oracle) within 1e-3 like every stage
gate, and in fact within 5e-5 (logged); log-probabilities of a one-layer model within 1e-3.
"""
import pytest
import torch

from allophant_amd import spec as S, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from allophant_amd import estimator, lib

    assert lib.load() is not None
    return estimator


def _hard_audio(n, length, seed):
    g = torch.Generator().manual_seed(seed)
    audio = torch.randn(n, length, generator=g) * 0.1
    lengths = torch.full((n,), length, dtype=torch.int64)
    third = length // 3
    audio[0, :third] = 0.0                                   # digital silence, then speech-like noise
    audio[1, third: 2 * third] *= 1e-4                        # a quiet passage between loud ones
    audio[2] += 3.0                                           # DC offset 30 x the signal (do_normalize removes the global mean only)
    audio[2, third:] -= 6.0                                   # ... which changes sign mid-utterance
    audio[3, ::997] += 50.0                                   # isolated clicks
    if n > 4:
        ramp = torch.logspace(-6, 0, length)                  # every frame spans a different scale; within a frame up to 2^20
        audio[4] = torch.randn(length, generator=g) * ramp
        audio[4, 5::10] *= 1e-6
    if n > 5:
        lengths[5] = length - 4321                            # a ragged tail (zero padding behind it)
        audio[5, int(lengths[5]):] = 0.0
    return audio, lengths


def _max_abs_valid(got, want, frame_lengths):
    worst = 0.0
    for i, t in enumerate(frame_lengths.tolist()):
        worst = max(worst, float((got[i, :t] - want[i, :t]).abs().max()))
    return worst


@pytest.mark.parametrize("precision", ["f16x3", "bf16x3", "f16", "bf16"])
def test_conv_extractor_on_hard_audio_against_oracle(amd, precision):
    from oracle import allophant_oracle as O

    enc = S.xlsr_300m_encoder()
    enc["layers"] = 1  # the conv stack is what is under test; one encoder layer keeps the oracle fast
    spec = S.multitask_spec(enc, ["syllabic", "long"], allophone_layer=True)
    spec["shared_phones"] = 80
    state = synthetic.make_state_dict(spec, seed=21)
    tfi = synthetic.make_inventory(spec, 27, seed=21)
    audio, lengths = _hard_audio(6, 40000, seed=77)
    ref, ref_len, inter = O.predict(audio, lengths, state, spec, tfi, synthetic.category_offsets(spec), keep_intermediates=True)
    est = amd.Estimator(spec, state, "cuda:0", precision)
    pred = est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(6, dtype=torch.long)), tfi, True, _keep_hidden=True)
    est.check_finite()
    assert torch.equal(pred.lengths.cpu(), ref_len)
    conv = est.debug_fetch("conv")
    worst_conv = _max_abs_valid(conv, inter["conv_out"], ref_len)
    split = precision.endswith("x3")
    print(f"[conv0 mfma] {precision}: conv extractor output max abs {worst_conv:.3g}")
    # conv layer 0 itself is fp32-grade in every mode; layers 1-6 run in the handle's mode
    assert worst_conv < (1e-3 if split else 6e-2), worst_conv
    if split:
        worst = 0.0
        for name, expected in ref.items():
            got = pred.outputs[name].cpu()
            for i, t in enumerate(ref_len.tolist()):
                worst = max(worst, float((got[:t, i] - expected[:t, i]).abs().max()))
        assert worst < 1e-3, worst
    est.close()


def test_first_layer_alone_matches_fp32_to_rounding(amd):
    """conv layer 0 in isolation: a model whose conv layers 1-6 are (nearly) the identity on the first layer's output is not
    expressible, so the first layer is read back through the raw workspace hook instead: planes of conv-0's output (hi + lo)
    against an fp64 evaluation of the first layer, on the hard audio, to 2e-5 absolute on O(1) values (the planes hold 22 bits)."""
    import torch.nn.functional as F

    enc = S.xlsr_300m_encoder()
    enc["layers"] = 1
    spec = S.baseline_spec(enc, 12)
    state = synthetic.make_state_dict(spec, seed=5)
    length, n_plain = 24000, 2
    hard, _ = _hard_audio(5, length, seed=3)
    plain, _ = synthetic.make_audio(n_plain, length, seed=9)
    # the raw buffer (conv-0 output planes) is reused by conv layers 2 and 4 for their own, shorter outputs: its first
    # N * T3 rows are gone after the pass -- two plain utterances take that place, the hard ones are read back whole
    audio = torch.cat([plain, hard], 0)
    n = audio.shape[0]
    lengths = torch.full((n,), length, dtype=torch.int64)
    est = amd.Estimator(spec, state, "cuda:0", "f16x3")
    est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(n, dtype=torch.long)), None, True, _keep_hidden=True)
    torch.cuda.synchronize()
    import ctypes as C

    T1 = (length - 10) // 5 + 1
    T3 = ((T1 - 3) // 2 + 1 - 3) // 2 + 1
    assert n * T3 <= n_plain * T1
    raw = torch.empty(n * T1 * 512, dtype=torch.float32)  # 2 planes x 2 bytes = 4 bytes per element: same count of fp32 words
    ld = C.c_int64(0)
    code = est._lib.amx_debug_fetch(est._handle, 3, 0, C.c_void_p(raw.data_ptr()), raw.numel(), C.byref(ld))
    assert code == 0
    halves = raw.view(torch.float16).view(n * T1, 512 // 32, 2, 32)  # interleaved planes: [row][block][hi | lo][32]
    got = (halves[:, :, 0].float() + halves[:, :, 1].float()).reshape(n, T1, 512)[n_plain:]
    # oracle's first layer (input normalisation, conv, LayerNorm over channels, exact GELU): acoustic_model.py:762-767, 839
    p = "_acoustic_model._model.feature_extractor.conv_layers.0."
    mask = torch.arange(length).unsqueeze(0) < lengths.unsqueeze(1)
    x = audio.double()
    mean = (x * mask).sum(1, keepdim=True) / lengths.unsqueeze(1)
    var = (((x - mean) ** 2) * mask).sum(1, keepdim=True) / lengths.unsqueeze(1)
    x = ((x - mean) / torch.sqrt(var + 1e-7)) * mask
    y = F.conv1d(x.unsqueeze(1), state[p + "conv.weight"].double(), state[p + "conv.bias"].double(), stride=5).transpose(1, 2)
    y = F.layer_norm(y, (512,), state[p + "layer_norm.weight"].double(), state[p + "layer_norm.bias"].double(), 1e-5)
    want = F.gelu(y).float()[n_plain:]
    worst = float((got - want).abs().max())
    print(f"[conv0 mfma] first layer vs fp64 evaluation: max abs {worst:.3g}")
    assert worst < 2e-5, worst
    est.close()


@pytest.mark.parametrize("precision", ["f16x3", "bf16x3"])
def test_group_norm_first_layer_on_hard_audio_against_oracle(amd, precision):
    """The group-norm family (wav2vec2-base shape): GroupNorm statistics from the utterance's 10 x 10 sample covariance
    (``conv0_gn_cov_kernel``: the convolution is not evaluated a second time) and the apply pass on the matrix pipe
    (``conv0_mfma_kernel<GN>``), on the same adversarial audio; the statistics run over every frame of the PADDED length
    (transformers' Wav2Vec2GroupNormConvLayer normalises the padded batch tensor), so the ragged tail matters."""
    from oracle import allophant_oracle as O

    enc = S.wav2vec2_base_encoder()
    enc["layers"] = 1
    spec = S.multitask_spec(enc, ["syllabic", "long"], allophone_layer=True)
    spec["shared_phones"] = 80
    state = synthetic.make_state_dict(spec, seed=23)
    tfi = synthetic.make_inventory(spec, 27, seed=23)
    audio, lengths = _hard_audio(6, 40000, seed=78)
    ref, ref_len, inter = O.predict(audio, lengths, state, spec, tfi, synthetic.category_offsets(spec), keep_intermediates=True)
    est = amd.Estimator(spec, state, "cuda:0", precision)
    pred = est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(6, dtype=torch.long)), tfi, True, _keep_hidden=True)
    est.check_finite()
    conv = est.debug_fetch("conv")
    T = conv.shape[1]
    worst_conv = float((conv - inter["conv_out"]).abs().max())  # no attention mask in this family: every padded frame counts
    print(f"[conv0 mfma, group norm] {precision}: conv extractor output max abs {worst_conv:.3g}")
    assert worst_conv < 1e-3, worst_conv
    worst = 0.0
    for name, expected in ref.items():
        got = pred.outputs[name].cpu()
        for i, t in enumerate(ref_len.tolist()):
            worst = max(worst, float((got[:t, i] - expected[:t, i]).abs().max()))
    assert worst < 1e-3, worst
    est.close()
