"""Head dimensions other than 64 (round-5 review, item 5).  The reference builds whatever ``model_id`` names
(``/root/reference/allophant/network/acoustic_model.py:796-826``): XLS-R 1B / 2B have head dimensions 80 / 120.  On the device the
Q / K / V rows are padded to 64 or 128 columns (zeros) and the first attention kernel runs an instance templated on that width;
only the real columns of the output leave.

Against goldens the REAL reference produced (g13: hidden 160 / 2 heads = 80; g13b: hidden 64 / 2 heads = 32, post-LN) and against
the CPU oracle for 96 / 120 / 128 / 40 / 8, padded and packed rows, all four arithmetic modes."""
import pytest
import torch

from allophant_amd import spec as S, synthetic
from golden_util import Golden, max_abs_valid_bm, max_abs_valid_tm

pytestmark = pytest.mark.gpu
GATE = 1e-3


@pytest.fixture(scope="module")
def amd():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from allophant_amd import estimator, lib

    assert lib.load() is not None
    return estimator


@pytest.mark.parametrize("name", ["g13_tiny_head_dim_80", "g13b_tiny_head_dim_32"])
@pytest.mark.parametrize("precision", ["f16x3", "bf16x3"])
def test_reference_goldens(amd, name, precision):
    g = Golden(name)
    est = amd.Estimator(g.spec, g.state_dict(), "cuda:0", precision)
    batch = amd.Batch(g.audio.cuda(), g.lengths, torch.zeros(len(g.lengths), dtype=torch.long))
    pred = est.predict(batch, g.tfi, True, _keep_hidden=True)
    assert list(pred.outputs.keys()) == g.output_names
    assert torch.equal(pred.lengths.cpu(), g.frame_lengths)
    for k in g.output_names:
        assert max_abs_valid_tm(pred.outputs[k].cpu(), g.logprobs(k), g.frame_lengths) < GATE, k
    for i in g.hidden_indices():
        assert max_abs_valid_bm(est.debug_fetch("hidden", i), g.hidden(i), g.frame_lengths) < GATE, i
    # the same batch on packed rows (the default for a ragged batch) and the reference decoder's alignments
    pred = est.predict(batch, g.tfi, True)
    for k in g.output_names:
        assert max_abs_valid_tm(pred.outputs[k].cpu(), g.logprobs(k), g.frame_lengths) < GATE, k
    decoded = est.greedy_decode(pred)
    for k in g.output_names:
        for i in range(len(g.lengths)):
            tokens, timesteps, _ = g.tokens(k, i)
            assert torch.equal(decoded[k][i][0].tokens, tokens) and torch.equal(decoded[k][i][0].timesteps, timesteps), (k, i)
    est.close()


@pytest.mark.parametrize("hidden,heads,groups", [(192, 2, 4), (240, 2, 5), (256, 2, 4), (384, 3, 8), (80, 2, 2), (64, 8, 4), (960, 8, 20)])
def test_head_dims_against_oracle(amd, hidden, heads, groups):
    """head_dim 96, 120 (XLS-R 2B's), 128, 128 with three heads, 40, 8 and 120 at eight heads (hidden 960: the products of the layers
    run on the ping-pong kernel): 5 ragged utterances, packed and padded rows, f16x3 and bf16x3 within 1e-3; the single-plane modes
    within their measured bounds."""
    from oracle import allophant_oracle as O

    enc = S.tiny_encoder(2)
    enc.update(hidden=hidden, heads=heads, ffn=2 * hidden, pos_groups=groups)
    spec = S.multitask_spec(enc, ["syllabic", "long"], embedding_size=16, train_phonemes=9, n_features=5, allophone_layer=True)
    spec["shared_phones"] = 11
    S.validate(spec)
    state = synthetic.make_state_dict(spec, seed=hidden + heads)
    tfi = synthetic.make_inventory(spec, 7, seed=3)
    audio, lengths = synthetic.make_audio(5, 24000, seed=hidden, ragged=True)
    ref, ref_len = O.predict(audio, lengths, state, spec, tfi, synthetic.category_offsets(spec))
    batch = amd.Batch(audio.cuda(), lengths, torch.zeros(5, dtype=torch.long))
    for precision, gate in (("f16x3", GATE), ("bf16x3", GATE), ("f16", 6e-2), ("bf16", 5e-1)):
        est = amd.Estimator(spec, state, "cuda:0", precision)
        for no_pack in (False, True):
            pred = est.predict(batch, tfi, True, _no_pack=no_pack)
            assert torch.equal(pred.lengths.cpu(), ref_len)
            worst = max(max_abs_valid_tm(pred.outputs[k].cpu(), ref[k], ref_len) for k in ref)
            assert worst < gate, (precision, no_pack, worst)
        est.check_finite()
        est.close()


def test_heads_wider_than_128_are_refused(amd):
    enc = S.tiny_encoder(1)
    enc.update(hidden=272, heads=2, ffn=256)  # head_dim 136
    spec = S.baseline_spec(enc, 5)
    with pytest.raises(ValueError, match="head_dim"):
        S.validate(spec)
    from allophant_amd import lib as L
    from allophant_amd.estimator import _spec_to_structs
    import ctypes as C

    cfg, descs = _spec_to_structs(spec, "f16x3")
    handle = C.c_void_p()
    code = L.load().amx_create(C.byref(handle), 0, C.byref(cfg), descs, len(descs), (L.AmxTensor * 1)(), 0)
    assert code == L.AMX_EINVAL and b"head_dim" in L.load().amx_last_error(None)


@pytest.mark.parametrize("precision", ["f16x3", "bf16x3"])
def test_xlsr_1b_width_golden(amd, precision):
    """Golden g14 from the real reference: hidden 1280 (rows wider than 1024: the second instance of the row kernels), 16 heads of
    80, the positional convolution with 80 channels per group (the grouped implicit GEMM: the window kernel is the 64-channel
    form), composition head -- the WIDTH of ``facebook/wav2vec2-xls-r-1b`` on two layers."""
    g = Golden("g14_xlsr1b_width")
    est = amd.Estimator(g.spec, g.state_dict(), "cuda:0", precision)
    batch = amd.Batch(g.audio.cuda(), g.lengths, torch.zeros(len(g.lengths), dtype=torch.long))
    for no_pack in (True, False):
        pred = est.predict(batch, g.tfi, True, _no_pack=no_pack)
        assert list(pred.outputs.keys()) == g.output_names and torch.equal(pred.lengths.cpu(), g.frame_lengths)
        worst = max(max_abs_valid_tm(pred.outputs[k].cpu(), g.logprobs(k), g.frame_lengths) for k in g.output_names)
        assert worst < GATE, (no_pack, worst)
    pred = est.predict(batch, g.tfi, True, _keep_hidden=True)
    for i in g.hidden_indices():
        assert max_abs_valid_bm(est.debug_fetch("hidden", i)[:, :, ::8], g.hidden(i), g.frame_lengths) < GATE, i
    est.close()


@pytest.mark.parametrize("hidden,heads,ffn", [(1280, 16, 5120), (1920, 16, 7680)])
def test_xlsr_1b_2b_shapes_large_batch_against_oracle(amd, hidden, heads, ffn):
    """The XLS-R 1B / 2B layer shapes (``spec.xlsr_1b_encoder`` / ``xlsr_2b_encoder``, three layers of them) on 24 x 10 s: the
    products run on the ping-pong kernel and the layers take the LayerNorm fold with 20 / 30 column blocks per row
    (``amx_pass_info``); utterances 0 and 13 against the CPU oracle on each of them alone."""
    from oracle import allophant_oracle as O

    enc = S.xlsr_1b_encoder() if hidden == 1280 else S.xlsr_2b_encoder()
    assert (enc["hidden"], enc["heads"], enc["ffn"]) == (hidden, heads, ffn)
    enc["layers"] = 3
    spec = S.multitask_spec(enc, ["syllabic", "long", "nasal"], embedding_size=64, train_phonemes=20, n_features=8, allophone_layer=True)
    spec["shared_phones"] = 24
    state = synthetic.make_state_dict(spec, seed=hidden)
    tfi = synthetic.make_inventory(spec, 15, seed=2)
    audio, lengths = synthetic.make_audio(24, 160000, seed=hidden + 1)
    est = amd.Estimator(spec, state, "cuda:0", "f16x3")
    pred = est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(24, dtype=torch.long)), tfi, True)
    info = est.pass_info()
    assert info["ln_fold"] == 1, info
    est.check_finite()
    offsets = synthetic.category_offsets(spec)
    for i in (0, 13):
        ref, ref_len = O.predict(audio[i:i + 1].contiguous(), lengths[i:i + 1], state, spec, tfi, offsets)
        t_i = int(ref_len[0])
        worst = max((pred.outputs[k][:t_i, i].cpu() - ref[k][:t_i, 0]).abs().max().item() for k in ref)
        assert worst < GATE, (i, worst)
    est.close()
