"""The group-norm / post-LN wav2vec 2.0 variant (``feat_extract_norm="group"``, ``conv_bias=False``,
``do_stable_layer_norm=False``; wav2vec2-base / -large) through the C ABI: the reference builds whatever ``model_id`` names
(acoustic_model.py:775-826) and calls the model with ``attention_mask=None`` when the preprocessor has
``return_attention_mask=False`` (acoustic_model.py:814,842-846).

Against the goldens generated from the REAL reference (g11, g11b tiny; g12 wav2vec2-base shape) and against the CPU oracle
(pinned to the reference on the same variant by oracle/gen_golden.py) on fresh inputs.  Gate: logits / log-probs < 1e-3 on
valid frames, greedy alignments equal."""
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

    assert lib.load() is not None  # fails loudly if liballophant_amx.so is missing
    return estimator


def _batch(amd, audio, lengths):
    return amd.Batch(audio.cuda(), lengths, torch.zeros(len(lengths), dtype=torch.long))


@pytest.mark.parametrize("name", ["g11_tiny_groupnorm_postln", "g11b_tiny_groupnorm_masked"])
@pytest.mark.parametrize("precision", ["f16x3", "bf16x3"])
def test_tiny_variant_goldens(amd, name, precision):
    g = Golden(name)
    assert g.spec["feat_extract_norm"] == "group" and g.spec["stable_layer_norm"] is False
    est = amd.Estimator(g.spec, g.state_dict(), "cuda:0", precision)
    batch = _batch(amd, g.audio, g.lengths)
    pred = est.predict(batch, g.tfi, True, _keep_hidden=True)
    assert list(pred.outputs.keys()) == g.output_names
    assert torch.equal(pred.lengths.cpu(), g.frame_lengths)  # the downsampled lengths, with or without the attention mask
    for k in g.output_names:
        assert max_abs_valid_tm(pred.outputs[k].cpu(), g.logprobs(k), g.frame_lengths) < GATE, k
    assert max_abs_valid_bm(est.debug_fetch("conv"), g.conv_out(), g.frame_lengths) < GATE
    for i in g.hidden_indices():
        assert max_abs_valid_bm(est.debug_fetch("hidden", i), g.hidden(i), g.frame_lengths) < GATE, i
    # the default call (ragged batch: packed rows where the variant masks, OUTPUT_i classifiers on kept hidden states)
    plain = est.predict(batch, g.tfi, True)
    raw = est.predict(batch, g.tfi, log_probabilities=False)
    for k in g.output_names:
        assert max_abs_valid_tm(plain.outputs[k].cpu(), g.logprobs(k), g.frame_lengths) < GATE, k
        assert max_abs_valid_tm(raw.outputs[k].cpu(), g.logits(k), g.frame_lengths) < GATE, k
    if precision == "f16x3":
        decoded = est.greedy_decode(plain)
        for k in g.output_names:
            for i in range(len(g.lengths)):
                tokens, timesteps, score = g.tokens(k, i)
                got = decoded[k][i][0]
                assert torch.equal(got.tokens, tokens) and torch.equal(got.timesteps, timesteps), (k, i)
                assert abs(got.score - score) < 1e-2 * max(1.0, abs(score))
    est.close()


@pytest.mark.parametrize("precision", ["f16x3", "bf16x3"])
def test_wav2vec2_base_shape_golden(amd, precision):
    """wav2vec2-base shape (768 / 12 layers / 12 heads / 3072), 36 attribute heads + composed phoneme head + allophone
    pass-through, attention_mask=None: log-probs, logits, conv output and hidden states [0, 1, 6, 12] of the reference."""
    g = Golden("g12_w2v2base_multitask")
    est = amd.Estimator(g.spec, g.state_dict(), "cuda:0", precision)
    batch = _batch(amd, g.audio, g.lengths)
    pred = est.predict(batch, g.tfi, True, _keep_hidden=True)
    assert list(pred.outputs.keys()) == g.output_names and torch.equal(pred.lengths.cpu(), g.frame_lengths)
    worst = max(max_abs_valid_tm(pred.outputs[k].cpu(), g.logprobs(k), g.frame_lengths) for k in g.output_names)
    assert worst < GATE, worst
    assert max_abs_valid_bm(est.debug_fetch("conv")[:, :, ::8], g.conv_out(), g.frame_lengths) < GATE
    for i in g.hidden_indices():
        assert max_abs_valid_bm(est.debug_fetch("hidden", i)[:, :, ::8], g.hidden(i), g.frame_lengths) < GATE, i
    raw = est.predict(batch, g.tfi, log_probabilities=False)
    worst = max(max_abs_valid_tm(raw.outputs[k].cpu(), g.logits(k), g.frame_lengths) for k in g.output_names)
    assert worst < GATE, worst
    if precision == "f16x3":
        decoded = est.greedy_decode(est.predict(batch, g.tfi, True))
        for k in g.output_names:
            for i in range(len(g.lengths)):
                tokens, timesteps, _score = g.tokens(k, i)
                assert torch.equal(decoded[k][i][0].tokens, tokens) and torch.equal(decoded[k][i][0].timesteps, timesteps), (k, i)
    est.close()


def _check_against_oracle(amd, spec, seed, n, length, precisions=("f16x3",), no_pack_too=False):
    from oracle import allophant_oracle as O

    S.validate(spec)
    state = synthetic.make_state_dict(spec, seed=seed)
    composed = bool(spec.get("embedding_size"))
    tfi = synthetic.make_inventory(spec, 27, seed=seed) if composed else None
    offsets = synthetic.category_offsets(spec) if composed else None
    audio, lengths = synthetic.make_audio(n, length, seed=4000 + seed, ragged=True)
    ref, ref_len = O.predict(audio, lengths, state, spec, tfi, offsets)
    for precision in precisions:
        est = amd.Estimator(spec, state, "cuda:0", precision)
        calls = [False, True] if no_pack_too else [False]
        for no_pack in calls:
            pred = est.predict(_batch(amd, audio, lengths), tfi, True, _no_pack=no_pack)
            assert list(pred.outputs) == list(ref) and torch.equal(pred.lengths.cpu(), ref_len)
            worst = max(max_abs_valid_tm(pred.outputs[k].cpu(), ref[k], ref_len) for k in ref)
            assert worst < GATE, (precision, no_pack, worst)
            est.check_finite()
        if precision == "f16x3":
            decoded = est.greedy_decode(pred)
            for k in list(ref)[-3:]:
                for i, (tokens, timesteps, _score) in enumerate(O.greedy_ctc(ref[k].transpose(0, 1).contiguous(), ref_len)):
                    assert torch.equal(decoded[k][i][0].tokens, tokens) and torch.equal(decoded[k][i][0].timesteps, timesteps), (k, i)
        est.close()


@pytest.mark.parametrize("masked", [False, True])
def test_wav2vec2_base_shape_batch_against_oracle(amd, masked):
    """6 ragged utterances of up to 5 s at wav2vec2-base shape against the oracle: the large-batch kernels (ping-pong GEMM with
    GELU -> planes for the norm-free conv layers, the group-norm conv0 passes) in both mask modes; with the mask the ragged
    batch also takes the packed-row path, which must agree with the padded layout (AMX_FLAG_NO_PACK)."""
    enc = S.wav2vec2_base_encoder()
    enc["use_attention_mask"] = masked
    spec = S.multitask_spec(enc, allophone_layer=True)
    spec["shared_phones"] = 80
    _check_against_oracle(amd, spec, seed=31 + int(masked), n=6, length=80000, precisions=("f16x3", "bf16x3"), no_pack_too=masked)


def test_wav2vec2_large_shape_against_oracle(amd):
    """wav2vec2-large (1024 / 24 / 16 / 4096, group norm, post-LN, no attention mask): 24 post-LN layers deep."""
    enc = S.xlsr_300m_encoder()
    enc.update(feat_extract_norm="group", conv_bias=False, stable_layer_norm=False, use_attention_mask=False)
    spec = S.hierarchical_spec(enc, allophone_layer=False)
    _check_against_oracle(amd, spec, seed=41, n=3, length=48000)


@pytest.mark.parametrize("norm,bias,stable,masked", [("layer", True, False, True), ("group", True, True, True),
                                                      ("layer", False, True, False), ("group", False, True, False),
                                                      ("layer", False, False, False)])
def test_mixed_variants_tiny_against_oracle(amd, norm, bias, stable, masked):
    """`Wav2Vec2Config` lets the four switches vary independently: every combination the two released families do not cover,
    on the tiny shape with OUTPUT_i classifiers (the kept hidden states of either layer ordering)."""
    enc = S.tiny_encoder(3)
    enc.update(feat_extract_norm=norm, conv_bias=bias, stable_layer_norm=stable, use_attention_mask=masked)
    spec = S.multitask_spec(enc, ["syllabic", "long", "nasal"], embedding_size=16, train_phonemes=9, n_features=5)
    by_name = {c["name"]: c for c in spec["classes"]}
    by_name["syllabic"]["dependencies"] = ["OUTPUT_0"]
    by_name["long"]["dependencies"] = ["OUTPUT_2", "syllabic"]
    by_name["nasal"]["dependencies"] = ["OUTPUT_3"]
    _check_against_oracle(amd, spec, seed=51, n=5, length=7000, precisions=("f16x3", "bf16x3"), no_pack_too=masked)


def test_full_size_variant_properties(amd):
    """BASELINE config-2 geometry (32 x 10 s) on the wav2vec2-base variant: two utterances against the oracle run on the
    WHOLE padded batch is too slow, and without the attention mask an utterance's result depends on its padding -- so the
    check is (a) the first two utterances against the oracle on a 2-utterance batch padded to the same 10 s (the GroupNorm
    and the unmasked attention see the same frames), (b) probabilities normalise, (c) bitwise reproducibility."""
    from oracle import allophant_oracle as O

    spec = S.multitask_spec(S.wav2vec2_base_encoder(), allophone_layer=True)
    spec["shared_phones"] = 80
    state = synthetic.make_state_dict(spec, seed=0)
    tfi = synthetic.make_inventory(spec, 27, seed=0)
    audio, lengths = synthetic.make_audio(32, 160000, seed=1234, ragged=True)
    est = amd.Estimator(spec, state, "cuda:0", "f16x3")
    pred = est.predict(_batch(amd, audio, lengths), tfi)
    again = est.predict(_batch(amd, audio, lengths), tfi)
    assert pred.lengths.tolist() == S.frame_lengths(lengths.tolist(), spec)
    T = pred.outputs["phoneme"].shape[0]
    valid = (torch.arange(T).unsqueeze(1) < pred.lengths.unsqueeze(0)).cuda()
    for k, out in pred.outputs.items():
        assert torch.equal(out, again.outputs[k]), k
        assert torch.isfinite(out[valid]).all(), k
        assert (out.exp().sum(-1)[valid] - 1).abs().max().item() < 1e-4, k
    # no operator mixes utterances (GroupNorm is per utterance and channel, attention per utterance): rows 0-1 of the batch
    # equal the 2-utterance batch padded to the same length -- which the oracle can afford
    pick = [0, 1]
    sub_audio, sub_lengths = audio[pick].contiguous(), lengths[pick].clone()
    assert int(sub_lengths.max()) == audio.shape[1]  # utterance 0 has the full length: same padded length
    ref, ref_len = O.predict(sub_audio, sub_lengths, state, spec, tfi, synthetic.category_offsets(spec))
    worst = 0.0
    for k in ref:
        for j, i in enumerate(pick):
            t_i = int(ref_len[j])
            worst = max(worst, (pred.outputs[k][:t_i, i].cpu() - ref[k][:t_i, j]).abs().max().item())
    assert worst < GATE, worst
    est.close()


def test_unknown_norm_is_refused(amd):
    enc = S.tiny_encoder(1)
    enc["feat_extract_norm"] = "batch"
    spec = S.baseline_spec(enc, 5)
    with pytest.raises(ValueError):
        amd.Estimator(spec, synthetic.make_state_dict(S.baseline_spec(S.tiny_encoder(1), 5), seed=0), "cuda:0", "f16x3")
