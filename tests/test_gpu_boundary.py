"""Boundary behaviour of the HIP path through the C ABI (round-2 additions): the training-inventory default of restored
composition checkpoints, checkpoint key aliases, the reference-signature greedy decoder, inventory switching, utterances
shorter than the receptive field, batches beyond the 32-bit plane offsets, and the pinned prefetcher."""
import ctypes as C
import json
import os

import pytest
import torch

from allophant_amd import spec as S, synthetic
from golden_util import GOLDEN_DIR, max_abs_valid_tm

pytestmark = pytest.mark.gpu
GATE = 1e-3


@pytest.fixture(scope="module")
def amd():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from allophant_amd import estimator, lib

    assert lib.load() is not None  # fails loudly without the built library: there is no fallback
    return estimator


def _restricted_checkpoint():
    """A composition + allophone-layer checkpoint in the reference's schema whose indexer state is the one the REAL
    reference dumped for the synthetic table (tests/golden/g9_restricted_indexer.json)."""
    from allophant_amd.checkpoint import make_checkpoint

    with open(os.path.join(GOLDEN_DIR, "g9_restricted_indexer.json"), encoding="utf-8") as f:
        g9 = json.load(f)
    shared = g9["state"]["language_allophones"]["shared_phones"]
    n_features = len(g9["training_matrix"][0])
    spec = S.multitask_spec(S.tiny_encoder(2), ["syllabic", "long", "nasal"], embedding_size=16,
                            train_phonemes=len(g9["state"]["phoneme_inventory"]), n_features=n_features,
                            allophone_layer=True)
    spec["shared_phones"] = len(shared)
    offsets = g9["category_offsets"]
    spec["composition_categories"] = [b - a for a, b in zip(offsets, offsets[1:] + [g9["embedding_rows"]])]
    state = synthetic.make_state_dict(spec, seed=31)
    return make_checkpoint(spec, state, synthetic_encoder=True, indexer_state=g9["state"]), spec, state, g9


def test_restored_composition_checkpoint_predicts_with_the_training_inventory(amd):
    """`predict(batch)` without target_feature_indices falls back to the training inventory like upstream
    (acoustic_model.py:214-221; run.py:753): the matrix is rebuilt from the embedded table and must equal the one the real
    reference derived (golden g9), and the outputs must equal the oracle's with that matrix passed explicitly."""
    from oracle import allophant_oracle as O

    checkpoint, spec, state, g9 = _restricted_checkpoint()
    est, indexer = amd.Estimator.restore(checkpoint, "cuda:0")
    training = torch.tensor(g9["training_matrix"], dtype=torch.int64)
    assert torch.equal(est._training_inventory, training)
    assert indexer.phoneme_inventory("deu") == [] and indexer.phoneme_inventory("spa") == g9["inventories"]["spa"]
    audio, lengths = synthetic.make_audio(3, 9000, seed=4, ragged=True)
    batch = amd.Batch(audio.cuda(), lengths, torch.zeros(3, dtype=torch.long))
    pred = est.predict(batch)  # no target_feature_indices
    ref, ref_len = O.predict(audio, lengths, state, spec, training, synthetic.category_offsets(spec))
    assert list(pred.outputs) == list(ref) and torch.equal(pred.lengths.cpu(), ref_len)
    assert pred.outputs["phoneme"].shape[-1] == len(g9["shared_phones"]) + 1
    assert max(max_abs_valid_tm(pred.outputs[k].cpu(), ref[k], ref_len) for k in ref) < GATE
    # an explicit inventory still overrides, and None afterwards means the training inventory again (not "the last one")
    es = indexer.composition_feature_matrix(indexer.phoneme_inventory("spa"))
    pred_es = est.predict(batch, es)
    assert pred_es.outputs["phoneme"].shape[-1] == len(g9["inventories"]["spa"]) + 1
    again = est.predict(batch)
    assert torch.equal(again._flat, pred._flat)
    est.close()


def test_restore_takes_the_branch_of_a_real_checkpoint(amd, tmp_path):
    """The branch a real ``allophant.pt`` takes through ``Estimator.restore`` (reference estimator.py:229-249, 1085-1126): no
    private ``amx_*`` key anywhere -- the XLS-R-300m shape comes from ``nn.acoustic_model.model_id``, the embedding-table
    layout from the embedded attribute table, ``shared_phones`` from the shape of ``_allophone_matrices`` -- through a
    ``torch.save``d file, at full XLS-R shape, predicting with the restored training inventory against the oracle."""
    from oracle import allophant_oracle as O
    from allophant_amd.checkpoint import make_checkpoint

    with open(os.path.join(GOLDEN_DIR, "g9_restricted_indexer.json"), encoding="utf-8") as f:
        g9 = json.load(f)
    shared = g9["state"]["language_allophones"]["shared_phones"]
    phonemes = g9["state"]["phoneme_inventory"]
    spec = S.multitask_spec(S.xlsr_300m_encoder(), ["syllabic", "long", "nasal"], embedding_size=640, train_phonemes=len(phonemes),
                            n_features=len(g9["training_matrix"][0]), allophone_layer=True)
    spec["shared_phones"] = len(shared)
    offsets = g9["category_offsets"]
    spec["composition_categories"] = [b - a for a, b in zip(offsets, offsets[1:] + [g9["embedding_rows"]])]
    state = synthetic.make_state_dict(spec, seed=7)
    matrices = state["_projection._layers.phoneme._allophone_layer._allophone_matrices"]
    assert matrices.shape[1] == len(shared) + 1 and matrices.shape[2] == len(phonemes) + 1
    checkpoint = make_checkpoint(spec, state, synthetic_encoder=False, indexer_state=g9["state"])
    checkpoint["additional"] = {}  # what upstream stores there is user data; nothing of ours
    assert "amx" not in json.dumps({k: v for k, v in checkpoint.items() if k != "model_state"})
    assert checkpoint["config"]["nn"]["acoustic_model"]["model_id"] == "facebook/wav2vec2-xls-r-300m"
    path = tmp_path / "allophant.pt"
    torch.save(checkpoint, path)
    est, indexer = amd.Estimator.restore(str(path), "cuda:0")
    assert est._spec["hidden"] == 1024 and est._spec["layers"] == 24 and est._spec["shared_phones"] == len(shared)
    assert est._spec["composition_categories"] == spec["composition_categories"]
    training = torch.tensor(g9["training_matrix"], dtype=torch.int64)
    assert torch.equal(est._training_inventory, training)
    audio, lengths = synthetic.make_audio(2, 32000, seed=12, ragged=True)
    pred = est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(2, dtype=torch.long)))  # training inventory
    ref, ref_len = O.predict(audio, lengths, state, spec, training, synthetic.category_offsets(spec))
    assert list(pred.outputs) == list(ref) == ["syllabic", "long", "nasal", "phone", "phoneme"]
    assert torch.equal(pred.lengths.cpu(), ref_len)
    assert max(max_abs_valid_tm(pred.outputs[k].cpu(), ref[k], ref_len) for k in ref) < GATE
    es = indexer.composition_feature_matrix(indexer.phoneme_inventory("spa"))
    assert est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(2, dtype=torch.long)), es).outputs["phoneme"].shape[-1] == \
        len(g9["inventories"]["spa"]) + 1
    est.close()


def test_estimator_without_training_inventory_still_raises(amd):
    spec = S.multitask_spec(S.tiny_encoder(1), ["syllabic"], embedding_size=16, train_phonemes=5, n_features=3)
    est = amd.Estimator(spec, synthetic.make_state_dict(spec, seed=1), "cuda:0")
    audio, lengths = synthetic.make_audio(1, 4000, seed=1)
    with pytest.raises(ValueError, match="target_feature_indices"):
        est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(1, dtype=torch.long)))
    est.close()


def test_checkpoint_key_aliases(amd):
    """State dicts as older torch / transformers write them (`conv.weight_g` / `weight_v` instead of
    `parametrizations.weight.original0/1`) and with the `encoder._layers.*` alias keys that upstream's layer-slicing
    assignment leaves behind for graphs using OUTPUT_i restore to the same predictions."""
    from allophant_amd.checkpoint import make_checkpoint

    spec = S.hierarchical_spec(S.tiny_encoder(3), ["syllabic", "long"], embedding_size=16, train_phonemes=7, n_features=4)
    spec["classes"][0]["dependencies"] = ["OUTPUT_1"]
    state = synthetic.make_state_dict(spec, seed=8)
    tfi = synthetic.make_inventory(spec, 6, seed=8)
    audio, lengths = synthetic.make_audio(2, 8000, seed=8, ragged=True)
    batch = amd.Batch(audio.cuda(), lengths, torch.zeros(2, dtype=torch.long))
    est, _ = amd.Estimator.restore(make_checkpoint(spec, state, synthetic_encoder=True), "cuda:0")
    base = est.predict(batch, tfi)._flat.clone()
    est.close()
    aliased = {}
    pos = "_acoustic_model._model.encoder.pos_conv_embed.conv."
    for k, v in state.items():
        if k == pos + "parametrizations.weight.original0":
            aliased[pos + "weight_g"] = v
        elif k == pos + "parametrizations.weight.original1":
            aliased[pos + "weight_v"] = v
        else:
            aliased[k] = v
        if ".encoder.layers." in k:
            aliased[k.replace(".encoder.layers.", ".encoder._layers.")] = v  # duplicate alias, same storage upstream
    assert len(aliased) > len(state)
    est2, _ = amd.Estimator.restore(make_checkpoint(spec, aliased, synthetic_encoder=True), "cuda:0")
    assert torch.equal(est2.predict(batch, tfi)._flat, base)
    est2.close()


def test_reference_signature_greedy_decoder(amd):
    """`GreedyCTCDecoder()(outputs.transpose(1, 0), lengths)` as the reference's decode loop calls it (run.py:767-774,
    README.md:120-125): on the strided view (no copy), on the contiguous transpose, with a non-zero blank index; equal to
    the oracle decoder on the same emissions, bit for bit."""
    from oracle import allophant_oracle as O

    spec = S.multitask_spec(S.tiny_encoder(2), ["syllabic", "long", "nasal"], embedding_size=16, train_phonemes=9, n_features=5)
    est = amd.Estimator(spec, synthetic.make_state_dict(spec, seed=3), "cuda:0")
    tfi = synthetic.make_inventory(spec, 11, seed=3)
    audio, lengths = synthetic.make_audio(4, 30000, seed=3, ragged=True)
    pred = est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(4, dtype=torch.long)), tfi)
    decoder = amd.GreedyCTCDecoder()
    decoders = amd.feature_decoders(type("I", (), {"feature_names": list(pred.outputs)})())
    assert set(decoders) == set(pred.outputs)
    for name, out in pred.outputs.items():
        expected = O.greedy_ctc(out.cpu().transpose(0, 1).contiguous(), pred.lengths)
        for form in (out.transpose(1, 0), out.transpose(1, 0).contiguous()):
            hyps = decoder(form, pred.lengths)
            assert len(hyps) == 4 and all(len(h) == 1 for h in hyps)
            for (hyp,), (tokens, timesteps, score) in zip(hyps, expected):
                assert torch.equal(hyp.tokens, tokens) and torch.equal(hyp.timesteps, timesteps), name
                assert hyp.words == [] and abs(hyp.score - float(score)) < 1e-3 * max(1.0, abs(float(score)))
        # whole-prediction form decodes the same
        (first,), = est.greedy_decode(pred)[name][:1]
        assert torch.equal(first.tokens, expected[0][0])
    # a different blank index (constructor argument of the reference class)
    out = pred.outputs["phoneme"]
    blank = 2
    got = amd.GreedyCTCDecoder(blank)(out.transpose(1, 0), pred.lengths)
    want = O.greedy_ctc(out.cpu().transpose(0, 1).contiguous(), pred.lengths, blank)
    for (hyp,), (tokens, timesteps, _) in zip(got, want):
        assert torch.equal(hyp.tokens, tokens) and torch.equal(hyp.timesteps, timesteps)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        decoder(out.cpu().transpose(1, 0), pred.lengths)
    est.close()


def test_inventory_switching_keeps_earlier_predictions_decodable(amd):
    """The per-language loop of the reference (run.py:742-753) passes a different feature matrix per batch.  Every
    inventory keeps its own composed matrix and output tables, so (a) alternating inventories reproduces the same bits,
    (b) decoding Predictions made under an earlier inventory after a later predict() uses the right block layout."""
    from oracle import allophant_oracle as O

    spec = S.multitask_spec(S.tiny_encoder(2), ["syllabic", "long"], embedding_size=16, train_phonemes=9, n_features=5,
                            allophone_layer=True)
    spec["shared_phones"] = 12
    est = amd.Estimator(spec, synthetic.make_state_dict(spec, seed=6), "cuda:0")
    audio, lengths = synthetic.make_audio(3, 12000, seed=6, ragged=True)
    batch = amd.Batch(audio.cuda(), lengths, torch.zeros(3, dtype=torch.long))
    inventories = [synthetic.make_inventory(spec, p, seed=p) for p in (5, 13, 8)]
    first = [est.predict(batch, tfi) for tfi in inventories]
    for _ in range(2):
        for tfi, ref in zip(inventories, first):
            assert torch.equal(est.predict(batch, tfi)._flat, ref._flat)
    # the current inventory is the 8-phone one; decode the 5-phone predictions
    decoded = est.greedy_decode(first[0])
    for name, out in first[0].outputs.items():
        want = O.greedy_ctc(out.cpu().transpose(0, 1).contiguous(), first[0].lengths)
        for n in range(3):
            assert torch.equal(decoded[name][n][0].tokens, want[n][0]), name
    # more inventories than the library caches (16): the oldest are rebuilt on demand
    many = [synthetic.make_inventory(spec, 3 + (i % 7), seed=100 + i) for i in range(20)]
    outs = [est.predict(batch, tfi)._flat.clone() for tfi in many]
    for tfi, ref in zip(many, outs):
        assert torch.equal(est.predict(batch, tfi)._flat, ref)
    est.close()


def test_utterance_shorter_than_the_receptive_field_is_refused(amd):
    """A 200-sample utterance survives the first conv layers but not the sixth: upstream's floor-division length formula
    gives 0 frames (frontend.py:192-203), truncating division would have reported 1.  The C ABI refuses the batch."""
    from allophant_amd import utils

    spec = S.baseline_spec(S.tiny_encoder(1), 10)
    assert utils.downsampled_lengths(torch.tensor([200, 399, 400]), spec["conv_kernel"], spec["conv_stride"]).tolist() == [0, 0, 1]
    est = amd.Estimator(spec, synthetic.make_state_dict(spec, seed=2), "cuda:0")
    audio, lengths = synthetic.make_audio(2, 16000, seed=2)
    for short in (200, 399):
        lengths[1] = short
        audio[1, short:] = 0
        with pytest.raises(ValueError, match="receptive field"):
            est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(2, dtype=torch.long)))
    lengths[1] = 400
    pred = est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(2, dtype=torch.long)))
    assert pred.lengths.tolist() == [49, 1]
    est.close()


@pytest.mark.parametrize("precision,tolerance", [("f16x3", 2e-4), ("f16", 6e-2)])
def test_batches_beyond_the_32_bit_plane_offsets(amd, precision, tolerance):
    """24 x 60 s: one conv-0 output plane would be 4.7 GB, past the 32-bit plane offsets of the kernels -- in the two-plane
    mode and in the single-plane mode alike (the same limit applies to both).  The C ABI refuses such a call (AMX_EINVAL)
    and reports the limit; the facade runs the batch as slices and the results equal solo runs of the utterances (f16: the
    batch and the solo run take differently tiled kernels, so they agree within that mode's measured error)."""
    from allophant_amd import lib

    spec = S.multitask_spec(S.xlsr_300m_encoder(), allophone_layer=True)
    spec["shared_phones"] = 80
    state = synthetic.make_state_dict(spec, seed=0)
    est = amd.Estimator(spec, state, "cuda:0", precision)
    tfi = synthetic.make_inventory(spec, 27, seed=0)
    n, length = 24, 960000
    handle = lib.load()
    n_max = int(handle.amx_max_utterances(est._handle, length))
    assert 16 <= n_max < n
    audio, lengths = synthetic.make_audio(n, length, seed=17, ragged=True)
    dev = audio.cuda()
    # the raw C call with all 24 utterances is refused, not silently wrong
    est._set_inventory(tfi)
    flat = torch.empty(16, device="cuda")
    out_lengths = torch.empty(n, dtype=torch.int64)
    code = handle.amx_forward(est._handle, C.c_void_p(dev.data_ptr()), C.cast(lengths.data_ptr(), C.POINTER(C.c_int64)), n,
                              length, C.c_void_p(flat.data_ptr()), C.cast(out_lengths.data_ptr(), C.POINTER(C.c_int64)), 0, None)
    assert code == lib.AMX_EINVAL and b"split the batch" in handle.amx_last_error(est._handle)
    pred = est.predict(amd.Batch(dev, lengths, torch.zeros(n, dtype=torch.long)), tfi)
    assert pred.lengths.tolist() == S.frame_lengths(lengths.tolist(), spec)
    for i in (0, n_max - 1, n_max, n - 1):  # both sides of the slice boundary
        n_i, t_i = int(lengths[i]), int(pred.lengths[i])
        solo = est.predict(amd.Batch(dev[i:i + 1, :n_i].contiguous(), lengths[i:i + 1], torch.zeros(1, dtype=torch.long)), tfi)
        for k in ("phoneme", "stress", "click"):
            assert (pred.outputs[k][:t_i, i] - solo.outputs[k][:t_i, 0]).abs().max().item() < tolerance, (i, k)
    decoded = est.greedy_decode(pred)
    assert len(decoded["phoneme"]) == n
    est.close()


def test_prefetcher_matches_the_plain_loop(amd):
    """`Prefetcher` (pinned collation ring + host-to-device copies on a side stream, one batch ahead) yields the same
    predictions as collating and copying synchronously, also when the ring is shorter than the number of batches (slots
    are reused behind their copy events)."""
    from allophant_amd import batching

    spec = S.multitask_spec(S.tiny_encoder(2), ["syllabic", "long"], embedding_size=16, train_phonemes=9, n_features=5)
    est = amd.Estimator(spec, synthetic.make_state_dict(spec, seed=9), "cuda:0")
    tfi = synthetic.make_inventory(spec, 7, seed=9)
    g = torch.Generator().manual_seed(5)
    corpus = [torch.randn(int(n), generator=g) * 0.1 for n in torch.randint(2000, 40000, (37,), generator=g)]
    sizes = [a.numel() for a in corpus]
    batcher = batching.Batcher(120000, "frames")
    index_batches = list(batcher.index_batches(len(corpus), sizes, order=batching.length_sorted_order(sizes)))
    assert len(index_batches) > 4
    plain = []
    for indices in index_batches:
        b = batching.collate([corpus[i] for i in indices])
        plain.append(est.predict(b.to("cuda:0"), tfi)._flat.clone())
    collator = batching.PinnedCollator(max_samples=120000, depth=2)
    prefetcher = batching.Prefetcher(index_batches, torch.device("cuda:0"), fetch=lambda ix: collator([corpus[i] for i in ix]))
    assert prefetcher.copy_stream != torch.cuda.current_stream()
    count = 0
    for batch, expected in zip(prefetcher, plain):
        assert batch.audio_features.is_cuda
        assert torch.equal(est.predict(batch, tfi)._flat, expected)
        count += 1
    assert count == len(plain)
    est.close()


def test_decoded_gather_over_a_one_rank_rccl_group(amd):
    """SURVEY 8 f1 on the device: `greedy_decode_device` keeps the alignments in HBM and `parallel.gather_decoded` moves
    them through RCCL (a one-rank group here: the box has one GPU; the two-rank logic runs under gloo in
    tests/test_distributed_cpu.py).  The gathered hypotheses equal the oracle decoder on the same log-probabilities."""
    import socket

    import torch.distributed as dist

    from allophant_amd.parallel import gather_decoded
    from oracle import allophant_oracle as O

    spec = S.multitask_spec(S.tiny_encoder(2), ["syllabic", "long"], embedding_size=16, train_phonemes=9, n_features=5,
                            allophone_layer=True)
    spec["shared_phones"] = 11
    est = amd.Estimator(spec, synthetic.make_state_dict(spec, seed=8), "cuda:0")
    tfi = synthetic.make_inventory(spec, 11, seed=8)
    audio, lengths = synthetic.make_audio(3, 24000, seed=8, ragged=True)
    pred = est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(3, dtype=torch.long)), tfi)
    decoded = est.greedy_decode_device(pred)
    assert decoded.tokens.is_cuda and decoded.names == list(pred.outputs)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        got = gather_decoded(decoded, ["phoneme", "syllabic"], 3, torch.device("cuda", 0), dst=0)
    finally:
        if created:
            dist.destroy_process_group()
    assert list(got) == ["phoneme", "syllabic"]
    for name in got:
        want = O.greedy_ctc(pred.outputs[name].cpu().transpose(0, 1).contiguous(), pred.lengths)
        assert len(got[name]) == 3
        for (hyp,), (tokens, timesteps, score) in zip(got[name], want):
            assert torch.equal(hyp.tokens, tokens) and torch.equal(hyp.timesteps, timesteps), name
            assert abs(hyp.score - float(score)) < 1e-3 * max(1.0, abs(float(score)))
    est.close()


def test_c_abi_with_plain_host_buffers(amd):
    """The boundary is plain pointers and sizes: drive `liballophant_amx.so` through ctypes with numpy arrays only -- weights,
    audio, lengths, inventory and the output block are host memory (`AMX_FLAG_HOST_IO`: the library stages them over PCIe and
    returns synchronised), the stream is the null stream -- and compare with the torch-tensor façade on the same model."""
    import numpy as np

    from allophant_amd import lib as L
    from allophant_amd.estimator import _spec_to_structs

    spec = S.multitask_spec(S.tiny_encoder(2), ["syllabic", "long"], embedding_size=16, train_phonemes=9, n_features=5)
    state = synthetic.make_state_dict(spec, seed=12)
    tfi = synthetic.make_inventory(spec, 10, seed=12)
    audio, lengths = synthetic.make_audio(3, 20000, seed=12, ragged=True)
    est = amd.Estimator(spec, state, "cuda:0")
    want = est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(3, dtype=torch.long)), tfi)
    want_out = {k: v.cpu().numpy() for k, v in want.outputs.items()}
    est.close()

    lib = L.load()
    cfg, descs = _spec_to_structs(spec, "f16x3")
    arrays = {k: np.ascontiguousarray(v.detach().cpu().numpy(), dtype=np.float32) for k, v in state.items()}
    tensors = (L.AmxTensor * len(arrays))()
    for i, (name, a) in enumerate(arrays.items()):
        tensors[i].name = name.encode()
        tensors[i].data = a.ctypes.data_as(C.POINTER(C.c_float))
        tensors[i].numel = a.size
    handle = C.c_void_p()
    L.check(lib, None, lib.amx_create(C.byref(handle), 0, C.byref(cfg), descs, len(descs), tensors, len(arrays)))
    try:
        tfi_np = np.ascontiguousarray(tfi.numpy(), dtype=np.int64)
        offsets = np.cumsum([1] + list(spec["composition_categories"]), dtype=np.int64)[:-1].copy()
        L.check(lib, handle, lib.amx_set_inventory(handle, tfi_np.ctypes.data_as(C.POINTER(C.c_int64)), tfi_np.shape[0], tfi_np.shape[1],
                                                   offsets.ctypes.data_as(C.POINTER(C.c_int64)), None))
        audio_np = np.ascontiguousarray(audio.numpy(), dtype=np.float32)
        len_np = np.ascontiguousarray(lengths.numpy(), dtype=np.int64)
        n, l = audio_np.shape
        n_out, t, total = C.c_int(), C.c_int64(), C.c_int64()
        L.check(lib, handle, lib.amx_output_layout(handle, n, l, None, C.byref(n_out), C.byref(t), C.byref(total)))
        layout = (L.AmxOutputDesc * n_out.value)()
        L.check(lib, handle, lib.amx_output_layout(handle, n, l, layout, C.byref(n_out), C.byref(t), C.byref(total)))
        out = np.empty(total.value, dtype=np.float32)
        out_len = np.empty(n, dtype=np.int64)
        L.check(lib, handle, lib.amx_forward(handle, C.c_void_p(audio_np.ctypes.data), len_np.ctypes.data_as(C.POINTER(C.c_int64)), n, l,
                                             C.c_void_p(out.ctypes.data), out_len.ctypes.data_as(C.POINTER(C.c_int64)),
                                             L.FLAG_HOST_IO, None))
        assert out_len.tolist() == want.lengths.tolist()
        for d in layout:
            name = d.name.decode()
            got = out[d.offset: d.offset + t.value * n * d.classes].reshape(t.value, n, d.classes)
            for i, f in enumerate(out_len.tolist()):
                assert np.array_equal(got[:f, i], want_out[name][:f, i]), name
    finally:
        lib.amx_destroy(handle)


def test_small_entry_points(amd):
    """`amx_device_bytes` (weights, then weights + workspace), `amx_synchronize`, and the ABI-version check of `amx_create`."""
    from allophant_amd import lib as L
    from allophant_amd.estimator import _spec_to_structs

    spec = S.baseline_spec(S.tiny_encoder(1), 6)
    state = synthetic.make_state_dict(spec, seed=2)
    est = amd.Estimator(spec, state, "cuda:0")
    weights_only = est.device_bytes
    assert weights_only > 0
    audio, lengths = synthetic.make_audio(2, 8000, seed=2)
    est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(2, dtype=torch.long)))
    est.synchronize()
    assert est.device_bytes > weights_only  # the workspace of the first geometry
    est.close()
    # a binding built against another header is refused before anything is allocated
    lib = L.load()
    cfg, descs = _spec_to_structs(spec, "f16x3")
    cfg.abi_version = L.AMX_ABI_VERSION + 1
    handle = C.c_void_p()
    code = lib.amx_create(C.byref(handle), 0, C.byref(cfg), descs, len(descs), (L.AmxTensor * 1)(), 0)
    assert code != 0 and not handle.value
    with pytest.raises(ValueError, match="ABI version"):
        L.check(lib, None, code)


def test_c_abi_gather_over_a_one_rank_rccl_communicator(amd):
    """`amx_gather_outputs`: the exchange step of data parallelism behind the C ABI, for hosts that do not go through
    torch.distributed.  A communicator of ONE rank (all this box has) created with the RCCL under /opt/rocm through ctypes, the
    way a C host would with ncclCommInitRank: the root's receive buffers must hold this rank's output block and frame lengths
    bit for bit, and the [T, N, C] views of the received block at the offsets of amx_output_layout must equal the prediction."""
    import ctypes as C
    import os

    path = "/opt/rocm/lib/librccl.so"
    if not os.path.exists(path):
        pytest.skip("no RCCL under /opt/rocm")
    os.environ["AMX_RCCL_LIBRARY"] = path  # the library resolves ncclSend / ncclRecv in the RCCL the communicator comes from
    rccl = C.CDLL(path, mode=C.RTLD_GLOBAL)

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]

    uid = UniqueId()
    rccl.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    torch.cuda.set_device(0)
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0

    from allophant_amd import lib as L

    handle = L.load()
    spec = S.multitask_spec(S.tiny_encoder(2), ["syllabic", "long"], embedding_size=16, train_phonemes=9, n_features=5)
    state = synthetic.make_state_dict(spec, seed=3)
    est = amd.Estimator(spec, state, "cuda:0", "f16x3")
    audio, lengths = synthetic.make_audio(3, 5000, seed=5, ragged=True)
    tfi = synthetic.make_inventory(spec, 6, seed=3)
    pred = est.predict(amd.Batch(audio.cuda(), lengths, torch.zeros(3, dtype=torch.long)), tfi)
    flat = pred._flat
    frame_lengths = pred.lengths.to("cuda:0")
    received = torch.full_like(flat, float("nan"))
    received_lengths = torch.full_like(frame_lengths, -1)
    stream = torch.cuda.current_stream().cuda_stream
    code = handle.amx_gather_outputs(comm, 0, 1, 0, C.c_void_p(flat.data_ptr()), flat.numel(), C.c_void_p(received.data_ptr()),
                                     C.c_void_p(frame_lengths.data_ptr()), 3, C.c_void_p(received_lengths.data_ptr()), C.c_void_p(stream))
    assert code == 0, handle.amx_dist_last_error().decode()
    torch.cuda.synchronize()
    assert torch.equal(received, flat) and torch.equal(received_lengths, frame_lengths)
    first = flat.data_ptr()
    for name, out in pred.outputs.items():
        offset = (out.data_ptr() - first) // 4
        t, n, c = out.shape
        assert torch.equal(received[offset: offset + t * n * c].view(t, n, c), out), name
    # argument errors are refused before anything is enqueued
    assert handle.amx_gather_outputs(comm, 0, 1, 0, None, flat.numel(), None, None, 3, None, C.c_void_p(stream)) == L.AMX_EINVAL
    assert handle.amx_gather_outputs(None, 0, 1, 0, C.c_void_p(flat.data_ptr()), 1, C.c_void_p(received.data_ptr()), None, 0, None,
                                     C.c_void_p(stream)) == L.AMX_EINVAL
    assert rccl.ncclCommDestroy(comm) == 0
    est.close()
