"""Host-side logic and the C-ABI surface (no compute calls: there is no GPU here)."""
import ctypes as C
import os
import re

import pytest
import torch

from allophant_amd import checkpoint, lib, spec as S, synthetic
from allophant_amd.estimator import Batch, Predictions, _spec_to_structs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    handle = lib.load()
    header = open(os.path.join(ROOT, "include", "allophant_amx.h")).read()
    declared = set(re.findall(r"\b(amx_[a-z_]+)\s*\(", header))
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    for symbol in declared:
        assert hasattr(handle, symbol), symbol


def test_struct_layout_matches_header():
    # sizes implied by include/allophant_amx.h (all members are 4-byte aligned scalars / arrays)
    assert C.sizeof(lib.AmxConfig) == 4 * (3 + 8 + 8 + 6 + 1 + 5 + 4)  # + the four variant fields of ABI 4
    assert C.sizeof(lib.AmxClassDesc) == 48 + 4 * 3 + 4 * 64 + 4 * 2  # + time_heads, time_positional (ABI 2)
    assert C.sizeof(lib.AmxOutputDesc) == 48 + 4 + 4 + 8  # int32 + padding + int64
    assert C.sizeof(lib.AmxTensor) == 24


def test_create_without_gpu_fails_loudly():
    """No CPU fallback: creating a model on a box without a HIP device is an error, never a silent slow path."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    handle = lib.load()
    spec = S.baseline_spec(S.tiny_encoder(1), 5)
    cfg, descs = _spec_to_structs(spec, "f16x3")
    out = C.c_void_p()
    tensors = (lib.AmxTensor * 1)()
    tensors[0].name = b"x"
    code = handle.amx_create(C.byref(out), 0, C.byref(cfg), descs, len(descs), tensors, 0)
    assert code != 0 and not out.value
    message = handle.amx_last_error(None).decode()
    assert "no CPU fallback" in message or "HIP" in message


def test_spec_to_structs_dependencies():
    spec = S.hierarchical_spec(S.tiny_encoder(2), ["syllabic", "long"], embedding_size=16, train_phonemes=9, n_features=5)
    spec["classes"][0]["dependencies"] = ["OUTPUT_1"]
    cfg, descs = _spec_to_structs(spec, "bf16")
    assert cfg.precision == 0 and cfg.embedding_size == 16 and cfg.n_conv == 7
    assert descs[0].deps[0] == lib.dep_output_layer(1) == -3
    assert list(descs[2].deps[:3]) == [lib.DEP_OUTPUT, 0, 1]
    assert descs[2].out_features == 16 and descs[0].out_features == 4
    with pytest.raises(ValueError):
        _spec_to_structs(spec, "fp8")


def test_validate_mirrors_reference_errors():
    enc = S.tiny_encoder(1)
    with pytest.raises(ValueError, match="duplicate"):
        S.validate(dict(enc, classes=[{"name": "a", "size": 2, "dependencies": ["OUTPUT"]}] * 2))
    with pytest.raises(ValueError, match="reserved"):
        S.validate(dict(enc, classes=[{"name": "OUTPUT_1", "size": 2, "dependencies": ["OUTPUT"]}]))
    with pytest.raises(ValueError, match="requires a dependency"):
        S.validate(dict(enc, classes=[{"name": "a", "size": 2, "dependencies": []}]))
    with pytest.raises(ValueError, match="requires 'OUTPUT'"):
        S.validate(dict(enc, classes=[{"name": "a", "size": 2, "dependencies": ["b"]},
                                      {"name": "b", "size": 2, "dependencies": ["a"]}]))
    with pytest.raises(ValueError, match="cycle"):
        S.validate(dict(enc, classes=[{"name": "a", "size": 2, "dependencies": ["b", "OUTPUT"]},
                                      {"name": "b", "size": 2, "dependencies": ["a"]}]))


def test_output_names_follow_reference_order():
    spec = S.multitask_spec(S.tiny_encoder(1), ["x", "y"], embedding_size=16, allophone_layer=True)
    assert S.output_names(spec) == ["x", "y", "phone", "phoneme"]
    spec["classes"] = [
        {"name": "phoneme", "size": 9, "dependencies": ["OUTPUT", "x", "y"]},
        {"name": "x", "size": 3, "dependencies": ["OUTPUT_0"]},
        {"name": "y", "size": 2, "dependencies": ["x", "OUTPUT"]},
    ]
    assert S.output_names(spec) == ["x", "y", "phone", "phoneme"]


def test_synthetic_state_dict_layout_and_determinism():
    spec = S.multitask_spec(S.xlsr_300m_encoder(), allophone_layer=True)
    spec["layers"] = 1  # keep it small
    sd = synthetic.make_state_dict(spec, seed=0)
    assert sd["_acoustic_model._model.feature_extractor.conv_layers.0.conv.weight"].shape == (512, 1, 10)
    assert sd["_acoustic_model._model.encoder.pos_conv_embed.conv.parametrizations.weight.original1"].shape == (1024, 64, 128)
    assert sd["_projection._layers.phoneme._time_distributed_layer.weight"].shape == (640, 1024)
    assert sd["_projection._layers.phoneme._composition_layer._attribute_embeddings.weight"].shape == (112, 640)
    assert sd["_projection._layers.stress._time_distributed_layer.weight"].shape == (4, 1024)
    again = synthetic.make_state_dict(spec, seed=0)
    assert all(torch.equal(sd[k], again[k]) for k in sd)
    other = synthetic.make_state_dict(spec, seed=1)
    assert not torch.equal(sd["_acoustic_model._model.encoder.layer_norm.weight"],
                           other["_acoustic_model._model.encoder.layer_norm.weight"])
    tfi = synthetic.make_inventory(spec, 27)
    assert tfi.shape == (27, 37) and tfi.dtype == torch.int64 and int(tfi.max()) <= 2 and int(tfi.min()) >= 0
    assert synthetic.category_offsets(spec).tolist()[:3] == [1, 4, 7]


def test_checkpoint_schema_round_trip(tmp_path):
    spec = S.multitask_spec(S.tiny_encoder(2), ["syllabic", "long"], embedding_size=16, train_phonemes=9, n_features=5,
                            allophone_layer=True)
    spec["shared_phones"] = 11
    sd = synthetic.make_state_dict(spec, seed=3)
    ckpt = checkpoint.make_checkpoint(spec, sd, synthetic_encoder=True)
    for field in ("config", "allophant_version", "feature_size", "sample_rate", "attribute_graph", "epoch",
                  "phonetic_indexer_state", "dataset_meta_data", "model_state", "additional", "history",
                  "optimization_states"):
        assert field in ckpt  # reference Checkpoint fields (estimator.py:208-219)
    path = tmp_path / "allophant.pt"
    torch.save(ckpt, path)
    loaded = torch.load(path, map_location="cpu", weights_only=True)
    restored = checkpoint.spec_from_checkpoint(loaded)
    for key in ("classes", "dependency_blanks", "embedding_size", "allophone_layer", "hidden", "layers", "shared_phones",
                "composition_categories"):
        assert restored[key] == spec[key], key
    bad = dict(ckpt, sample_rate=8000)
    with pytest.raises(ValueError, match="sampling rate"):
        checkpoint.spec_from_checkpoint(bad)
    # a real-model checkpoint (no encoder override) resolves to the XLS-R-300m shape
    plain = checkpoint.make_checkpoint(S.baseline_spec(S.xlsr_300m_encoder(), 40), {}, synthetic_encoder=False)
    assert checkpoint.spec_from_checkpoint(plain)["hidden"] == 1024


def test_encoder_table_is_cross_checked_against_the_weights():
    """`MODEL_ID_ENCODERS` is written from memory: an entry that disagrees with the checkpoint's own tensors (conv biases,
    conv norms, layer count, widths) is refused instead of silently dropping weights."""
    encoder = S.tiny_encoder(1)
    spec = S.baseline_spec(encoder, 7)
    sd = synthetic.make_state_dict(spec, seed=1)
    good = checkpoint.make_checkpoint(spec, sd, synthetic_encoder=True)
    assert checkpoint.spec_from_checkpoint(good).get("conv_bias", True)
    for wrong, needle in (({"conv_bias": False}, "conv_bias"), ({"feat_extract_norm": "group"}, "feat_extract_norm"),
                          ({"layers": 3}, "encoder layers"), ({"ffn": 512}, "hidden / ffn")):
        bad = checkpoint.make_checkpoint(spec, sd, synthetic_encoder=True)
        bad["additional"]["amx_encoder"] = dict(bad["additional"]["amx_encoder"], **wrong)
        with pytest.raises(ValueError, match=needle):
            checkpoint.spec_from_checkpoint(bad)
    # the group-norm variant's own state dict passes under its own description and fails under the XLS-R one
    variant = dict(encoder, feat_extract_norm="group", conv_bias=False, stable_layer_norm=False, use_attention_mask=False)
    vspec = S.baseline_spec(variant, 7)
    vsd = synthetic.make_state_dict(vspec, seed=1)
    checkpoint.spec_from_checkpoint(checkpoint.make_checkpoint(vspec, vsd, synthetic_encoder=True))
    mixed = checkpoint.make_checkpoint(spec, vsd, synthetic_encoder=True)
    with pytest.raises(ValueError, match="does not match the checkpoint's weights"):
        checkpoint.spec_from_checkpoint(mixed)


def test_adapter_checkpoints_are_accepted_and_their_weights_ignored():
    """wav2vec 2.0 adapters (`add_adapter`; round-5 review, "missing" item 6).  The reference builds them with the HF model and reads
    nothing of their output (`.hidden_states` = encoder outputs, ``acoustic_model.py:839-853``; golden g15 from the real reference):
    a checkpoint that owns adapter weights resolves to the same device configuration as one without, the weights are not in the
    descriptors that go to ``amx_create``, and the one adapter configuration the reference itself cannot run is refused with its
    reason."""
    from allophant_amd.estimator import _spec_to_structs

    plain = S.baseline_spec(S.tiny_encoder(1), 7)
    adapted = S.baseline_spec(dict(S.tiny_encoder(1), add_adapter=True, num_adapter_layers=2), 7)
    S.validate(adapted)
    sd = synthetic.make_state_dict(adapted, seed=1)
    adapter_keys = [k for k in sd if ".adapter." in k]
    assert len(adapter_keys) == 4 and tuple(sd[synthetic.AM + "adapter.layers.1.conv.weight"].shape) == (2 * 128, 128, 3)
    assert {k: v for k, v in sd.items() if k not in adapter_keys}.keys() == synthetic.make_state_dict(plain, seed=1).keys()
    cfg_a, descs_a = _spec_to_structs(adapted, "f16x3")
    cfg_p, descs_p = _spec_to_structs(plain, "f16x3")
    assert bytes(cfg_a) == bytes(cfg_p) and len(descs_a) == len(descs_p)
    # a checkpoint described WITHOUT the flag whose state dict holds an adapter: detected from the weights
    ckpt = checkpoint.make_checkpoint(plain, sd, synthetic_encoder=True)
    assert checkpoint.spec_from_checkpoint(ckpt)["add_adapter"] is True
    # an adapter that changes the width: the reference sizes its classifiers for `output_hidden_size` and feeds them encoder states
    with pytest.raises(ValueError, match="output_hidden_size"):
        S.validate(dict(adapted, output_hidden_size=48))
    sd["_acoustic_model._model.adapter.proj.weight"] = torch.zeros(48, 128)
    with pytest.raises(ValueError, match="adapter projects to 48"):
        checkpoint.spec_from_checkpoint(checkpoint.make_checkpoint(plain, sd, synthetic_encoder=True))


def test_time_layer_classifiers_round_trip_and_struct_fields():
    """`time_layer` (MultiheadAttentionConfig, config.py:596-610) survives checkpoint write/read, reaches the C ABI
    structs, and num_heads must divide the classifier width like nn.MultiheadAttention asserts."""
    from golden_util import Golden

    spec = Golden("g8_tiny_time_layer").spec
    layers = {c["name"]: c.get("time_layer") for c in spec["classes"]}
    assert layers["long"] == {"num_heads": 2, "positional_embeddings": True} and layers["syllabic"] is None
    sd = synthetic.make_state_dict(spec, seed=8)
    prefix = "_projection._layers.long._time_distributed_layer."
    for leaf in ("input_projection.weight", "layer_norm.bias", "attention.in_proj_weight", "attention.out_proj.bias"):
        assert prefix + leaf in sd  # module tree of ProjectingMultiheadAttention (acoustic_model.py:237-253)
    restored = checkpoint.spec_from_checkpoint(checkpoint.make_checkpoint(spec, sd, synthetic_encoder=True))
    assert restored["classes"] == spec["classes"]
    _, descs = _spec_to_structs(restored, "f16x3")
    by_name = {d.name.decode(): d for d in descs}
    assert (by_name["long"].time_heads, by_name["long"].time_positional) == (2, 1)
    assert (by_name["nasal"].time_heads, by_name["nasal"].time_positional) == (1, 0)
    assert by_name["syllabic"].time_heads == 0
    bad = S.multitask_spec(S.tiny_encoder(1), ["long"], embedding_size=0, train_phonemes=5, n_features=3)
    bad["classes"][0]["time_layer"] = {"num_heads": 5, "positional_embeddings": False}
    with pytest.raises(ValueError, match="divisible"):
        S.validate(bad)


def test_batch_and_predictions_containers():
    b = Batch(torch.zeros(3, 10), torch.tensor([10, 4, 7]), torch.zeros(3, dtype=torch.long))
    assert len(b) == 3 and b.size() == 3 and "Features" in repr(b)
    moved = b.to("cpu")
    assert isinstance(moved, Batch) and moved.audio_features.shape == (3, 10)
    p = Predictions({"a": torch.zeros(2, 3, 4)}, torch.tensor([2, 1, 2]))
    assert len(p) == 3 and p.task_count() == 1


def test_mask_and_length_helpers_match_reference_goldens():
    """``mask_sequence`` / ``downsampled_lengths`` (reference utils.py:45-76, acoustic_model.py:832-835) against the integer
    goldens recorded from the reference (g4) and the spec-level frame formula."""
    import numpy as np

    from allophant_amd import utils

    z = np.load(os.path.join(ROOT, "tests", "golden", "g4_integer.npz"))
    spec = S.xlsr_300m_encoder()
    samples = torch.from_numpy(z["lengths_in"])
    frames = utils.downsampled_lengths(samples, spec["conv_kernel"], spec["conv_stride"])
    assert torch.equal(frames, torch.from_numpy(z["lengths_out"]))
    assert frames.tolist() == S.frame_lengths(samples.tolist(), spec)
    recorded = torch.from_numpy(z["mask_lengths"])
    assert torch.equal(utils.mask_sequence(recorded), torch.from_numpy(z["mask"]).bool())  # the reference's own mask
    lengths = torch.tensor([5, 2, 7])
    mask = utils.mask_sequence(lengths)
    assert mask.shape == (3, 7) and mask.sum(1).tolist() == [5, 2, 7] and bool(mask[1, 1]) and not bool(mask[1, 2])
    assert torch.equal(utils.mask_sequence(lengths, inverse=True), ~mask)
    assert torch.equal(utils.mask_sequence(lengths, batch_first=False), mask.t())
    assert utils.mask_sequence(lengths, max_length=4).shape == (3, 4)
    assert utils.mask_sequence(lengths, max_length=9, start=2).shape == (3, 7)


def test_bench_helpers_hash_gate_and_core_count(tmp_path, monkeypatch):
    """bench.py hygiene (round-2 review, item 8): `roofline.traffic` is only reported from a PMC file measured on THESE kernel
    sources; the CPU baseline knows physical cores from logical CPUs."""
    import json
    import os

    import bench

    h = bench.kernel_source_hash()
    assert len(h) == 16 and int(h, 16) >= 0 and h == bench.kernel_source_hash()
    cores, logical = bench.physical_cores()
    assert 1 <= cores <= logical
    profiles = tmp_path / "profiles"
    profiles.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: "feedfacefeedface")
    monkeypatch.setattr(bench, "TRAFFIC_FILES", ("new.json", "old.json"))
    (profiles / "old.json").write_text(json.dumps({"f16x3": {"hbm_bytes_per_launch": 1.0}}))  # no hash: a round-2 file
    data, why = bench.load_traffic("f16x3")
    assert data is None and "traffic not reported" in why and "None" in why
    (profiles / "new.json").write_text(json.dumps({"kernel_source_hash": "0123456789abcdef", "f16x3": {"hbm_bytes_per_launch": 2.0}}))
    data, why = bench.load_traffic("f16x3")
    assert data is None and "0123456789abcdef" in why
    (profiles / "new.json").write_text(json.dumps({"kernel_source_hash": "feedfacefeedface", "f16x3": {"hbm_bytes_per_launch": 3.0}}))
    data, why = bench.load_traffic("f16x3")
    assert data == {"hbm_bytes_per_launch": 3.0} and "new.json" in why and "feedfacefeedface" in why
    assert bench.load_traffic("bf16") == (None, None) or bench.load_traffic("bf16")[0] is None
    assert os.path.isdir(os.path.join(os.path.dirname(os.path.abspath(bench.__file__)), "allophant_amd", "csrc"))


def test_checkpoint_model_id_selects_the_wav2vec2_variant():
    """`nn.acoustic_model.model_id` decides the encoder the reference builds (acoustic_model.py:775-826): a checkpoint naming
    wav2vec2-base restores as the group-norm / post-LN variant without the attention mask, XLS-R as before, an unknown id is
    refused, and an explicit `amx_encoder` override carries the variant keys."""
    base = S.baseline_spec(S.wav2vec2_base_encoder(), 40)
    base["model_id"] = "facebook/wav2vec2-base"
    restored = checkpoint.spec_from_checkpoint(checkpoint.make_checkpoint(base, {}, synthetic_encoder=False))
    assert restored["feat_extract_norm"] == "group" and restored["conv_bias"] is False
    assert restored["stable_layer_norm"] is False and restored["use_attention_mask"] is False
    assert (restored["hidden"], restored["layers"], restored["heads"], restored["ffn"]) == (768, 12, 12, 3072)
    xlsr = checkpoint.spec_from_checkpoint(checkpoint.make_checkpoint(S.baseline_spec(S.xlsr_300m_encoder(), 40), {}))
    assert xlsr.get("feat_extract_norm", "layer") == "layer" and xlsr.get("stable_layer_norm", True)
    for model_id, shape in (("facebook/wav2vec2-xls-r-1b", (1280, 48, 16, 5120)), ("facebook/wav2vec2-xls-r-2b", (1920, 48, 16, 7680)),
                            ("facebook/mms-300m", (1024, 24, 16, 4096)), ("facebook/mms-1b", (1280, 48, 16, 5120))):
        named = S.baseline_spec(S.xlsr_300m_encoder(), 40)
        named["model_id"] = model_id
        wide = checkpoint.spec_from_checkpoint(checkpoint.make_checkpoint(named, {}))
        assert (wide["hidden"], wide["layers"], wide["heads"], wide["ffn"]) == shape, model_id
    unknown = S.baseline_spec(S.xlsr_300m_encoder(), 40)
    unknown["model_id"] = "someone/some-model"
    with pytest.raises(ValueError):
        checkpoint.spec_from_checkpoint(checkpoint.make_checkpoint(unknown, {}))
    enc = S.tiny_encoder(1)
    enc.update(feat_extract_norm="group", conv_bias=False, stable_layer_norm=False, use_attention_mask=False)
    tiny = checkpoint.spec_from_checkpoint(checkpoint.make_checkpoint(S.baseline_spec(enc, 7), {}, synthetic_encoder=True))
    assert tiny["feat_extract_norm"] == "group" and tiny["use_attention_mask"] is False
    cfg, _ = _spec_to_structs(tiny, "f16x3")
    assert (cfg.feat_extract_norm, cfg.conv_bias, cfg.stable_layer_norm, cfg.use_attention_mask) == (lib.NORM_GROUP, 0, 0, 0)
    cfg, _ = _spec_to_structs(S.baseline_spec(S.tiny_encoder(1), 7), "f16x3")
    assert (cfg.feat_extract_norm, cfg.conv_bias, cfg.stable_layer_norm, cfg.use_attention_mask) == (lib.NORM_LAYER, 1, 1, 1)


def test_bench_work_model_matches_the_survey_accounting():
    """`bench.py`'s roofline accounting (algorithmic FLOPs per step, the figure `roofline.achieved` is computed from) against the
    per-config totals of SURVEY.md Appendix D: 12.309 / 11.899 / 24.392 TFLOP per batch for configs 2 / 4 / 5, attention 0.783 /
    0.390 / 7.073, conv stage 1.570 (config 2); and the routing of the products to kernel classes for both wav2vec 2.0 families."""
    import bench

    # (config 1: the baseline schema on 1 x 3 s -- 0.110 TFLOP, 149 frames; its products are too short for the ping-pong kernel)
    expected = {1: (0.110, 0.002, 149, 1, 3), 2: (12.309, 0.783, 499, 98, 5), 4: (11.899, 0.390, 249, 97, 5), 5: (24.392, 7.073, 2999, 98, 5)}
    for number, (utterances, seconds, _phones, hierarchical) in bench.CONFIG_PRESETS.items():
        w = bench.work_model(bench.build_spec(hierarchical), utterances, int(seconds * 16000), 2)
        total, attention, frames, pp_launches, ln_launches = expected[number]
        assert abs(w["total"] / 1e12 - total) < 0.01, (number, w["total"])
        assert abs(w["attention"] / 1e12 - attention) < 0.001
        assert w["frames_per_utt"] == frames
        assert (w["gemm_pp_launches"], w["gemm_ln_launches"]) == (pp_launches, ln_launches), number
    w = bench.work_model(bench.build_spec(False), 32, 160000, 2)
    assert abs((w["conv0"] + w["gemm_ln"] + w["conv_tail"]) / 1e12 - 1.570) < 0.001
    # the group-norm family has no fused conv + LayerNorm kernel: conv layers 1-5 are ping-pong products with a GELU epilogue
    base = bench.work_model(bench.build_spec(False, "w2v2-base"), 32, 160000, 2)
    assert base["gemm_ln_launches"] == 0 and base["gemm_pp_launches"] == 5 + 1 + 4 * 12 + 1
    large = bench.work_model(bench.build_spec(False, "w2v2-large"), 32, 160000, 2)
    assert large["gemm_ln_launches"] == 0 and abs(large["total"] - w["total"]) < 1e9
    # the XLS-R 1B / 2B shapes (round 6): 48 layers of four ping-pong products + feature projection + phoneme head, the conv stage of
    # the 300M model; totals from the formulas above at hidden 1280 / 1920
    from allophant_amd import spec as S

    for name, total, attention in (("xlsr-1b", 34.139, 1.958), ("xlsr-2b", 73.341, 2.937)):
        spec = bench.build_spec(False, name)
        S.validate(spec)
        big = bench.work_model(spec, 32, 160000, 2)
        assert (big["gemm_pp_launches"], big["gemm_ln_launches"]) == (4 * 48 + 2, 5)
        assert abs(big["total"] / 1e12 - total) < 0.01 and abs(big["attention"] / 1e12 - attention) < 0.001, (name, big["total"], big["attention"])
        assert big["gemm_ln"] == w["gemm_ln"] and big["conv0"] == w["conv0"]
