"""The pieces of the reference-side binding that run without a GPU: deriving the spec and the
training inventory from a LIVE reference model.  The real reference is imported through oracle/ref_import.py (test
infrastructure); the test is skipped where /root/reference does not exist (the GPU box)."""
import os
import sys

import pytest
import torch

from allophant_amd import spec as S, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/allophant"), reason="needs the reference checkout")


def _reference_model(spec, n_train):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_import

    ref_import.install()
    table = None
    if spec.get("embedding_size") or spec.get("allophone_layer"):
        table = synthetic.make_inventory(spec, n_train, seed=5)
        for f, ncat in enumerate(spec["composition_categories"]):
            table[:ncat, f] = torch.arange(ncat)  # every category occurs: n_f == max + 1 (acoustic_model.py:196-207)
    # EmbeddingCompositionLayer.__init__ adds the category offsets to the table IN PLACE (acoustic_model.py:205): keep a copy
    expected = None if table is None else table.clone()
    _estimator, model = ref_import.build_reference_estimator(spec, table)
    return model, expected


def _comparable(spec):
    keys = ("conv_dim", "conv_kernel", "conv_stride", "hidden", "layers", "heads", "ffn", "pos_kernel", "pos_groups", "eps",
            "do_normalize", "dependency_blanks", "embedding_size", "allophone_layer", "composition_categories")
    out = {k: spec.get(k) for k in keys}
    out["variant"] = (spec.get("feat_extract_norm", "layer"), bool(spec.get("conv_bias", True)),
                      bool(spec.get("stable_layer_norm", True)), bool(spec.get("use_attention_mask", True)))
    out["classes"] = [{"name": c["name"], "dependencies": list(c["dependencies"]), "time_layer": c.get("time_layer")}
                      for c in spec["classes"]]
    out["embedding_size"] = out["embedding_size"] or None
    return out


def test_spec_from_a_live_reference_model_multitask_allophones():
    spec = S.multitask_spec(S.tiny_encoder(2), ["syllabic", "long", "nasal"], embedding_size=16, train_phonemes=9,
                            n_features=5, n_values=3, allophone_layer=True)
    spec["shared_phones"] = 11
    model, table = _reference_model(spec, 11)
    derived = S.spec_from_reference_model(model)
    assert _comparable(derived) == _comparable(spec)
    assert derived["shared_phones"] == 11
    sizes = {c["name"]: c["size"] for c in derived["classes"]}
    assert sizes["syllabic"] == 3 and sizes["long"] == 3
    assert torch.equal(S.training_inventory_of_reference_model(model), table)
    # the state_dict of that model carries exactly the keys the packer reads
    state = synthetic.make_state_dict(spec, seed=0)
    assert set(state) <= set(model.state_dict())


def test_spec_from_a_live_reference_model_hierarchical_time_layer():
    spec = S.hierarchical_spec(S.tiny_encoder(2), ["syllabic", "long"], embedding_size=16, train_phonemes=8, n_features=4,
                               dependency_blanks=False)
    by_name = {c["name"]: c for c in spec["classes"]}
    by_name["syllabic"]["dependencies"] = ["OUTPUT_1"]
    by_name["long"].update(dependencies=["syllabic", "OUTPUT"], time_layer={"num_heads": 2, "positional_embeddings": True})
    S.validate(spec)
    model, table = _reference_model(spec, 8)
    derived = S.spec_from_reference_model(model)
    assert _comparable(derived) == _comparable(spec)
    assert {c["name"]: c["size"] for c in derived["classes"]} == {c["name"]: c["size"] for c in spec["classes"]}
    assert torch.equal(S.training_inventory_of_reference_model(model), table)


def test_spec_from_a_live_reference_model_baseline():
    spec = S.baseline_spec(S.tiny_encoder(1), 14)
    model, _ = _reference_model(spec, 14)
    derived = S.spec_from_reference_model(model)
    assert _comparable(derived) == _comparable(spec)
    assert derived["classes"][0]["size"] == 14 and S.training_inventory_of_reference_model(model) is None


def test_spec_from_a_live_reference_model_groupnorm_postln():
    """The reference builds whatever ``model_id`` names (acoustic_model.py:775-826): a group-norm / post-LN wav2vec 2.0 whose
    preprocessor has return_attention_mask=False comes out as that variant, and its state_dict has no conv bias and a norm
    behind conv layer 0 only."""
    enc = S.tiny_encoder(2)
    enc.update(feat_extract_norm="group", conv_bias=False, stable_layer_norm=False, use_attention_mask=False)
    spec = S.baseline_spec(enc, 12)
    model, _ = _reference_model(spec, 12)
    derived = S.spec_from_reference_model(model)
    assert _comparable(derived) == _comparable(spec)
    assert derived["feat_extract_norm"] == "group" and derived["use_attention_mask"] is False
    state = synthetic.make_state_dict(spec, seed=0)
    reference_keys = set(model.state_dict())
    assert set(state) <= reference_keys
    extractor = {k for k in reference_keys if ".feature_extractor." in k}
    assert extractor == {k for k in state if ".feature_extractor." in k}  # no bias, one norm: exactly the generated keys
