"""Reader for checkpoint dicts in the reference ``Checkpoint`` schema (allophant/estimator.py:199-249).

The reference stores ``Checkpoint.Schema().dump(checkpoint)`` with ``torch.save``: a plain dict with the fields
``config, allophant_version, feature_size, sample_rate, attribute_graph{nodes, node_indices, edges}, epoch,
phonetic_indexer_state, dataset_meta_data, model_state, additional, history, optimization_states``.  Only the fields that
shape the prediction path are read here; the encoder shape and variant follow ``nn.acoustic_model.model_id`` (XLS-R-300m
in default_config.toml:34-37 and every released checkpoint; the table ``MODEL_ID_ENCODERS`` also knows the group-norm /
post-LN wav2vec2-base and -large) unless the checkpoint carries an explicit ``additional["amx_encoder"]`` override
(synthetic checkpoints used by the plumbing tests -- real hub checkpoints are not reachable offline).

Composition models: upstream rebuilds the embedding-table layout (``_category_offsets``, a non-persistent buffer) from the
phonetic indexer, which itself is rebuilt from ``phonetic_indexer_state`` = {phoneme_inventory, language_allophones,
table_file} (phonetic_features.py:111-115, 746-786; acoustic_model.py:191-207, 422-446).  ``indexer_from_checkpoint`` does
the same from the embedded table text: the training phones are ``language_allophones.shared_phones`` for allophone models
and ``phoneme_inventory`` otherwise.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple

import torch

from . import spec as _spec

XLSR_MODEL_IDS = ("facebook/wav2vec2-xls-r-300m",)


def _large_groupnorm_encoder() -> Dict[str, Any]:
    encoder = _spec.xlsr_300m_encoder()
    encoder.update(feat_extract_norm="group", conv_bias=False, stable_layer_norm=False, use_attention_mask=False)
    return encoder


# `nn.acoustic_model.model_id` -> encoder shape.  The reference reads config.json / preprocessor_config.json of whatever id
# the config names (acoustic_model.py:775-826); the hub is unreachable offline, so the shapes of the public checkpoints are
# tabulated here from their published configs [from memory -- an id outside this table needs `additional["amx_encoder"]`
# or the live-model binding `spec.spec_from_reference_model`, which reads the real config objects].
MODEL_ID_ENCODERS = {
    "facebook/wav2vec2-xls-r-300m": _spec.xlsr_300m_encoder,           # every released Allophant checkpoint
    "facebook/wav2vec2-xls-r-1b": _spec.xlsr_1b_encoder,               # hidden 1280 / head_dim 80 (round 6)
    "facebook/wav2vec2-xls-r-2b": _spec.xlsr_2b_encoder,               # hidden 1920 / head_dim 120
    "facebook/wav2vec2-large-xlsr-53": _spec.xlsr_300m_encoder,        # same shape and variant (layer norm, pre-LN, mask)
    "facebook/wav2vec2-large-lv60": _spec.xlsr_300m_encoder,
    "facebook/mms-300m": _spec.xlsr_300m_encoder,                      # the MMS pre-trained models keep the XLS-R architectures
    "facebook/mms-1b": _spec.xlsr_1b_encoder,                          # (their fine-tuned ASR heads add `adapter_attn_dim`: refused)
    "facebook/wav2vec2-base": _spec.wav2vec2_base_encoder,             # group norm, post-LN, return_attention_mask=False
    "facebook/wav2vec2-large": _large_groupnorm_encoder,
}
ENCODER_KEYS = ("conv_dim", "conv_kernel", "conv_stride", "hidden", "layers", "heads", "ffn", "pos_kernel", "pos_groups", "eps",
                "do_normalize", "feat_extract_norm", "conv_bias", "stable_layer_norm", "use_attention_mask")


def _time_config(c: Dict[str, Any]) -> Optional[Dict[str, Any]]:
    """``MultiheadAttentionConfig`` dump of a class's ``time_layer`` (config.py:596-610)."""
    layer = c.get("time_layer")
    if not layer:
        return None
    return {"type": "multi-head-attention", "num_heads": int(layer.get("num_heads", 1)),
            "positional_embeddings": bool(layer.get("positional_embeddings", False))}


def make_checkpoint(spec: Dict[str, Any], state_dict: Dict[str, torch.Tensor], synthetic_encoder: bool = False,
                    indexer_state: Optional[Dict[str, Any]] = None) -> Dict[str, Any]:
    """Synthetic checkpoint dict with the reference's field names (used by tests and the config-1 plumbing case).
    With ``indexer_state`` (a ``PhoneticIndexerState`` dump) the composition layout is NOT stored under ``additional``:
    it has to be rebuilt from the embedded table like upstream does."""
    classes = spec["classes"]
    nodes = [
        {"name": c["name"], "size": c["size"], "time_layer_config": _time_config(c), "dependencies": list(c["dependencies"])}
        for c in classes
    ]
    index = {c["name"]: i for i, c in enumerate(classes)}
    edges = [[index[d] for d in c["dependencies"] if not _spec.OUTPUT_PATTERN.match(d)] for c in classes]
    projection = {
        "classes": [{"name": c["name"], "dependencies": list(c["dependencies"]), "time_layer": _time_config(c), "loss": {"type": "CTC"}}
                    for c in classes],
        "feature_set": "phoible",
        "phoneme_layer": "allophones" if spec.get("allophone_layer") else "shared",
        "acoustic_model_dropout": 0.0,
        "dependency_blanks": bool(spec.get("dependency_blanks", True)),
        "allophone_l2_alpha": 10.0,
        "embedding_composition": {"embedding_size": spec["embedding_size"]} if spec.get("embedding_size") else None,
    }
    additional: Dict[str, Any] = {}
    if synthetic_encoder:
        additional["amx_encoder"] = {k: spec[k] for k in ENCODER_KEYS if k in spec}
    if spec.get("composition_categories") is not None and indexer_state is None:
        additional["amx_composition_categories"] = list(spec["composition_categories"])
    if spec.get("shared_phones") is not None:
        additional["amx_shared_phones"] = int(spec["shared_phones"])
    return {
        "config": {"nn": {"projection": projection,
                          "acoustic_model": {"type": "wav2vec2-pretrained", "model_id": spec.get("model_id", XLSR_MODEL_IDS[0])},
                          "loss": {"type": "CTC"}}},
        "allophant_version": "1.0.0",
        "feature_size": 1,
        "sample_rate": 16000,
        "attribute_graph": {"nodes": nodes, "node_indices": index, "edges": edges},
        "epoch": {"epoch": 0, "step": 0},
        "phonetic_indexer_state": indexer_state,
        "dataset_meta_data": [],
        "model_state": state_dict,
        "additional": additional,
        "history": [],
        "optimization_states": None,
    }


def indexer_from_checkpoint(checkpoint: Dict[str, Any]):
    """``(AttributeTable, training phones)`` from ``phonetic_indexer_state``, or ``(None, None)`` when the checkpoint has
    no embedded table."""
    from .phonetic import AttributeTable

    state = checkpoint.get("phonetic_indexer_state")
    if not state or not state.get("table_file"):
        return None, None
    allophones = state.get("language_allophones")
    inventory = list(state.get("phoneme_inventory") or [])
    # PhoneticAttributeIndexer.from_config (phonetic_features.py:746-786): the attribute subset is every classifier name
    # and class dependency of the projection, in order of first appearance
    subset: Optional[List[str]] = None
    projection = ((checkpoint.get("config") or {}).get("nn") or {}).get("projection") or {}
    if projection.get("classes"):
        subset = []
        for entry in projection["classes"]:
            for name in (entry["name"], *entry.get("dependencies", [])):
                if not _spec.OUTPUT_PATTERN.match(name) and name not in subset:
                    subset.append(name)
    if allophones and allophones.get("shared_phones"):
        # an allophone-layer checkpoint: inventories restricted to the training languages, like upstream's restored indexer
        table = AttributeTable(state["table_file"], subset, inventory, allophones)
        training = list(allophones["shared_phones"])
    else:
        # without a mapping upstream's `from_config` restores an UNRESTRICTED indexer (the match falls through to
        # `phoneme_subset = None`, phonetic_features.py:765-775); the training phones are the recorded inventory
        table = AttributeTable(state["table_file"], subset)
        training = inventory
    return table, training


_EXTRACTOR = "_acoustic_model._model.feature_extractor.conv_layers."
_ENCODER = "_acoustic_model._model.encoder."


def check_encoder_against_state(encoder: Dict[str, Any], state: Dict[str, torch.Tensor], source: str) -> None:
    """Cross-checks an encoder description (a ``MODEL_ID_ENCODERS`` entry is written from memory) against what the weights
    themselves say: a wrong ``conv_bias`` would silently drop the biases, a wrong ``feat_extract_norm`` the norms.  Checked:
    conv biases, the per-layer conv norms, layer and conv counts, hidden / ffn / conv widths and kernels.
    ``stable_layer_norm`` and ``use_attention_mask`` leave NO trace in the weights (both encoder variants own the same
    tensors; the mask is a preprocessor setting) -- they cannot be verified here and rest on the table alone."""
    def has(key: str) -> bool:
        return key in state

    problems: List[str] = []
    if not has(_EXTRACTOR + "0.conv.weight"):
        return  # not a wav2vec 2.0 state dict of the reference's naming: amx_create reports what is missing
    bias = has(_EXTRACTOR + "0.conv.bias")
    if bias != bool(encoder.get("conv_bias", True)):
        problems.append(f"conv_bias={encoder.get('conv_bias', True)} but the state dict "
                        f"{'holds' if bias else 'has no'} conv_layers.0.conv.bias")
    later_norm = has(_EXTRACTOR + "1.layer_norm.weight")
    norm = "layer" if later_norm else "group"
    if norm != encoder.get("feat_extract_norm", "layer"):
        problems.append(f"feat_extract_norm={encoder.get('feat_extract_norm', 'layer')!r} but the state dict "
                        f"{'holds' if later_norm else 'has no'} conv_layers.1.layer_norm (= {norm!r})")
    n_conv = 0
    while has(f"{_EXTRACTOR}{n_conv}.conv.weight"):
        n_conv += 1
    if n_conv != len(encoder["conv_kernel"]):
        problems.append(f"{len(encoder['conv_kernel'])} conv layers described, {n_conv} in the state dict")
    else:
        for i in range(n_conv):
            shape = tuple(state[f"{_EXTRACTOR}{i}.conv.weight"].shape)
            if shape[0] != encoder["conv_dim"] or shape[2] != encoder["conv_kernel"][i]:
                problems.append(f"conv layer {i}: weight {shape}, described as {encoder['conv_dim']} channels, kernel "
                                f"{encoder['conv_kernel'][i]}")
    layers = 0
    while has(f"{_ENCODER}layers.{layers}.attention.q_proj.weight"):
        layers += 1
    if layers and layers != encoder["layers"]:
        problems.append(f"{encoder['layers']} encoder layers described, {layers} in the state dict")
    if layers:
        q = tuple(state[f"{_ENCODER}layers.0.attention.q_proj.weight"].shape)
        f = tuple(state[f"{_ENCODER}layers.0.feed_forward.intermediate_dense.weight"].shape)
        if q[0] != encoder["hidden"] or f[0] != encoder["ffn"]:
            problems.append(f"hidden / ffn described as {encoder['hidden']} / {encoder['ffn']}, the weights are {q[0]} / {f[0]}")
    # wav2vec 2.0 adapters (`add_adapter`): the reference runs them and reads nothing of their output (spec.validate), so their
    # weights are simply not uploaded -- unless the adapter changes the width the classifiers were built for, which the reference
    # itself cannot run (acoustic_model.py:822 sizes them by `output_hidden_size`, the states keep `hidden_size`)
    adapter = _EXTRACTOR.replace("feature_extractor.conv_layers.", "adapter.")
    if has(adapter + "proj.weight") and tuple(state[adapter + "proj.weight"].shape)[0] != encoder["hidden"]:
        problems.append(f"the adapter projects to {tuple(state[adapter + 'proj.weight'].shape)[0]} columns: classifiers sized for that "
                        f"width cannot read encoder states of {encoder['hidden']} (the reference fails on this checkpoint too)")
    elif has(adapter + "layers.0.conv.weight"):
        encoder.setdefault("add_adapter", True)
    if problems:
        raise ValueError(f"the encoder description from {source} does not match the checkpoint's weights: " + "; ".join(problems)
                         + ".  Pass the right description in additional['amx_encoder'].")


def spec_from_checkpoint(checkpoint: Dict[str, Any]) -> Dict[str, Any]:
    for field in ("config", "attribute_graph", "model_state", "sample_rate", "feature_size"):
        if field not in checkpoint:
            raise ValueError(f"checkpoint is missing the field {field!r} of the reference Checkpoint schema")
    if checkpoint["sample_rate"] != 16000:
        # acoustic_model.py:788-794
        raise ValueError(
            "Audio resampling config and the sampling rate required by Wav2Vec2 do not match. "
            f"Expected 16000kHz, got {checkpoint['sample_rate']}kHz")
    nn_config = checkpoint["config"]["nn"]
    additional = checkpoint.get("additional") or {}
    acoustic = nn_config.get("acoustic_model", {})
    if "amx_encoder" in additional:
        encoder = dict(additional["amx_encoder"])
        check_encoder_against_state(encoder, checkpoint["model_state"], "additional['amx_encoder']")
    elif acoustic.get("model_id") in MODEL_ID_ENCODERS:
        encoder = MODEL_ID_ENCODERS[acoustic["model_id"]]()
        check_encoder_against_state(encoder, checkpoint["model_state"], f"MODEL_ID_ENCODERS[{acoustic['model_id']!r}]")
    else:
        raise ValueError(f"Unsupported model type: {acoustic.get('type')!r} / {acoustic.get('model_id')!r}")
    projection = nn_config["projection"]
    spec = dict(encoder)
    spec["classes"] = []
    for n in checkpoint["attribute_graph"]["nodes"]:
        entry = {"name": n["name"], "size": int(n["size"]), "dependencies": list(n["dependencies"])}
        layer = n.get("time_layer_config")
        if layer:
            entry["time_layer"] = {"num_heads": int(layer.get("num_heads", 1)),
                                   "positional_embeddings": bool(layer.get("positional_embeddings", False))}
        spec["classes"].append(entry)
    spec["dependency_blanks"] = bool(projection.get("dependency_blanks", True))
    composition = projection.get("embedding_composition")
    spec["embedding_size"] = int(composition["embedding_size"]) if composition else None
    spec["allophone_layer"] = projection.get("phoneme_layer", "shared") != "shared"
    state = checkpoint["model_state"]
    emb_key = "_projection._layers.phoneme._composition_layer._attribute_embeddings.weight"
    spec["composition_categories"] = additional.get("amx_composition_categories")
    if spec["embedding_size"] and spec["composition_categories"] is None:
        table, training = indexer_from_checkpoint(checkpoint)
        if table is not None and training:
            counts = table.category_counts(training)
            if emb_key in state and int(state[emb_key].shape[0]) != 1 + sum(counts):
                raise ValueError(
                    f"the embedded attribute table yields {1 + sum(counts)} attribute embeddings for the training phones, "
                    f"the checkpoint holds {int(state[emb_key].shape[0])}")
            spec["composition_categories"] = counts
    if spec["embedding_size"] and spec["composition_categories"] is None:
        raise ValueError(
            "composition checkpoints need the per-feature category counts (`_category_offsets` is a non-persistent "
            f"buffer upstream); table rows available: {state[emb_key].shape[0] if emb_key in state else 'n/a'}")
    if spec["allophone_layer"]:
        key = "_projection._layers.phoneme._allophone_layer._allophone_matrices"
        if "amx_shared_phones" in additional:
            spec["shared_phones"] = int(additional["amx_shared_phones"])
        elif key in state:
            spec["shared_phones"] = int(state[key].shape[1]) - 1
    _spec.validate(spec)
    return spec
