"""Procedural (seeded, name-keyed) weights in the reference's ``Allophant.state_dict()`` key layout.

Real checkpoints (``kgnlp/allophant*`` on the Hugging Face hub) are unreachable offline, so benchmarks, smoke tests and the
full-size golden vectors use weights generated here.  Every tensor is drawn from its own ``torch.Generator`` seeded with
``crc32(key) ^ seed`` so that a tensor's values depend only on its name, shape and the seed -- the same call on the GPU box
reproduces bit-identical weights without shipping 1.26 GB of fixtures.  Key layout: SURVEY.md Appendix B (reference
``allophant/estimator.py:216,1076,1122`` stores ``Allophant.state_dict()`` as ``Checkpoint.model_state``).
"""
from __future__ import annotations

import math
import zlib
from typing import Any, Dict, List, Optional

import torch
from torch import Tensor

from . import spec as _spec

AM = "_acoustic_model._model."
PROJ = "_projection._layers."


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _randn(key: str, seed: int, shape, std: float = 1.0, mean: float = 0.0) -> Tensor:
    return torch.randn(*shape, generator=_gen(key, seed), dtype=torch.float32) * std + mean


def head_input_size(spec: Dict[str, Any], node: Dict[str, Any]) -> int:
    """Input width of a classifier (reference acoustic_model.py:309-330)."""
    sizes = {c["name"]: c["size"] for c in spec["classes"]}
    blanks = bool(spec.get("dependency_blanks", True))
    total = 0
    for dep in node["dependencies"]:
        if _spec.OUTPUT_PATTERN.match(dep):
            total += spec["hidden"]
        else:
            total += sizes[dep] + (_spec.BLANK_OFFSET if blanks else 0)
    return total


def make_state_dict(spec: Dict[str, Any], seed: int = 0, include_unused: bool = True) -> Dict[str, Tensor]:
    """Generates a complete ``Allophant.state_dict()`` for ``spec`` (fp32 CPU tensors)."""
    sd: Dict[str, Tensor] = {}
    C, D, Fd = spec["conv_dim"], spec["hidden"], spec["ffn"]

    def put(key: str, shape, std: float = 1.0, mean: float = 0.0):
        sd[key] = _randn(key, seed, shape, std, mean)

    if include_unused:
        put(AM + "masked_spec_embed", (D,), 1.0)
    c_in = 1
    group = spec.get("feat_extract_norm", "layer") == "group"
    for i, k in enumerate(spec["conv_kernel"]):
        p = f"{AM}feature_extractor.conv_layers.{i}."
        put(p + "conv.weight", (C, c_in, k), math.sqrt(2.0 / (c_in * k)))
        if spec.get("conv_bias", True):
            put(p + "conv.bias", (C,), 0.05)
        # "group": GroupNorm behind layer 0 only (same key names, transformers Wav2Vec2GroupNormConvLayer)
        if not group or i == 0:
            put(p + "layer_norm.weight", (C,), 0.1, 1.0)
            put(p + "layer_norm.bias", (C,), 0.1)
        c_in = C
    p = AM + "feature_projection."
    put(p + "layer_norm.weight", (C,), 0.1, 1.0)
    put(p + "layer_norm.bias", (C,), 0.1)
    put(p + "projection.weight", (D, C), 1.0 / math.sqrt(C))
    put(p + "projection.bias", (D,), 0.05)
    p = AM + "encoder.pos_conv_embed.conv."
    put(p + "bias", (D,), 0.05)
    put(p + "parametrizations.weight.original0", (1, 1, spec["pos_kernel"]), 0.2, 1.5)
    put(p + "parametrizations.weight.original1", (D, D // spec["pos_groups"], spec["pos_kernel"]), 1.0)
    put(AM + "encoder.layer_norm.weight", (D,), 0.1, 1.0)
    put(AM + "encoder.layer_norm.bias", (D,), 0.1)
    for i in range(spec["layers"]):
        p = f"{AM}encoder.layers.{i}."
        for name in ("q_proj", "k_proj", "v_proj"):
            put(p + f"attention.{name}.weight", (D, D), 1.0 / math.sqrt(D))
            put(p + f"attention.{name}.bias", (D,), 0.05)
        put(p + "attention.out_proj.weight", (D, D), 0.5 / math.sqrt(D))
        put(p + "attention.out_proj.bias", (D,), 0.02)
        put(p + "layer_norm.weight", (D,), 0.1, 1.0)
        put(p + "layer_norm.bias", (D,), 0.1)
        put(p + "feed_forward.intermediate_dense.weight", (Fd, D), 1.0 / math.sqrt(D))
        put(p + "feed_forward.intermediate_dense.bias", (Fd,), 0.05)
        put(p + "feed_forward.output_dense.weight", (D, Fd), 0.5 / math.sqrt(Fd))
        put(p + "feed_forward.output_dense.bias", (D,), 0.02)
        put(p + "final_layer_norm.weight", (D,), 0.1, 1.0)
        put(p + "final_layer_norm.bias", (D,), 0.1)

    if spec.get("add_adapter") and include_unused:
        # Wav2Vec2Adapter (strided Conv1d + GLU layers behind the encoder): owned by the reference's state dict, computed by its HF
        # model and never read by `Estimator.predict` (spec.validate) -- the device never receives these tensors
        k = int(spec.get("adapter_kernel_size", 3))
        for i in range(int(spec.get("num_adapter_layers", 3))):
            put(f"{AM}adapter.layers.{i}.conv.weight", (2 * D, D, k), 1.0 / math.sqrt(D * k))
            put(f"{AM}adapter.layers.{i}.conv.bias", (2 * D,), 0.05)

    E = spec.get("embedding_size")
    for node in spec["classes"]:
        p = f"{PROJ}{node['name']}."
        n_in = head_input_size(spec, node)
        composed = node["name"] == _spec.PHONEME and E
        allophone = node["name"] == _spec.PHONEME and spec.get("allophone_layer")
        # with an allophone layer the classifier predicts the shared phone inventory (acoustic_model.py:389-397)
        classes_out = spec.get("shared_phones", node["size"]) if allophone else node["size"]
        n_out = E if composed else classes_out + _spec.BLANK_OFFSET
        if node.get("time_layer"):
            # ProjectingMultiheadAttention (acoustic_model.py:237-253); key names of SURVEY.md Appendix B
            t = p + "_time_distributed_layer."
            put(t + "input_projection.weight", (n_out, n_in), 2.0 / math.sqrt(n_in))
            put(t + "input_projection.bias", (n_out,), 0.1)
            put(t + "layer_norm.weight", (n_out,), 0.1, 1.0)
            put(t + "layer_norm.bias", (n_out,), 0.1)
            put(t + "attention.in_proj_weight", (3 * n_out, n_out), 1.5 / math.sqrt(n_out))
            put(t + "attention.in_proj_bias", (3 * n_out,), 0.1)
            put(t + "attention.out_proj.weight", (n_out, n_out), 1.5 / math.sqrt(n_out))
            put(t + "attention.out_proj.bias", (n_out,), 0.1)
        else:
            put(p + "_time_distributed_layer.weight", (n_out, n_in), 2.0 / math.sqrt(n_in))
            put(p + "_time_distributed_layer.bias", (n_out,), 0.1)
        if composed:
            rows = 1 + sum(spec["composition_categories"])
            put(p + "_composition_layer._attribute_embeddings.weight", (rows, E), 0.6)
        if node["name"] == _spec.PHONEME and spec.get("allophone_layer") and include_unused:
            # (n_langs, shared_phones+1, phonemes+1); unused in predict mode (acoustic_model.py:161-167)
            sd[p + "_allophone_layer._allophone_matrices"] = torch.zeros(
                2, spec.get("shared_phones", node["size"]) + 1, node["size"] + 1)
    return sd


def make_inventory(spec: Dict[str, Any], phones: int, seed: int = 0) -> Tensor:
    """Synthetic ``composition_feature_matrix`` (reference phonetic_features.py:808-818 contract):
    int64 [P, F] with 0 <= tfi[p, f] < n_f.  Stands in for real inventories ('es', ['es','it'], ...) because the
    Allophoible table is not available offline."""
    cats = spec["composition_categories"]
    g = _gen(f"inventory/{phones}", seed)
    cols = [torch.randint(0, n, (phones,), generator=g, dtype=torch.int64) for n in cats]
    return torch.stack(cols, 1)


def category_offsets(spec: Dict[str, Any]) -> Tensor:
    """``cumsum([1, n_0, n_1, ...])[:-1]`` (reference acoustic_model.py:196-207)."""
    cats = [1] + list(spec["composition_categories"])
    return torch.tensor(cats, dtype=torch.int64).cumsum(0)[:-1]


def make_audio(n: int, length: int, seed: int = 1234, ragged: bool = False):
    """Synthetic 16 kHz batch: ``randn * 0.1`` (+ slow sinusoid so the mean is not ~0), zero right-padded to
    ``max(lengths)`` like the reference batcher (batching.py:174).  Returns (audio [N, L] f32, lengths [N] i64)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    audio = torch.randn(n, length, generator=g, dtype=torch.float32) * 0.1
    t = torch.arange(length, dtype=torch.float32) / 16000.0
    audio = audio + 0.05 * torch.sin(2 * math.pi * 220.0 * t).unsqueeze(0) + 0.01
    if ragged:
        lengths = torch.randint(length // 2, length + 1, (n,), generator=g, dtype=torch.int64)
        lengths[0] = length  # the batch is padded to exactly max(lengths) (utils.py:62-63)
    else:
        lengths = torch.full((n,), length, dtype=torch.int64)
    mask = torch.arange(length).unsqueeze(0) < lengths.unsqueeze(1)
    return audio * mask, lengths
