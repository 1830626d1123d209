"""TEST INFRASTRUCTURE ONLY: CPU oracle for the Allophant acoustic-encoder forward path (``Estimator.predict``).

This is a from-scratch fp32 CPU restatement of the reference's algorithm for the hot path, written with plain
``torch.nn.functional`` primitives (no ``transformers`` import, no reference import).  It is the checker for the HIP
path in ``allophant_amd/csrc``; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product path never routes through it.

PARITY PIN: this restatement is pinned against the *real* reference (imported in the build container through
``oracle/ref_import.py``) by ``oracle/gen_golden.py``; the resulting inputs/outputs are committed under
``tests/golden/`` and ``tests/test_oracle_golden.py`` re-checks the oracle against them on every run.  The upstream repo
holds no tests or golden vectors for this path, so outputs of the reference itself are the pin.

The heavy arithmetic of the path lives in a third-party dependency that is not vendored in /root/reference:
``transformers==4.41.2`` (``pyproject.toml:23``), class ``Wav2Vec2Model``; the build container has transformers 5.15.0,
which is what the goldens were generated with.  Its published algorithm is restated in ``wav2vec2_hidden_states`` below
for both variants the reference can be pointed at (it builds whatever ``model_id`` names, acoustic_model.py:775-826):
``feat_extract_norm="layer"`` + ``do_stable_layer_norm=True`` (XLS-R, every released Allophant checkpoint) and
``feat_extract_norm="group"`` + ``do_stable_layer_norm=False`` (wav2vec2-base / -large: GroupNorm over time behind conv
layer 0 only, bias-free convs, post-LN encoder layers; their preprocessor has ``return_attention_mask=False``, so the
reference calls the model with ``attention_mask=None``, acoustic_model.py:814,842-846).  The reference's own call site is
``allophant/network/acoustic_model.py:837-853``.

Reference lines followed (all relative to /root/reference):
  mask_sequence                     allophant/utils.py:45-76
  zero_mean_unit_var_norm           allophant/network/acoustic_model.py:762-767
  conv_length / downsampled_lengths allophant/network/frontend.py:192-203, acoustic_model.py:832-835
  Wav2Vec2AcousticModel.forward     allophant/network/acoustic_model.py:837-853
  HierarchicalProjection.forward    allophant/network/acoustic_model.py:471-524 (+ 309-330 dependency sizes)
  HierarchicalClassifier.forward    allophant/network/acoustic_model.py:284-306
  EmbeddingCompositionLayer         allophant/network/acoustic_model.py:191-234
  AllophoneMapping.forward(predict) allophant/network/acoustic_model.py:161-167
  AttributeGraph.sort               allophant/attribute_graph.py:124-199
  Estimator.predict                 allophant/estimator.py:1035-1046
  GreedyCTCDecoder.__call__         allophant/predictions.py:194-207
"""
from __future__ import annotations

import math
import re
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F
from torch import Tensor

OUTPUT = "OUTPUT"
OUTPUT_PATTERN = re.compile(r"^OUTPUT(?:_(\d+))?$")  # allophant/config.py:636-637
PHONEME = "phoneme"  # allophant/config.py:638
PHONE = "phone"  # allophant/config.py:639
BLANK_OFFSET = 1  # allophant/config.py:555

_AM = "_acoustic_model._model."
_PROJ = "_projection._layers."


# ----------------------------------------------------------------------------------------------------------------
# integer / mask helpers
# ----------------------------------------------------------------------------------------------------------------
def mask_sequence(lengths: Tensor, max_length: Optional[int] = None) -> Tensor:
    """allophant/utils.py:45-76 (batch_first, non-inverse form used by the hot path)."""
    if max_length is None:
        max_length = int(lengths.max())
    return torch.arange(max_length).unsqueeze(0) < lengths.unsqueeze(1)


def downsampled_lengths(lengths: Tensor, kernels: Sequence[int], strides: Sequence[int]) -> Tensor:
    """frontend.py:192-203 with use_padding=False applied once per conv layer (acoustic_model.py:832-835)."""
    for k, s in zip(kernels, strides):
        lengths = torch.div(lengths - k, s, rounding_mode="floor") + 1
    return lengths


def zero_mean_unit_var_norm(features: Tensor, lengths: Tensor, mask: Tensor) -> Tensor:
    """acoustic_model.py:762-767.  NB the mean sums the *padded* row (relies on zero padding)."""
    means = (features.sum(1) / lengths).unsqueeze(1)
    deviations = (features - means) * mask
    variances = (deviations ** 2).sum(1) / lengths
    return ((features - means) / (variances.unsqueeze(1) + 1e-7).sqrt()) * mask


def topological_order(classes: Sequence[Dict[str, Any]]) -> List[int]:
    """Order in which ``AttributeGraph.sort`` (attribute_graph.py:124-199) yields the nodes of an acyclic graph.

    Tarjan's SCC on a DAG emits every node when its depth-first visit finishes; roots are tried in index order and the
    edges of a node are its class (non-OUTPUT) dependencies in listed order (attribute_graph.py:67-74).
    """
    index = {c["name"]: i for i, c in enumerate(classes)}
    edges = [[index[d] for d in c["dependencies"] if not OUTPUT_PATTERN.match(d)] for c in classes]
    state = [0] * len(classes)  # 0 new, 1 open, 2 done
    order: List[int] = []

    def visit(node: int) -> None:
        if state[node] == 2:
            return
        if state[node] == 1:
            raise ValueError("Dependency cycle detected")
        state[node] = 1
        for target in edges[node]:
            visit(target)
        state[node] = 2
        order.append(node)

    for root in range(len(classes)):
        visit(root)
    return order


# ----------------------------------------------------------------------------------------------------------------
# wav2vec 2.0 (both variants: layer-norm extractor + pre-LN encoder, group-norm extractor + post-LN encoder), restated
# ----------------------------------------------------------------------------------------------------------------
def _pos_conv_weight(state: Dict[str, Tensor]) -> Tensor:
    base = _AM + "encoder.pos_conv_embed.conv."
    if base + "parametrizations.weight.original0" in state:
        g = state[base + "parametrizations.weight.original0"]
        v = state[base + "parametrizations.weight.original1"]
    else:  # older torch / transformers naming
        g = state[base + "weight_g"]
        v = state[base + "weight_v"]
    # weight_norm(dim=2): the norm is taken over dims (0, 1) separately for every kernel tap
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return v * (g / norm)


def feature_encoder(x: Tensor, state: Dict[str, Tensor], spec: Dict[str, Any]) -> Tensor:
    """``feat_extract_norm="layer"``: 7 x [Conv1d -> LayerNorm over channels -> exact GELU] (Wav2Vec2LayerNormConvLayer);
    ``"group"``: [Conv1d -> GroupNorm(C groups of one channel: statistics over TIME, of the padded batch tensor) -> GELU]
    for layer 0 (Wav2Vec2GroupNormConvLayer), then 6 x [Conv1d -> GELU] (Wav2Vec2NoLayerNormConvLayer).  ``conv_bias``
    False: bias-free convolutions.  Returns [N, T, C] (channels last)."""
    group = spec.get("feat_extract_norm", "layer") == "group"
    h = x.unsqueeze(1)
    for i, (k, s) in enumerate(zip(spec["conv_kernel"], spec["conv_stride"])):
        p = f"{_AM}feature_extractor.conv_layers.{i}."
        bias = state[p + "conv.bias"] if spec.get("conv_bias", True) else None
        h = F.conv1d(h, state[p + "conv.weight"], bias, stride=s)
        if group:
            if i == 0:
                h = F.group_norm(h, h.shape[1], state[p + "layer_norm.weight"], state[p + "layer_norm.bias"], 1e-5)
        else:
            h = h.transpose(-2, -1)
            h = F.layer_norm(h, (h.shape[-1],), state[p + "layer_norm.weight"], state[p + "layer_norm.bias"], 1e-5)
            h = h.transpose(-2, -1)
        h = F.gelu(h)
    return h.transpose(1, 2)


def wav2vec2_hidden_states(
    audio: Tensor, lengths: Tensor, state: Dict[str, Tensor], spec: Dict[str, Any], keep_intermediates: bool = False,
    padded: bool = False,
) -> Tuple[List[Tensor], Tensor, Dict[str, Tensor]]:
    """Returns (hidden_states list of [N,T,D] incl. final LN state, frame lengths, intermediates).

    `add_adapter` (Wav2Vec2Adapter behind the encoder): transformers returns ``hidden_states=encoder_outputs.hidden_states`` -- the
    adapter's result is `last_hidden_state` only -- and the reference reads the tuple (acoustic_model.py:839-853), so there is
    nothing to restate: golden g15 (a live adapter in the real reference) equals this function's output.

    ``padded`` (no upstream counterpart; the product's AMX_FLAG_PADDED): the batch is a block of a larger batch and keeps
    that batch's padded length, ``audio.shape[1] >= max(lengths)`` -- the mask is built for that length, which is what the
    reference computes for these utterances inside the larger batch."""
    eps = spec["eps"]
    inter: Dict[str, Tensor] = {}
    mask = mask_sequence(lengths, audio.shape[1] if padded else None)
    if mask.shape[1] != audio.shape[1]:
        raise ValueError("the batch must be padded to exactly max(lengths) (utils.py:62-63 / acoustic_model.py:765-767)")
    x = zero_mean_unit_var_norm(audio, lengths, mask) if spec.get("do_normalize", True) else audio
    if keep_intermediates:
        inter["normed_audio"] = x
    feats = feature_encoder(x, state, spec)  # [N, T, C]
    if keep_intermediates:
        inter["conv_out"] = feats
    frame_lengths = downsampled_lengths(lengths, spec["conv_kernel"], spec["conv_stride"])
    T = feats.shape[1]
    frame_mask = torch.arange(T).unsqueeze(0) < frame_lengths.unsqueeze(1)  # == _get_feature_vector_attention_mask

    # `use_attention_mask` False (preprocessor return_attention_mask=False): attention_mask=None -- no frame is zeroed and
    # every key is attended to (acoustic_model.py:842-846); the returned frame lengths are the downsampled ones either way
    masked = bool(spec.get("use_attention_mask", True))
    stable = bool(spec.get("stable_layer_norm", True))

    p = _AM + "feature_projection."
    h = F.layer_norm(feats, (feats.shape[-1],), state[p + "layer_norm.weight"], state[p + "layer_norm.bias"], eps)
    h = F.linear(h, state[p + "projection.weight"], state[p + "projection.bias"])
    if masked:
        h = h * frame_mask.unsqueeze(-1)  # hidden_states[~mask] = 0

    k = spec["pos_kernel"]
    pos = F.conv1d(
        h.transpose(1, 2), _pos_conv_weight(state), state[_AM + "encoder.pos_conv_embed.conv.bias"],
        padding=k // 2, groups=spec["pos_groups"],
    )
    if k % 2 == 0:
        pos = pos[:, :, :-1]
    h = h + F.gelu(pos).transpose(1, 2)

    H = spec["heads"]
    D = h.shape[-1]
    dh = D // H
    N = h.shape[0]
    # additive key-padding bias (finfo.min on padded keys; padded *queries* are still computed)
    bias = torch.zeros(N, 1, 1, T)
    if masked:
        bias.masked_fill_(~frame_mask[:, None, None, :], torch.finfo(torch.float32).min)
    if not stable:
        # Wav2Vec2Encoder (post-LN): the encoder LayerNorm sits behind the positional convolution
        h = F.layer_norm(h, (D,), state[_AM + "encoder.layer_norm.weight"], state[_AM + "encoder.layer_norm.bias"], eps)
    hidden_states: List[Tensor] = []
    for i in range(spec["layers"]):
        hidden_states.append(h)
        p = f"{_AM}encoder.layers.{i}."
        # pre-LN (Wav2Vec2EncoderLayerStableLayerNorm): attention on LayerNorm(h); post-LN (Wav2Vec2EncoderLayer): on h
        a = F.layer_norm(h, (D,), state[p + "layer_norm.weight"], state[p + "layer_norm.bias"], eps) if stable else h
        q = F.linear(a, state[p + "attention.q_proj.weight"], state[p + "attention.q_proj.bias"])
        kk = F.linear(a, state[p + "attention.k_proj.weight"], state[p + "attention.k_proj.bias"])
        v = F.linear(a, state[p + "attention.v_proj.weight"], state[p + "attention.v_proj.bias"])
        q = q.view(N, T, H, dh).transpose(1, 2)
        kk = kk.view(N, T, H, dh).transpose(1, 2)
        v = v.view(N, T, H, dh).transpose(1, 2)
        scores = torch.matmul(q, kk.transpose(2, 3)) * (dh ** -0.5) + bias
        attn = torch.matmul(torch.softmax(scores, dim=-1), v)
        attn = attn.transpose(1, 2).reshape(N, T, D)
        h = h + F.linear(attn, state[p + "attention.out_proj.weight"], state[p + "attention.out_proj.bias"])

        def ffn(x):
            x = F.gelu(F.linear(x, state[p + "feed_forward.intermediate_dense.weight"],
                                state[p + "feed_forward.intermediate_dense.bias"]))
            return F.linear(x, state[p + "feed_forward.output_dense.weight"], state[p + "feed_forward.output_dense.bias"])

        if stable:
            h = h + ffn(F.layer_norm(h, (D,), state[p + "final_layer_norm.weight"], state[p + "final_layer_norm.bias"], eps))
        else:
            h = F.layer_norm(h, (D,), state[p + "layer_norm.weight"], state[p + "layer_norm.bias"], eps)
            h = h + ffn(h)
            h = F.layer_norm(h, (D,), state[p + "final_layer_norm.weight"], state[p + "final_layer_norm.bias"], eps)
    if stable:
        h = F.layer_norm(h, (D,), state[_AM + "encoder.layer_norm.weight"], state[_AM + "encoder.layer_norm.bias"], eps)
    hidden_states.append(h)
    return hidden_states, frame_lengths, inter


# ----------------------------------------------------------------------------------------------------------------
# hierarchical multi-head projection
# ----------------------------------------------------------------------------------------------------------------
def category_offsets_from_table(train_table: Tensor) -> Tensor:
    """acoustic_model.py:196-207: offsets = cumsum([1, n_0, n_1, ...])[:-1] with n_f = max(column f) + 1."""
    num_categories = torch.cat((torch.zeros(1, dtype=torch.long), train_table.long().max(0).values)) + 1
    return num_categories.cumsum(0)[:-1]


def composed_embeddings(embedding: Tensor, tfi: Tensor, offsets: Tensor) -> Tensor:
    """acoustic_model.py:219-232: [E, P+1] matrix of the blank embedding (row 0) and per-phone sums."""
    indices = tfi.long() + offsets.unsqueeze(0)
    phones = F.embedding_bag(indices, embedding, mode="sum")
    blank = F.embedding_bag(torch.zeros(1, 1, dtype=torch.long), embedding, mode="sum")
    return torch.cat((blank, phones)).T


def sinusoidal_positions(positions: int, size: int) -> Tensor:
    """SinusoidalPositionEmbeddings.get_positions (acoustic_model.py:34-69): base_k = exp(-2k ln(1e4) / size) repeated
    for the (sin, cos) pair; even columns sin, odd columns cos."""
    component = torch.exp(torch.arange(0, size, 2, dtype=torch.float) * -(math.log(10000) / size))
    bases = torch.stack([component] * 2, 1).view(-1)
    pe = torch.arange(positions, dtype=torch.float).unsqueeze(1) * bases
    pe[:, 0::2] = torch.sin(pe[:, 0::2])
    pe[:, 1::2] = torch.cos(pe[:, 1::2])
    return pe.view(positions, -1)


def time_layer_forward(u: Tensor, frame_lengths: Tensor, state: Dict[str, Tensor], prefix: str, heads: int,
                       positional: bool) -> Tensor:
    """ProjectingMultiheadAttention.forward (acoustic_model.py:255-268) on time-major ``u`` [T, N, in]: Linear ->
    LayerNorm(eps 1e-5) -> (+ positions) -> nn.MultiheadAttention(C, heads) over time with the key-padding mask
    ``mask_sequence(lengths, inverse=True)``; dropout is the identity in eval mode."""
    x = F.linear(u, state[prefix + "input_projection.weight"], state[prefix + "input_projection.bias"])
    x = F.layer_norm(x, (x.shape[-1],), state[prefix + "layer_norm.weight"], state[prefix + "layer_norm.bias"], 1e-5)
    T, N, C = x.shape
    if positional:
        x = x + sinusoidal_positions(T, C).unsqueeze(1)
    qkv = F.linear(x, state[prefix + "attention.in_proj_weight"], state[prefix + "attention.in_proj_bias"])
    q, k, v = qkv.split(C, -1)
    dh = C // heads
    # [N, heads, T, dh]
    q, k, v = (t.reshape(T, N, heads, dh).permute(1, 2, 0, 3) for t in (q, k, v))
    scores = (q / math.sqrt(dh)) @ k.transpose(-1, -2)
    key_mask = torch.arange(T).unsqueeze(0) >= frame_lengths.unsqueeze(1)  # True = padded key
    scores = scores.masked_fill(key_mask[:, None, None, :], float("-inf"))
    o = torch.softmax(scores, -1) @ v  # [N, heads, T, dh]
    o = o.permute(2, 0, 1, 3).reshape(T, N, C)
    return F.linear(o, state[prefix + "attention.out_proj.weight"], state[prefix + "attention.out_proj.bias"])


def projection_forward(
    hidden_states: List[Tensor], state: Dict[str, Tensor], spec: Dict[str, Any], tfi: Optional[Tensor],
    category_offsets: Optional[Tensor] = None, frame_lengths: Optional[Tensor] = None,
) -> Dict[str, Tensor]:
    """HierarchicalProjection.forward in predict mode on time-major inputs ([T,N,D] each); returns raw logits."""
    outputs: Dict[str, Tensor] = {f"{OUTPUT}_{i}": h for i, h in enumerate(hidden_states)}
    outputs[OUTPUT] = hidden_states[-1]
    blanks = bool(spec.get("dependency_blanks", True))
    classes = spec["classes"]
    result: Dict[str, Tensor] = {}
    for ci in topological_order(classes):
        node = classes[ci]
        name, deps = node["name"], node["dependencies"]
        if len(deps) == 1 and OUTPUT_PATTERN.match(deps[0]):
            u = outputs[deps[0]]
        else:
            parts = []
            for d in deps:
                if OUTPUT_PATTERN.match(d):
                    parts.append(outputs[d])
                else:
                    logits = outputs[d] if blanks else outputs[d][..., BLANK_OFFSET:]
                    parts.append(torch.softmax(logits, -1))
            u = torch.cat(parts, -1)
        p = f"{_PROJ}{name}."
        layer = node.get("time_layer")
        if layer:
            if frame_lengths is None:
                raise ValueError("time-layer classifiers need the frame lengths (key-padding mask)")
            y = time_layer_forward(u, frame_lengths, state, p + "_time_distributed_layer.", int(layer.get("num_heads", 1)),
                                   bool(layer.get("positional_embeddings", False)))
        else:
            y = F.linear(u, state[p + "_time_distributed_layer.weight"], state[p + "_time_distributed_layer.bias"])
        emb_key = p + "_composition_layer._attribute_embeddings.weight"
        if emb_key in state:
            if tfi is None:
                raise ValueError("the oracle takes the inventory explicitly (the training table is a non-persistent buffer)")
            if category_offsets is None:
                raise ValueError("category_offsets required for the composition layer")
            composed = composed_embeddings(state[emb_key], tfi, category_offsets)
            y = (y @ composed) / torch.tensor(math.sqrt(composed.shape[0]))
        if name == PHONEME and spec.get("allophone_layer", False):
            # AllophoneMapping.forward(predict=True): the same tensor under both names, "phone" first
            result[PHONE] = y
            outputs[PHONE] = y
        result[name] = y
        outputs[name] = y
    return result


def predict(
    audio: Tensor, lengths: Tensor, state: Dict[str, Tensor], spec: Dict[str, Any], tfi: Optional[Tensor] = None,
    category_offsets: Optional[Tensor] = None, log_probabilities: bool = True, keep_intermediates: bool = False,
    padded: bool = False,
):
    """Estimator.predict restated.  Returns (outputs: name -> [T,N,C], frame lengths[, intermediates])."""
    with torch.inference_mode():
        hidden, frame_lengths, inter = wav2vec2_hidden_states(audio, lengths, state, spec, keep_intermediates, padded)
        time_major = [h.transpose(0, 1) for h in hidden]
        logits = projection_forward(time_major, state, spec, tfi, category_offsets, frame_lengths)
        if log_probabilities:
            logits = {k: F.log_softmax(v, -1) for k, v in logits.items()}
        if keep_intermediates:
            inter["hidden_states"] = hidden
            return logits, frame_lengths, inter
        return logits, frame_lengths


# ----------------------------------------------------------------------------------------------------------------
# greedy CTC
# ----------------------------------------------------------------------------------------------------------------
def greedy_ctc(log_emissions: Tensor, lengths: Tensor, blank: int = 0):
    """predictions.py:194-207 on batch-major [N,T,C]; returns per utterance (tokens, 1-based timesteps, score)."""
    values, indices = torch.max(log_emissions, dim=-1)
    out = []
    for i in range(indices.shape[0]):
        length = int(lengths[i])
        idx = indices[i, :length]
        decoded, sizes = torch.unique_consecutive(idx, return_counts=True)
        keep = decoded != blank
        timesteps = (sizes.cumsum(0) - sizes + 1)[keep]
        out.append((decoded[keep], timesteps, values[i, :length].sum()))
    return out
