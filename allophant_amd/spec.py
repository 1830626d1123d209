"""Model shape description for the Allophant acoustic-encoder forward path.

A *spec* is a plain dict (JSON-serialisable) carrying exactly the fields of the reference configuration that shape the
``Estimator.predict`` path:

* wav2vec 2.0 encoder shape (``facebook/wav2vec2-xls-r-300m`` for every released checkpoint,
  reference ``allophant/package_data/default_config.toml:34-37``),
* the classifier graph ``classes[].{name,size,dependencies}`` + ``dependency_blanks``
  (``allophant/config.py:624-712``, ``allophant/attribute_graph.py:17-41``),
* ``embedding_size`` of the compositional phoneme layer (``allophant/config.py:666-676``) or ``None``,
* ``allophone_layer``: whether predict-mode publishes ``"phone"`` next to ``"phoneme"``
  (``allophant/network/acoustic_model.py:161-167``).
"""
from __future__ import annotations

import copy
import re
from typing import Any, Dict, List, Optional, Sequence

OUTPUT = "OUTPUT"
OUTPUT_PATTERN = re.compile(r"^OUTPUT(?:_(\d+))?$")
PHONEME = "phoneme"
PHONE = "phone"
BLANK_OFFSET = 1

# The 36 articulatory attribute classifiers of the released multitask / hierarchical models
# (reference allophant/package_data/default_config.toml:61-99); every attribute is ternary (+, -, 0).
PHOIBLE_ATTRIBUTES = [
    "stress", "syllabic", "short", "long", "consonantal", "sonorant", "continuant", "delayedRelease", "approximant",
    "tap", "trill", "nasal", "lateral", "labial", "round", "labiodental", "coronal", "anterior", "distributed",
    "strident", "dorsal", "high", "low", "front", "back", "tense", "retractedTongueRoot", "advancedTongueRoot",
    "periodicGlottalSource", "epilaryngealSource", "spreadGlottis", "constrictedGlottis", "fortis",
    "raisedLarynxEjective", "loweredLarynxImplosive", "click",
]


def xlsr_300m_encoder() -> Dict[str, Any]:
    """wav2vec2-xls-r-300m hyper-parameters (cross-checked by the 315 437 696 parameter count)."""
    return {
        "conv_dim": 512,
        "conv_kernel": [10, 3, 3, 3, 3, 2, 2],
        "conv_stride": [5, 2, 2, 2, 2, 2, 2],
        "hidden": 1024,
        "layers": 24,
        "heads": 16,
        "ffn": 4096,
        "pos_kernel": 128,
        "pos_groups": 16,
        "eps": 1e-5,
        "do_normalize": True,
    }


def wav2vec2_base_encoder() -> Dict[str, Any]:
    """``facebook/wav2vec2-base`` hyper-parameters: the group-norm feature extractor (GroupNorm over time behind conv layer 0,
    bias-free convs), the post-LN encoder (``do_stable_layer_norm=False``), and a preprocessor with
    ``return_attention_mask=False`` -- the reference then calls the model with ``attention_mask=None``
    (acoustic_model.py:814,842-846).  ``facebook/wav2vec2-large`` is the same variant at 1024 / 24 / 16 / 4096."""
    return {
        "conv_dim": 512,
        "conv_kernel": [10, 3, 3, 3, 3, 2, 2],
        "conv_stride": [5, 2, 2, 2, 2, 2, 2],
        "hidden": 768,
        "layers": 12,
        "heads": 12,
        "ffn": 3072,
        "pos_kernel": 128,
        "pos_groups": 16,
        "eps": 1e-5,
        "do_normalize": True,
        "feat_extract_norm": "group",
        "conv_bias": False,
        "stable_layer_norm": False,
        "use_attention_mask": False,
    }


def xlsr_1b_encoder() -> Dict[str, Any]:
    """``facebook/wav2vec2-xls-r-1b`` [memory: hidden 1280, 48 layers, 16 heads (head_dim 80), ffn 5120; the 300M model's feature
    extractor and variant -- the hub is unreachable offline, ``checkpoint.check_encoder_against_state`` cross-checks the widths
    and counts against the weights]."""
    encoder = xlsr_300m_encoder()
    encoder.update(hidden=1280, layers=48, heads=16, ffn=5120)
    return encoder


def xlsr_2b_encoder() -> Dict[str, Any]:
    """``facebook/wav2vec2-xls-r-2b`` [memory: hidden 1920, 48 layers, 16 heads (head_dim 120), ffn 7680]."""
    encoder = xlsr_300m_encoder()
    encoder.update(hidden=1920, layers=48, heads=16, ffn=7680)
    return encoder


def tiny_encoder(layers: int = 2) -> Dict[str, Any]:
    """Reduced shape used by the committed golden vectors (same operator sequence as XLS-R)."""
    return {
        "conv_dim": 32,
        "conv_kernel": [10, 3, 3, 3, 3, 2, 2],
        "conv_stride": [5, 2, 2, 2, 2, 2, 2],
        "hidden": 128,  # head_dim 64 like XLS-R (other head dimensions: goldens g13 / g13b)
        "layers": layers,
        "heads": 2,
        "ffn": 256,
        "pos_kernel": 16,
        "pos_groups": 4,
        "eps": 1e-5,
        "do_normalize": True,
    }


def baseline_spec(encoder: Dict[str, Any], phonemes: int) -> Dict[str, Any]:
    """BASELINE config 1: a single shared ``phoneme`` classifier Linear(D -> P+1), no composition."""
    spec = copy.deepcopy(encoder)
    spec.update(classes=[{"name": PHONEME, "size": phonemes, "dependencies": [OUTPUT]}], dependency_blanks=True,
                embedding_size=None, allophone_layer=False, composition_categories=None)
    return spec


def multitask_spec(encoder: Dict[str, Any], attributes: Sequence[str] = PHOIBLE_ATTRIBUTES, embedding_size: int = 640,
                   train_phonemes: int = 64, n_features: int = 37, n_values: int = 3,
                   allophone_layer: bool = False) -> Dict[str, Any]:
    """BASELINE config 2/3/5: independent attribute heads + compositional phoneme head, all on ``OUTPUT``."""
    spec = copy.deepcopy(encoder)
    classes = [{"name": a, "size": n_values, "dependencies": [OUTPUT]} for a in attributes]
    classes.append({"name": PHONEME, "size": train_phonemes, "dependencies": [OUTPUT]})
    spec.update(classes=classes, dependency_blanks=True, embedding_size=embedding_size,
                allophone_layer=allophone_layer, composition_categories=[n_values] * n_features)
    return spec


def hierarchical_spec(encoder: Dict[str, Any], attributes: Sequence[str] = PHOIBLE_ATTRIBUTES,
                      embedding_size: int = 640, train_phonemes: int = 64, n_features: int = 37, n_values: int = 3,
                      dependency_blanks: bool = True, allophone_layer: bool = False) -> Dict[str, Any]:
    """BASELINE config 4: the phoneme head sees ``cat(OUTPUT, softmax(attribute logits)...)``."""
    spec = multitask_spec(encoder, attributes, embedding_size, train_phonemes, n_features, n_values, allophone_layer)
    spec["classes"][-1]["dependencies"] = [OUTPUT] + list(attributes)
    spec["dependency_blanks"] = dependency_blanks
    return spec


def frame_lengths(lengths: Sequence[int], spec: Dict[str, Any]) -> List[int]:
    """``floor((len - k) / s) + 1`` per conv layer (reference frontend.py:192-203, acoustic_model.py:832-835)."""
    out = []
    for length in lengths:
        for k, s in zip(spec["conv_kernel"], spec["conv_stride"]):
            length = (length - k) // s + 1
        out.append(length)
    return out


def evaluation_order(classes: Sequence[Dict[str, Any]]) -> List[int]:
    """Order in which the reference evaluates the classifier heads.

    ``AttributeGraph.sort`` (reference attribute_graph.py:124-199) runs Tarjan's SCC, which on an acyclic graph yields a
    node when its depth-first visit completes: roots in index order, edges = class dependencies in listed order
    (attribute_graph.py:67-74).  Raises ``ValueError`` on a dependency cycle like the reference's
    ``DependencyCycleError`` path.
    """
    index = {c["name"]: i for i, c in enumerate(classes)}
    if len(index) != len(classes):
        raise ValueError("Dependencies contain duplicate keys")  # acoustic_model.py:354-355
    order: List[int] = []
    state = [0] * len(classes)
    for root in range(len(classes)):
        if state[root]:
            continue
        stack = [(root, 0)]
        state[root] = 1
        while stack:
            node, edge = stack.pop()
            deps = [d for d in classes[node]["dependencies"] if not OUTPUT_PATTERN.match(d)]
            if edge < len(deps):
                stack.append((node, edge + 1))
                target = index[deps[edge]]
                if state[target] == 1:
                    raise ValueError(f"Dependency cycle detected at {classes[target]['name']}")
                if state[target] == 0:
                    state[target] = 1
                    stack.append((target, 0))
            else:
                state[node] = 2
                order.append(node)
    return order


def output_names(spec: Dict[str, Any]) -> List[str]:
    """Keys of ``Predictions.outputs`` in the order the reference inserts them (acoustic_model.py:515-522, 161-167)."""
    names = []
    for ci in evaluation_order(spec["classes"]):
        name = spec["classes"][ci]["name"]
        if name == PHONEME and spec.get("allophone_layer"):
            names.append(PHONE)
        names.append(name)
    return names


def spec_from_reference_model(model) -> Dict[str, Any]:
    """The spec of a LIVE reference ``Allophant`` module (``estimator.model`` of kgnlp/allophant), read from the objects the
    reference itself builds -- this is what the binding in INTEGRATION.md section 2 calls before ``amx_create``:

    * encoder shape: the ``transformers.Wav2Vec2Config`` of ``model._acoustic_model._model`` (acoustic_model.py:796-826)
      and the pre-processor's ``do_normalize`` (``_normalize``, :815);
    * classifier graph: ``model._classes`` (configuration order, :970) with the dependencies recorded in
      ``model._projection._ordered_nodes`` (:362-381), ``_dependency_blanks`` (:360);
    * head shapes from the modules of ``model._projection._layers`` (``HierarchicalClassifier``, :270-306): plain
      ``nn.Linear`` or ``ProjectingMultiheadAttention`` (time layer), composition layer, allophone layer.

    Duck-typed: no reference import."""
    acoustic = model._acoustic_model
    config = acoustic._model.config
    spec: Dict[str, Any] = {
        "conv_dim": int(config.conv_dim[0]),
        "conv_kernel": [int(k) for k in config.conv_kernel],
        "conv_stride": [int(k) for k in config.conv_stride],
        "hidden": int(config.hidden_size),
        "layers": int(config.num_hidden_layers),
        "heads": int(config.num_attention_heads),
        "ffn": int(config.intermediate_size),
        "pos_kernel": int(config.num_conv_pos_embeddings),
        "pos_groups": int(config.num_conv_pos_embedding_groups),
        "eps": float(config.layer_norm_eps),
        "do_normalize": bool(getattr(acoustic, "_normalize", True)),
    }
    if any(int(d) != spec["conv_dim"] for d in config.conv_dim):
        raise ValueError("feature-extractor layers of different widths are not supported")
    # the variant: the reference builds whatever `model_id` names (acoustic_model.py:775-826)
    spec["feat_extract_norm"] = str(getattr(config, "feat_extract_norm", "layer"))
    spec["conv_bias"] = bool(getattr(config, "conv_bias", True))
    spec["stable_layer_norm"] = bool(getattr(config, "do_stable_layer_norm", True))
    spec["use_attention_mask"] = bool(getattr(acoustic, "_use_attention_mask", True))  # preprocessor return_attention_mask
    projection = model._projection
    dependencies = {name: [d.name for d in deps] for name, deps in projection._ordered_nodes}
    embedding_size = None
    categories = None
    shared_phones = None
    classes: List[Dict[str, Any]] = []
    for name in model._classes:
        head = projection._layers[name]
        layer = head._time_distributed_layer
        entry: Dict[str, Any] = {"name": name, "dependencies": dependencies[name]}
        if hasattr(layer, "input_projection"):  # ProjectingMultiheadAttention (acoustic_model.py:237-268)
            out_features = int(layer.input_projection.out_features)
            entry["time_layer"] = {"num_heads": int(layer.attention.num_heads),
                                   "positional_embeddings": layer.positional_embeddings is not None}
        else:
            out_features = int(layer.out_features)
        composition = getattr(head, "_composition_layer", None)
        allophones = getattr(head, "_allophone_layer", None)
        if composition is not None:
            embedding_size = out_features
            table = composition._dense_feature_table
            offsets = composition._category_offsets.view(-1).tolist()
            rows = int(composition._attribute_embeddings.weight.shape[0])
            categories = [int(b - a) for a, b in zip(offsets, offsets[1:] + [rows])]
            entry["size"] = int(table.shape[0])
        else:
            entry["size"] = out_features - BLANK_OFFSET
        if allophones is not None:
            matrices = allophones._allophone_matrices
            shared_phones = int(matrices.shape[1]) - BLANK_OFFSET
            entry["size"] = int(matrices.shape[2]) - BLANK_OFFSET
        classes.append(entry)
    spec.update(classes=classes, dependency_blanks=bool(projection._dependency_blanks), embedding_size=embedding_size,
                allophone_layer=bool(getattr(projection, "_uses_allophone_mapping", False)),
                composition_categories=categories)
    if shared_phones is not None:
        spec["shared_phones"] = shared_phones
    validate(spec)
    return spec


def training_inventory_of_reference_model(model):
    """``_dense_feature_table - _category_offsets`` of the reference's composition layer (acoustic_model.py:191-217): the
    inventory ``predict(batch)`` falls back to without ``target_feature_indices``, as a ``[P, F]`` int64 tensor, or
    ``None`` for models without a composition layer."""
    for head in model._projection._layers.values():
        composition = getattr(head, "_composition_layer", None)
        if composition is not None:
            return (composition._dense_feature_table - composition._category_offsets).detach().cpu().long()
    return None


def validate(spec: Dict[str, Any]) -> None:
    """Mirrors the configuration errors the reference raises while building the projection (acoustic_model.py:353-466)."""
    names = [c["name"] for c in spec["classes"]]
    if len(set(names)) != len(names):
        raise ValueError("Dependencies contain duplicate keys")
    if any(OUTPUT_PATTERN.match(n) for n in names):
        raise ValueError(f"{OUTPUT!r} is a reserved keyword")
    uses_output = False
    for c in spec["classes"]:
        if not c["dependencies"]:
            raise ValueError("Each class projection requires a dependency")
        for d in c["dependencies"]:
            m = OUTPUT_PATTERN.match(d)
            if m:
                uses_output = True
                if m.group(1) is not None and int(m.group(1)) > spec["layers"]:
                    raise ValueError(f"{d} exceeds the number of encoder layers")
            elif d not in names:
                raise ValueError(f"unknown dependency {d!r}")
    if not uses_output:
        raise ValueError(f"At least one of the input layers requires {OUTPUT!r} as a dependency")
    for c in spec["classes"]:
        layer = c.get("time_layer")
        if layer:
            # `time_layer` = MultiheadAttentionConfig(num_heads, positional_embeddings) (config.py:596-610): the classifier
            # becomes Linear -> LayerNorm -> (+ sinusoidal positions) -> nn.MultiheadAttention (acoustic_model.py:237-268)
            heads = int(layer.get("num_heads", 1))
            if c["name"] == PHONEME and spec.get("embedding_size"):
                width = int(spec["embedding_size"])
            elif c["name"] == PHONEME and spec.get("allophone_layer"):
                width = int(spec.get("shared_phones", c["size"])) + BLANK_OFFSET
            else:
                width = int(c["size"]) + BLANK_OFFSET
            if heads < 1 or width % heads:
                raise ValueError("embed_dim must be divisible by num_heads")  # nn.MultiheadAttention's own assertion
    evaluation_order(spec["classes"])
    if spec["hidden"] % spec["heads"] != 0:
        raise ValueError("hidden must be divisible by heads")
    if (spec["hidden"] // spec["heads"]) % 8 or spec["hidden"] // spec["heads"] > 128:
        raise ValueError("head_dim (hidden / heads) must be a multiple of 8 and at most 128 (amx_create)")
    if spec["hidden"] > 2048 or spec["hidden"] % 8:
        raise ValueError("hidden must be a multiple of 8 and at most 2048 (amx_create)")
    if spec["hidden"] % spec["pos_groups"] or (spec["hidden"] // spec["pos_groups"]) % 8 or spec["hidden"] // spec["pos_groups"] > 128:
        raise ValueError("hidden / pos_groups must be a multiple of 8 and at most 128 (amx_create)")
    if spec.get("add_adapter"):
        # `Wav2Vec2Config.add_adapter`: the reference reads `.hidden_states` of the HF model (acoustic_model.py:839-847), the tuple of
        # ENCODER outputs -- the adapter's result only reaches `last_hidden_state`, which nothing reads, and `downsampled_lengths`
        # (acoustic_model.py:832-835) counts the conv extractor alone.  So an adapter changes no output as long as the classifiers
        # are built for the encoder's own width; with another `output_hidden_size` the reference sizes them for the adapter's width
        # (acoustic_model.py:822: `config.output_hidden_size or d_model`) and fails in its first classifier on hidden-size states
        out_size = spec.get("output_hidden_size")
        if out_size not in (None, spec["hidden"]):
            raise ValueError(f"add_adapter with output_hidden_size={out_size} != hidden={spec['hidden']}: the reference sizes its classifiers "
                             "for the adapter's width but feeds them encoder states (mat1 and mat2 shapes cannot be multiplied)")
    if spec.get("feat_extract_norm", "layer") not in ("layer", "group"):
        # transformers' own message (Wav2Vec2FeatureEncoder.__init__)
        raise ValueError(f"`config.feat_extract_norm` is {spec['feat_extract_norm']}, but has to be one of ['group', 'layer']")
