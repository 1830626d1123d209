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


def tiny_encoder(layers: int = 2) -> Dict[str, Any]:
    """Reduced shape used by the committed golden vectors (same operator sequence as XLS-R)."""
    return {
        "conv_dim": 32,
        "conv_kernel": [10, 3, 3, 3, 3, 2, 2],
        "conv_stride": [5, 2, 2, 2, 2, 2, 2],
        "hidden": 128,  # head_dim 64 like XLS-R (the attention kernel is specialised for it)
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
