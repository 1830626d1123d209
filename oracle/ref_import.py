"""TEST INFRASTRUCTURE ONLY -- never imported by the product path (allophant_amd/).

Imports the *real* kgnlp/allophant reference from /root/reference in this (CPU-only, offline) container so that the
oracle restatement (oracle/allophant_oracle.py) can be pinned against it and golden vectors can be generated
(oracle/gen_golden.py).  The reference needs a number of non-numeric third-party packages that are absent from the
image (marshmallow*, toml, panphon, torchaudio, mashumaro, langcodes, zarr, stanza, epitran, phonemizer, mutagen,
tensorboard, and its own Rust extension ``allophant.phonemes``).  None of them executes on the hot path
(``Estimator.predict`` -> ``Allophant.forward``), so they are replaced by permissive stub modules; all numeric code
(torch, transformers' Wav2Vec2Model, the reference's own network/estimator modules) runs for real.

Nothing here travels to the GPU box: /root/reference does not exist there.  Only the committed fixtures under
tests/golden/ do.
"""
from __future__ import annotations

import dataclasses
import importlib.abc
import importlib.machinery
import importlib.metadata
import json
import os
import sys
import tempfile
import types
from typing import Any, Dict, List, Optional, Sequence

REFERENCE_ROOT = "/root/reference"

_STUB_TOPLEVEL = {
    "marshmallow", "marshmallow_oneofschema", "marshmallow_enum", "marshmallow_dataclass", "toml", "panphon",
    "torchaudio", "mashumaro", "langcodes", "mutagen", "zarr", "stanza", "epitran", "phonemizer", "tensorboard",
}
_STUB_FULL = {"allophant.phonemes", "torch.utils.tensorboard", "torch.utils.tensorboard.writer"}


class _PermissiveMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _make_dummy(name)

    def __getitem__(cls, item):
        return cls

    def __iter__(cls):
        return iter(())

    def __or__(cls, other):
        return cls

    def __ror__(cls, other):
        return cls


def _make_dummy(name: str = "Dummy"):
    class Dummy(metaclass=_PermissiveMeta):
        def __init__(self, *args, **kwargs):
            pass

        def __init_subclass__(cls, **kwargs):
            super().__init_subclass__()

        def __call__(self, *args, **kwargs):
            # decorators: return the decorated object unchanged
            if len(args) == 1 and not kwargs and (callable(args[0]) or isinstance(args[0], type)):
                return args[0]
            return Dummy()

        def __getattr__(self, item):
            if item.startswith("__") and item.endswith("__"):
                raise AttributeError(item)
            return Dummy()

        def __getitem__(self, item):
            return Dummy()

        def __iter__(self):
            return iter(())

    Dummy.__name__ = Dummy.__qualname__ = name
    return Dummy


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        full = f"{self.__name__}.{name}"
        if full in sys.modules:
            return sys.modules[full]
        value = _make_dummy(name)
        setattr(self, name, value)
        return value


class _StubLoader(importlib.abc.Loader):
    def create_module(self, spec):
        module = _StubModule(spec.name)
        module.__path__ = []  # behave like a package so sub-modules resolve through the finder
        return module

    def exec_module(self, module):
        name = module.__name__
        if name == "marshmallow_dataclass":
            def _dataclass(cls=None, **kwargs):
                def wrap(c):
                    c = dataclasses.dataclass(c)
                    c.Schema = _make_dummy("Schema")
                    return c
                return wrap if cls is None else wrap(cls)

            def _add_schema(cls=None, **kwargs):
                def wrap(c):
                    c.Schema = _make_dummy("Schema")
                    return c
                return wrap if cls is None else wrap(cls)

            module.dataclass = _dataclass
            module.add_schema = _add_schema
            module.class_schema = lambda *a, **k: _make_dummy("Schema")
        if name == "torchaudio.models.decoder":
            # torchaudio's public result type: NamedTuple(tokens, words, score, timesteps)
            import collections

            module.CTCHypothesis = collections.namedtuple("CTCHypothesis", ["tokens", "words", "score", "timesteps"])


class _StubFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        top = fullname.split(".")[0]
        if top in _STUB_TOPLEVEL or fullname in _STUB_FULL:
            return importlib.machinery.ModuleSpec(fullname, _StubLoader(), is_package=True)
        return None


_INSTALLED = False


def install() -> None:
    """Makes ``import allophant`` (the reference) work in this container."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}; goldens can only be generated in the build container")
    sys.meta_path.insert(0, _StubFinder())
    import pandas._typing
    import pandas.io.parsers.readers as readers

    if not hasattr(readers, "ReadCsvBuffer"):
        readers.ReadCsvBuffer = pandas._typing.ReadCsvBuffer
    real_version = importlib.metadata.version

    def _version(name):
        if name == "allophant":
            return "1.0.0"
        return real_version(name)

    importlib.metadata.version = _version
    sys.path.insert(0, REFERENCE_ROOT)
    _INSTALLED = True


def write_hf_model_dir(spec: Dict[str, Any], directory: str) -> str:
    """Writes config.json / preprocessor_config.json for a wav2vec2 of the given shape (see oracle spec)."""
    n = len(spec["conv_kernel"])
    config = {
        "model_type": "wav2vec2",
        "architectures": ["Wav2Vec2Model"],
        "hidden_size": spec["hidden"],
        "num_hidden_layers": spec["layers"],
        "num_attention_heads": spec["heads"],
        "intermediate_size": spec["ffn"],
        "hidden_act": "gelu",
        "feat_extract_activation": "gelu",
        "feat_extract_norm": spec.get("feat_extract_norm", "layer"),
        "conv_bias": bool(spec.get("conv_bias", True)),
        "conv_dim": [spec["conv_dim"]] * n,
        "conv_kernel": list(spec["conv_kernel"]),
        "conv_stride": list(spec["conv_stride"]),
        "num_feat_extract_layers": n,
        "do_stable_layer_norm": bool(spec.get("stable_layer_norm", True)),
        "num_conv_pos_embeddings": spec["pos_kernel"],
        "num_conv_pos_embedding_groups": spec["pos_groups"],
        "layer_norm_eps": spec["eps"],
        "mask_time_prob": 0.075,
        "hidden_dropout": 0.1, "attention_dropout": 0.1, "activation_dropout": 0.0, "feat_proj_dropout": 0.1,
        "layerdrop": 0.1, "final_dropout": 0.0,
        "vocab_size": 32,
    }
    if spec.get("add_adapter"):
        config.update(add_adapter=True, num_adapter_layers=int(spec.get("num_adapter_layers", 3)),
                      adapter_kernel_size=int(spec.get("adapter_kernel_size", 3)), adapter_stride=int(spec.get("adapter_stride", 2)),
                      output_hidden_size=spec.get("output_hidden_size") or spec["hidden"])
    preprocessor = {
        "do_normalize": True, "feature_extractor_type": "Wav2Vec2FeatureExtractor", "feature_size": 1,
        "padding_side": "right", "padding_value": 0, "return_attention_mask": bool(spec.get("use_attention_mask", True)),
        "sampling_rate": 16000,
    }
    os.makedirs(directory, exist_ok=True)
    with open(os.path.join(directory, "config.json"), "w") as f:
        json.dump(config, f)
    with open(os.path.join(directory, "preprocessor_config.json"), "w") as f:
        json.dump(preprocessor, f)
    return directory


class _FakeAttributes:
    """Stands in for ``ArticulatoryAttributes`` where the reference only reads ``dense_feature_table`` / ``len``."""

    def __init__(self, table):
        self.dense_feature_table = table

    def __len__(self):
        return int(self.dense_feature_table.shape[0])

    def subset(self, *_args, **_kwargs):
        return self


class _FakeLanguageAllophones:
    def __init__(self, shared_phone_count: int, phoneme_count: int, n_languages: int):
        self.languages = [f"l{i}" for i in range(n_languages)]
        self.shared_phones = list(range(shared_phone_count))
        # language index -> {phoneme index: [allophone (shared phone) indices]}
        self.allophones = {
            i: {p: [(p + i) % shared_phone_count] for p in range(phoneme_count)} for i in range(n_languages)
        }


class _FakeIndexer:
    def __init__(self, train_table, allophones: Optional[_FakeLanguageAllophones]):
        import numpy as np

        self.full_attributes = _FakeAttributes(train_table)
        self.phonemes = np.arange(train_table.shape[0])
        self.composition_features = [f"f{i}" for i in range(train_table.shape[1])]
        self.language_allophones = allophones
        self.allophone_data = types.SimpleNamespace(shared_phone_indexer=_FakeAttributes(train_table))


def build_reference_estimator(spec: Dict[str, Any], train_feature_table=None):
    """Builds the reference ``Estimator`` wrapping ``Allophant`` for an oracle spec.  Returns (estimator, model)."""
    install()
    import torch
    from allophant.attribute_graph import AttributeGraph, AttributeNode
    from allophant.config import (
        EmbeddingCompositionConfig, PhonemeLayerType, ProjectionConfig, ProjectionEntryConfig,
    )
    from allophant.estimator import Estimator
    from allophant.network.acoustic_model import Allophant, Wav2Vec2AcousticModel

    tmp = tempfile.mkdtemp(prefix="amx_hf_")
    write_hf_model_dir(spec, tmp)
    acoustic = Wav2Vec2AcousticModel(tmp, 16000, load_pretrained_weights=False)
    # the reference slices encoder layers through `encoder._layers` (acoustic_model.py:800-802); reproduce the call
    # it would make from `Allophant.from_config` so that state_dict aliasing matches.
    from allophant.network.acoustic_model import _highest_specific_output_layer

    from allophant.config import MultiheadAttentionConfig

    def time_config(c):
        layer = c.get("time_layer")
        if not layer:
            return None
        return MultiheadAttentionConfig(int(layer.get("num_heads", 1)), bool(layer.get("positional_embeddings", False)))

    nodes = [
        AttributeNode(c["name"], c["size"], time_config(c), list(c["dependencies"])) for c in spec["classes"]
    ]
    graph = AttributeGraph(nodes)
    highest = _highest_specific_output_layer(graph)
    if highest is not None:
        acoustic._model.encoder._layers = acoustic._model.encoder.layers[:highest]

    composition = spec.get("embedding_size")
    allophone_layer = bool(spec.get("allophone_layer", False))
    projection = ProjectionConfig(
        classes=[ProjectionEntryConfig(c["name"], list(c["dependencies"]), time_config(c)) for c in spec["classes"]],
        phoneme_layer=PhonemeLayerType.ALLOPHONES if allophone_layer else PhonemeLayerType.SHARED,
        dependency_blanks=bool(spec.get("dependency_blanks", True)),
        embedding_composition=EmbeddingCompositionConfig(composition) if composition else None,
    )
    indexer = None
    if composition or allophone_layer:
        phoneme_size = next(c["size"] for c in spec["classes"] if c["name"] == "phoneme")
        if train_feature_table is None:
            raise ValueError("train_feature_table required")
        table = torch.as_tensor(train_feature_table).long()
        allophones = None
        if allophone_layer:
            allophones = _FakeLanguageAllophones(table.shape[0], phoneme_size, 2)
        indexer = _FakeIndexer(table, allophones)
    model = Allophant(acoustic, graph, 1, projection, indexer)
    model.eval()
    estimator = Estimator(None, 1, 16000, graph, model, {})
    return estimator, model


def reference_predict(estimator, audio, lengths, tfi=None, log_probabilities=True):
    install()
    import torch
    from allophant.batching import Batch

    batch = Batch(torch.as_tensor(audio), torch.as_tensor(lengths).long(), torch.zeros(len(lengths), dtype=torch.long))
    return estimator.predict(batch, None if tfi is None else torch.as_tensor(tfi).long(), log_probabilities)
