"""Drop-in façade over ``liballophant_amx``: the reference's ``Estimator`` / ``Batch`` / ``Predictions`` names, shapes and
error behaviour for the prediction path.

Mirrors (reference file:line):
  * ``Batch``                allophant/dataset_processing.py:49-85
  * ``Predictions``          allophant/network/acoustic_model.py:908-926
  * ``Estimator.predict``    allophant/estimator.py:1035-1046
  * ``Estimator.restore``    allophant/estimator.py:1085-1126 (checkpoint dict schema estimator.py:199-249)
  * ``GreedyCTCDecoder``     allophant/predictions.py:189-207

PyTorch is used only as plumbing (device memory for inputs/outputs, the current HIP stream); all arithmetic happens in the
HIP kernels behind the C ABI.
"""
from __future__ import annotations

import ctypes as C
import dataclasses
from dataclasses import dataclass
from typing import Any, Dict, List, NamedTuple, Optional, Sequence, Tuple

import torch
from torch import Tensor

from . import lib as _lib
from . import spec as _spec


@dataclass
class Batch:
    """``Batch(audio_features [N, L] f32 zero right-padded, lengths [N] i64 samples, language_ids [N])``."""

    audio_features: Tensor
    lengths: Tensor
    language_ids: Tensor

    def pin_memory(self):
        self.audio_features = self.audio_features.pin_memory()
        self.lengths = self.lengths.pin_memory()
        self.language_ids = self.language_ids.pin_memory()
        return self

    def to(self, device, non_blocking: bool = False, copy: bool = False):
        moved = self.__class__(
            self.audio_features.to(device, non_blocking=non_blocking, copy=copy),
            self.lengths.to(device, non_blocking=non_blocking, copy=copy),
            self.language_ids.to(device, non_blocking=non_blocking, copy=copy),
        )
        if getattr(self, "_padded", False):
            moved._padded = True  # a slice of a larger batch (parallel.shard_batch(keep_length=True)): L may exceed max(lengths)
        slot = getattr(self, "_pinned_slot", None)
        if slot is not None:
            if moved.audio_features.device.type == "cuda" and non_blocking:
                # an asynchronous copy out of a `batching.PinnedCollator` slot: the collator refills the slot only behind
                # this event
                event = torch.cuda.Event()
                event.record(torch.cuda.current_stream(moved.audio_features.device))
                slot[0].mark_in_flight(self, event)
            elif moved.audio_features.data_ptr() == self.audio_features.data_ptr():
                moved._pinned_slot = slot  # same storage (a no-op move): still the collator's slot
        return moved

    def cuda(self, non_blocking: bool = False, copy: bool = False):
        return self.to("cuda", non_blocking, copy)

    def size(self) -> int:
        return len(self)

    def __len__(self) -> int:
        return self.lengths.numel()

    def __repr__(self) -> str:
        return "{}(Features: ({}; {}))".format(self.__class__.__name__, self.audio_features.shape, self.audio_features.dtype)


@dataclass
class Predictions:
    """``outputs``: name -> [T, N, C] (log-)probabilities, time-major; ``lengths``: [N] int64 output frames."""

    outputs: Dict[str, Tensor]
    lengths: Tensor
    # flat device buffer the outputs are views of (kept for the on-device greedy decoder); not part of the reference API
    _flat: Optional[Tensor] = dataclasses.field(default=None, repr=False, compare=False)
    _geometry: Optional[Tuple[int, int]] = dataclasses.field(default=None, repr=False, compare=False)
    # the `target_feature_indices` the outputs were computed under (their phoneme block is that many phones wide)
    _inventory: Optional[Tensor] = dataclasses.field(default=None, repr=False, compare=False)
    # assembled by parallel.gather_flat_predictions: one status word per rank (device tensor; non-zero = that rank reported a
    # range error -- AMX_ERANGE -- with its shard).  Left on the device so that the gather never synchronises the host.
    _status: Optional[Tensor] = dataclasses.field(default=None, repr=False, compare=False)

    def __len__(self) -> int:
        return len(self.lengths)

    def check_ranks(self) -> None:
        """Data-parallel predictions only: raises ``FloatingPointError`` if a rank reported activations beyond the range of the
        planes with the shard it contributed (reads one small tensor back: call it where the host reads the results anyway)."""
        if self._status is not None:
            bad = [r for r, st in enumerate(self._status.tolist()) if st]
            if bad:
                raise FloatingPointError(f"rank(s) {bad} reported a range error (AMX_ERANGE) with their shard of this batch")

    def task_count(self) -> int:
        return len(self.outputs)


class CTCHypothesis(NamedTuple):
    """Field-compatible with ``torchaudio.models.decoder.CTCHypothesis`` as used by the reference decoder."""

    tokens: Tensor
    words: List[str]
    score: float
    timesteps: Tensor


class Decoded(NamedTuple):
    """Greedy CTC alignments of a batch, one row per output: ``tokens`` / ``timesteps`` ``[O, N, T]`` int64 of which the
    first ``counts[o, n]`` entries are valid, ``counts`` ``[O, N]`` int32, ``scores`` ``[O, N]`` fp32 (sum of the per-frame
    maxima, predictions.py:205).  Device tensors when produced by ``Estimator.greedy_decode_device``."""
    names: List[str]
    tokens: Tensor
    timesteps: Tensor
    counts: Tensor
    scores: Tensor

    def select(self, names: List[str]) -> "Decoded":
        rows = [self.names.index(name) for name in names]
        index = torch.tensor(rows, dtype=torch.long, device=self.tokens.device)
        return Decoded(list(names), self.tokens.index_select(0, index), self.timesteps.index_select(0, index),
                       self.counts.index_select(0, index), self.scores.index_select(0, index))

    def hypotheses(self) -> Dict[str, List[List[CTCHypothesis]]]:
        """Host form: per output and utterance ``[CTCHypothesis(tokens, [], score, timesteps)]`` like the reference."""
        counts_h, scores_h = self.counts.cpu(), self.scores.cpu()
        tokens_h, timesteps_h = self.tokens.cpu(), self.timesteps.cpu()
        result: Dict[str, List[List[CTCHypothesis]]] = {}
        for o, name in enumerate(self.names):
            hyps = []
            for n in range(counts_h.shape[1]):
                k = int(counts_h[o, n])
                hyps.append([CTCHypothesis(tokens_h[o, n, :k].clone(), [], float(scores_h[o, n]), timesteps_h[o, n, :k].clone())])
            result[name] = hyps
        return result


def _spec_to_structs(spec: Dict[str, Any], precision: str):
    cfg = _lib.AmxConfig()
    cfg.abi_version = _lib.AMX_ABI_VERSION
    n = len(spec["conv_kernel"])
    if n > _lib.AMX_MAX_CONV:
        raise ValueError("too many conv layers")
    cfg.n_conv = n
    cfg.conv_dim = spec["conv_dim"]
    for i in range(n):
        cfg.conv_kernel[i] = spec["conv_kernel"][i]
        cfg.conv_stride[i] = spec["conv_stride"][i]
    cfg.hidden, cfg.layers, cfg.heads, cfg.ffn = spec["hidden"], spec["layers"], spec["heads"], spec["ffn"]
    cfg.pos_kernel, cfg.pos_groups = spec["pos_kernel"], spec["pos_groups"]
    cfg.eps = spec["eps"]
    cfg.do_normalize = int(spec.get("do_normalize", True))
    cfg.dependency_blanks = int(spec.get("dependency_blanks", True))
    cfg.embedding_size = int(spec.get("embedding_size") or 0)
    cfg.allophone_layer = int(bool(spec.get("allophone_layer", False)))
    if precision not in _lib.PRECISIONS:
        raise ValueError(f"unknown precision {precision!r}; expected one of {sorted(_lib.PRECISIONS)}")
    cfg.precision = _lib.PRECISIONS[precision]
    norm = spec.get("feat_extract_norm", "layer")
    if norm not in ("layer", "group"):
        raise ValueError(f"`feat_extract_norm` is {norm}, but has to be one of ['group', 'layer']")  # transformers' own message
    cfg.feat_extract_norm = _lib.NORM_GROUP if norm == "group" else _lib.NORM_LAYER
    cfg.conv_bias = int(bool(spec.get("conv_bias", True)))
    cfg.stable_layer_norm = int(bool(spec.get("stable_layer_norm", True)))
    cfg.use_attention_mask = int(bool(spec.get("use_attention_mask", True)))

    classes = spec["classes"]
    index = {c["name"]: i for i, c in enumerate(classes)}
    descs = (_lib.AmxClassDesc * len(classes))()
    for i, c in enumerate(classes):
        if len(c["name"].encode()) >= _lib.AMX_NAME_LEN:
            raise ValueError(f"classifier name too long: {c['name']}")
        descs[i].name = c["name"].encode()
        descs[i].size = c["size"]
        composed = c["name"] == _spec.PHONEME and cfg.embedding_size
        allophone = c["name"] == _spec.PHONEME and cfg.allophone_layer
        out_classes = spec.get("shared_phones", c["size"]) if allophone else c["size"]
        descs[i].out_features = cfg.embedding_size if composed else out_classes + _spec.BLANK_OFFSET
        layer = c.get("time_layer")
        descs[i].time_heads = int(layer.get("num_heads", 1)) if layer else 0
        descs[i].time_positional = int(bool(layer.get("positional_embeddings", False))) if layer else 0
        deps = c["dependencies"]
        if len(deps) > _lib.AMX_MAX_DEPS:
            raise ValueError("too many dependencies")
        descs[i].n_deps = len(deps)
        for j, d in enumerate(deps):
            m = _spec.OUTPUT_PATTERN.match(d)
            if m:
                descs[i].deps[j] = _lib.DEP_OUTPUT if m.group(1) is None else _lib.dep_output_layer(int(m.group(1)))
            else:
                if d not in index:
                    raise ValueError(f"unknown dependency {d!r}")
                descs[i].deps[j] = index[d]
    return cfg, descs


class Estimator:
    """Prediction-side replacement of the reference ``Estimator`` running on one MI355X.

    ``precision``: ``"f16x3"`` (default; fp32-grade split-precision MFMA, meets the 1e-3 logit gate), ``"bf16x3"``,
    ``"f16"`` or ``"bf16"`` (single-plane throughput modes; error measured in tests/ and DESIGN.md).
    """

    def __init__(self, spec: Dict[str, Any], state_dict: Dict[str, Tensor], device: str | torch.device = "cuda:0",
                 precision: str = "f16x3"):
        _spec.validate(spec)
        self._spec = spec
        self._lib = _lib.load()
        self._device = torch.device(device)
        if self._device.type != "cuda":
            raise RuntimeError("allophant_amd runs on an MI355X only (device must be cuda:N); there is no CPU fallback")
        self._index = self._device.index if self._device.index is not None else torch.cuda.current_device()
        self._precision = precision
        cfg, descs = _spec_to_structs(spec, precision)
        keep = []
        tensors = (_lib.AmxTensor * len(state_dict))()
        for i, (k, v) in enumerate(state_dict.items()):
            t = v.detach().to("cpu", torch.float32).contiguous()
            keep.append(t)
            tensors[i].name = k.encode()
            tensors[i].data = C.cast(t.data_ptr(), C.POINTER(C.c_float))
            tensors[i].numel = t.numel()
        handle = C.c_void_p()
        code = self._lib.amx_create(C.byref(handle), self._index, C.byref(cfg), descs, len(descs), tensors, len(state_dict))
        _lib.check(self._lib, None, code)
        self._handle = handle
        self._classes = [c["name"] for c in spec["classes"]]
        self._inventory: Optional[Tensor] = None
        self._training_inventory: Optional[Tensor] = None
        cats = spec.get("composition_categories")
        self._category_offsets = None
        if spec.get("embedding_size"):
            if cats is None:
                raise ValueError("composition models need `composition_categories` (number of values per feature)")
            self._category_offsets = torch.tensor([1] + list(cats), dtype=torch.int64).cumsum(0)[:-1].contiguous()

    # -- reference-compatible surface ------------------------------------------------------------------------------
    @property
    def classes(self) -> List[str]:
        return self._classes

    @property
    def precision(self) -> str:
        return self._precision

    @property
    def device_bytes(self) -> int:
        return int(self._lib.amx_device_bytes(self._handle))

    @classmethod
    def restore(cls, checkpoint_or_path, device: str = "cuda:0", precision: str = "f16x3"):
        """Builds an estimator from a checkpoint dict in the reference ``Checkpoint`` schema (estimator.py:199-249) or a
        path to one saved with ``torch.save``.  Returns ``(estimator, attribute_indexer)`` like the reference
        (estimator.py:1085-1126): the indexer is an ``allophant_amd.phonetic.AttributeTable`` rebuilt from the table text
        embedded in ``phonetic_indexer_state`` (``composition_feature_matrix``, ``phoneme_inventory``,
        ``composition_features``), or ``None`` for checkpoints without one."""
        from .checkpoint import indexer_from_checkpoint, spec_from_checkpoint

        if not isinstance(checkpoint_or_path, dict):
            checkpoint_or_path = torch.load(checkpoint_or_path, map_location="cpu", weights_only=True)
        spec = spec_from_checkpoint(checkpoint_or_path)
        indexer, training = indexer_from_checkpoint(checkpoint_or_path)
        estimator = cls(spec, checkpoint_or_path["model_state"], device, precision)
        if spec.get("embedding_size") and indexer is not None and training:
            # `predict(batch)` without target_feature_indices falls back to the training inventory upstream
            # (`_dense_feature_table`, acoustic_model.py:214-221); it is a non-persistent buffer there, rebuilt from the
            # indexer at construction, so it is rebuilt from the embedded table here as well
            estimator.set_training_inventory(indexer.composition_feature_matrix(training))
        return estimator, indexer

    def set_training_inventory(self, target_feature_indices: Tensor) -> None:
        """The inventory ``predict(batch)`` uses when called without ``target_feature_indices`` (upstream: the
        ``_dense_feature_table`` buffer of ``EmbeddingCompositionLayer``, acoustic_model.py:214-221)."""
        if not self._spec.get("embedding_size"):
            raise ValueError("model has no embedding composition layer")
        self._training_inventory = target_feature_indices.detach().to("cpu", torch.int64).contiguous()

    def _set_inventory(self, tfi: Tensor) -> None:
        tfi_cpu = tfi.detach().to("cpu", torch.int64).contiguous()
        if self._inventory is not None and self._inventory.shape == tfi_cpu.shape and torch.equal(self._inventory, tfi_cpu):
            return
        if tfi_cpu.dim() != 2 or tfi_cpu.shape[1] != self._category_offsets.numel():
            raise ValueError(
                f"target_feature_indices must be [phones, {self._category_offsets.numel()}] (composition_feature_matrix)")
        with torch.cuda.device(self._device):
            stream = torch.cuda.current_stream(self._device).cuda_stream
            code = self._lib.amx_set_inventory(
                self._handle, C.cast(tfi_cpu.data_ptr(), C.POINTER(C.c_int64)), tfi_cpu.shape[0], tfi_cpu.shape[1],
                C.cast(self._category_offsets.data_ptr(), C.POINTER(C.c_int64)), C.c_void_p(stream))
        _lib.check(self._lib, self._handle, code)
        self._inventory = tfi_cpu

    def predict(self, batch: Batch, target_feature_indices: Optional[Tensor] = None, log_probabilities: bool = True,
                _keep_hidden: bool = False, _timing: bool = False, _no_pack: bool = False, _no_graph: bool = False,
                _out: Optional[Tensor] = None) -> Predictions:
        """``Estimator.predict`` (reference estimator.py:1035-1046).  ``_no_pack`` (test hook) keeps the padded row layout
        through the encoder layers of a ragged batch (``AMX_FLAG_NO_PACK``); ``_no_graph`` (test hook) enqueues the pass launch
        by launch (``AMX_FLAG_NO_GRAPH``); ``_out`` (test / benchmark hook) is a flat fp32 device buffer to write the outputs into.

        Safe by default: the reference computes in fp32; here an activation beyond the range of the fp16 planes turns into
        non-finite logits.  Such a batch raises ``FloatingPointError`` from the first ``predict`` / ``synchronize`` issued after
        the offending pass has finished on the GPU (no host synchronisation is added: see ``amx_forward``)."""
        if self._spec.get("embedding_size"):
            if target_feature_indices is None:
                if self._training_inventory is None:
                    raise ValueError(
                        "composition models need `target_feature_indices`: the training inventory table is a "
                        "non-persistent buffer upstream (acoustic_model.py:214-221); restore the estimator from a checkpoint "
                        "that embeds its attribute table, or call set_training_inventory()")
                target_feature_indices = self._training_inventory
            self._set_inventory(target_feature_indices)
        audio = batch.audio_features
        if audio.dim() != 2:
            raise ValueError("audio_features must be [N, L]")
        audio = audio.to(self._device, torch.float32).contiguous()
        lengths = batch.lengths.detach().to("cpu", torch.int64).contiguous()
        N, L = audio.shape
        if lengths.numel() != N:
            raise ValueError("lengths must have one entry per utterance")
        padded = bool(getattr(batch, "_padded", False))  # a block of a larger batch that keeps the global padded length
        if N > 0 and int(lengths.max()) != L and not (padded and int(lengths.max()) <= L):
            raise ValueError("the batch must be padded to exactly max(lengths) (reference utils.py:62-63, acoustic_model.py:765-767)")
        with torch.cuda.device(self._device):
            n_out = C.c_int()
            T = C.c_int64()
            total = C.c_int64()
            code = self._lib.amx_output_layout(self._handle, N, L, None, C.byref(n_out), C.byref(T), C.byref(total))
            _lib.check(self._lib, self._handle, code)
            descs = (_lib.AmxOutputDesc * n_out.value)()
            code = self._lib.amx_output_layout(self._handle, N, L, descs, C.byref(n_out), C.byref(T), C.byref(total))
            _lib.check(self._lib, self._handle, code)
            if _out is not None:
                if _out.dtype != torch.float32 or _out.device != self._device or _out.numel() < total.value or not _out.is_contiguous():
                    raise ValueError("_out must be a contiguous fp32 buffer on the estimator's device with room for every output")
                flat = _out.view(-1)[: total.value]
            else:
                flat = torch.empty(total.value, dtype=torch.float32, device=self._device)
            out_lengths = torch.empty(N, dtype=torch.int64)
            flags = 0 if log_probabilities else _lib.FLAG_RAW_LOGITS
            if _keep_hidden:
                flags |= _lib.FLAG_KEEP_HIDDEN
            if _timing:
                flags |= _lib.FLAG_TIMING
            if _no_pack:
                flags |= _lib.FLAG_NO_PACK
            if padded:
                flags |= _lib.FLAG_PADDED
            if _no_graph:
                flags |= _lib.FLAG_NO_GRAPH
            stream = torch.cuda.current_stream(self._device).cuda_stream
            n_max = int(self._lib.amx_max_utterances(self._handle, L))
            if N <= n_max:
                code = self._lib.amx_forward(
                    self._handle, C.c_void_p(audio.data_ptr()), C.cast(lengths.data_ptr(), C.POINTER(C.c_int64)), N, L,
                    C.c_void_p(flat.data_ptr()), C.cast(out_lengths.data_ptr(), C.POINTER(C.c_int64)), flags,
                    C.c_void_p(stream))
                _lib.check(self._lib, self._handle, code)
            else:
                # a plane of the batch would pass 4 GiB (32-bit plane offsets in the kernels): run it as slices of
                # utterances padded to the same L -- no operator mixes utterances, so the results are those of one call
                if n_max < 1:
                    raise ValueError(f"utterances of {L} samples are too long for one forward pass")
                # the slices add to one range-check count (`check_finite`): the first one restarts it like any forward pass,
                # the later ones continue it (AMX_FLAG_CONTINUE) -- launch-only, no host synchronisation
                blocks = {d.offset: d.classes for d in descs}
                for lo in range(0, N, n_max):
                    hi = min(N, lo + n_max)
                    n = hi - lo
                    part_total = sum(T.value * n * c for c in blocks.values())
                    part = torch.empty(part_total, dtype=torch.float32, device=self._device)
                    part_lengths = torch.empty(n, dtype=torch.int64)
                    slice_lengths = lengths[lo:hi].contiguous()  # named: must outlive the call that reads its storage
                    code = self._lib.amx_forward(
                        self._handle, C.c_void_p(audio[lo:hi].data_ptr()),
                        C.cast(slice_lengths.data_ptr(), C.POINTER(C.c_int64)), n, L,
                        C.c_void_p(part.data_ptr()), C.cast(part_lengths.data_ptr(), C.POINTER(C.c_int64)),
                        flags | _lib.FLAG_PADDED | (_lib.FLAG_CONTINUE if lo > 0 else 0), C.c_void_p(stream))
                    _lib.check(self._lib, self._handle, code)
                    out_lengths[lo:hi] = part_lengths
                    src = 0
                    for offset, c in blocks.items():  # blocks in output order: offsets ascend with the part's own
                        flat[offset: offset + T.value * N * c].view(T.value, N, c)[:, lo:hi] = \
                            part[src: src + T.value * n * c].view(T.value, n, c)
                        src += T.value * n * c
                    part.record_stream(torch.cuda.current_stream(self._device))
            # keep `audio` alive until the asynchronous kernels have consumed it
            flat.record_stream(torch.cuda.current_stream(self._device))
            audio.record_stream(torch.cuda.current_stream(self._device))
        self._geom = (N, int(T.value))
        outputs: Dict[str, Tensor] = {}
        for d in descs:
            c = d.classes
            outputs[d.name.decode()] = flat[d.offset: d.offset + T.value * N * c].view(T.value, N, c)
        return Predictions(outputs, out_lengths.to(batch.lengths.device), flat, (N, L), self._inventory)

    def greedy_decode_device(self, predictions: Predictions) -> "Decoded":
        """On-device ``GreedyCTCDecoder`` over every output of ``predictions`` (reference predictions.py:194-207 applied
        per classifier as in run.py:767-774); the result stays in HBM (``Decoded``: no host copy, no synchronisation), which
        is what the data-parallel path gathers instead of log-probabilities (``parallel.gather_decoded``)."""
        if predictions._flat is None or predictions._geometry is None:
            raise ValueError("predictions were not produced by this estimator")
        N, L = predictions._geometry
        if predictions._inventory is not None:
            # the block layout depends on the inventory size: decode under the inventory of THESE predictions, whatever
            # later predict() calls selected (a cached inventory is re-selected without device work)
            self._set_inventory(predictions._inventory)
        names = list(predictions.outputs.keys())
        T = next(iter(predictions.outputs.values())).shape[0]
        with torch.cuda.device(self._device):
            tokens = torch.empty(len(names), N, T, dtype=torch.int64, device=self._device)
            timesteps = torch.empty_like(tokens)
            counts = torch.empty(len(names), N, dtype=torch.int32, device=self._device)
            scores = torch.empty(len(names), N, dtype=torch.float32, device=self._device)
            stream = torch.cuda.current_stream(self._device).cuda_stream
            frame_lengths = predictions.lengths.detach().to("cpu", torch.int64).contiguous()
            code = self._lib.amx_greedy_ctc(
                self._handle, C.c_void_p(predictions._flat.data_ptr()),
                C.cast(frame_lengths.data_ptr(), C.POINTER(C.c_int64)), N, L, C.c_void_p(tokens.data_ptr()),
                C.c_void_p(timesteps.data_ptr()), C.c_void_p(counts.data_ptr()), C.c_void_p(scores.data_ptr()),
                C.c_void_p(stream))
            _lib.check(self._lib, self._handle, code)
        return Decoded(names, tokens, timesteps, counts, scores)

    def greedy_decode(self, predictions: Predictions) -> Dict[str, List[List[CTCHypothesis]]]:
        """``greedy_decode_device`` fetched to the host as the reference's hypothesis lists.  Only token ids / timesteps /
        scores cross PCIe."""
        return self.greedy_decode_device(predictions).hypotheses()

    def debug_fetch(self, what: str, index: int = 0) -> Tensor:
        """Test hook: intermediates of the last ``predict(..., _keep_hidden=True)`` as CPU fp32 tensors."""
        code_of = {"conv": 0, "hidden": 1, "logits": 2}
        n_t = self._last_geometry()
        if what == "conv":
            shape = (n_t[0], n_t[1], self._spec["conv_dim"])
        elif what == "hidden":
            shape = (n_t[0], n_t[1], self._spec["hidden"])
        else:
            shape = None
        ld = C.c_int64(0)
        if shape is None:
            buf = torch.empty(n_t[0] * n_t[1] * 4096, dtype=torch.float32)
        else:
            buf = torch.empty(shape, dtype=torch.float32)
        code = self._lib.amx_debug_fetch(self._handle, code_of[what], index, C.c_void_p(buf.data_ptr()), buf.numel(), C.byref(ld))
        _lib.check(self._lib, self._handle, code)
        if shape is None:
            return buf[: n_t[0] * n_t[1] * ld.value].view(n_t[0] * n_t[1], ld.value)
        return buf

    def _last_geometry(self) -> Tuple[int, int]:
        if not hasattr(self, "_geom"):
            raise RuntimeError("no forward pass yet")
        return self._geom

    def timing_fetch(self) -> Dict[str, Tuple[float, int]]:
        """Measurement hook: {kernel class: (total ms, launches)} of the ``predict(..., _timing=True)`` calls since the
        previous fetch, from HIP events recorded on the launch stream."""
        n = len(_lib.KERNEL_CLASSES)
        ms = (C.c_float * n)()
        launches = (C.c_int32 * n)()
        _lib.check(self._lib, self._handle, self._lib.amx_timing_fetch(self._handle, ms, launches, n))
        return {k: (float(ms[i]), int(launches[i])) for i, k in enumerate(_lib.KERNEL_CLASSES)}

    def graph_info(self) -> Tuple[int, int]:
        """(forward passes recorded into HIP graphs, passes replayed from one) so far -- ``amx_graph_info``."""
        if _lib.AMX_ABI_VERSION < 5:
            return 0, 0
        captures, replays = C.c_int64(0), C.c_int64(0)
        _lib.check(self._lib, self._handle, self._lib.amx_graph_info(self._handle, C.byref(captures), C.byref(replays)))
        return int(captures.value), int(replays.value)

    def pass_info(self) -> Dict[str, int]:
        """Which optional forms the last ``predict`` took (``amx_pass_info``): ``ln_fold`` 1 = LayerNorm folded into the encoder
        products, ``packed`` 0 / 1 / 2 = padded rows / packed layers / packed from the feature projection on, ``graph`` 0 / 1 / 2
        = eager / recorded / replayed, ``rows`` = frames the encoder layers worked on."""
        if _lib.AMX_ABI_VERSION < 6:
            return {}
        n = len(_lib.PASS_INFO)
        info = (C.c_int32 * n)()
        _lib.check(self._lib, self._handle, self._lib.amx_pass_info(self._handle, info, n))
        return {k: int(info[i]) for i, k in enumerate(_lib.PASS_INFO)}

    def check_finite(self) -> None:
        """Range check of the last ``predict`` (``amx_check_finite``; no upstream counterpart -- the reference computes in
        fp32): waits for the stream and raises ``FloatingPointError`` when a valid frame holds non-finite logits, i.e. an
        activation left the range of the fp16 planes (|x| <= 65504) or the audio was not finite.  Weights cannot cause it:
        they are packed under per-tensor power-of-two scales.  ``precision="bf16x3"`` has the range of fp32."""
        stream = torch.cuda.current_stream(self._device).cuda_stream
        _lib.check(self._lib, self._handle, self._lib.amx_check_finite(self._handle, C.c_void_p(stream), None))

    def synchronize(self) -> None:
        """Waits for the stream; raises ``FloatingPointError`` if a pass issued since the last report left the range of the
        planes (``amx_synchronize`` -> ``AMX_ERANGE``)."""
        stream = torch.cuda.current_stream(self._device).cuda_stream
        _lib.check(self._lib, self._handle, self._lib.amx_synchronize(self._handle, C.c_void_p(stream)))

    def close(self) -> None:
        if getattr(self, "_handle", None):
            self._lib.amx_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GreedyCTCDecoder:
    """``GreedyCTCDecoder`` with the reference's signature (predictions.py:189-207), decoding on the device:

        decoder = GreedyCTCDecoder()
        hypotheses = decoder(outputs.transpose(1, 0), model_outputs.lengths)     # run.py:767-774, README.md:120-125

    ``log_emissions`` is a ``[N, T, C]`` fp32 tensor on an MI355X (any strides with a unit class stride: the transposed
    view of a ``[T, N, C]`` output is read in place, no copy), ``lengths`` the ``[N]`` frame lengths.  Returns, per
    utterance, ``[CTCHypothesis(tokens, [], score, timesteps)]`` like the reference.  There is no CPU path.

    Also accepted, for the whole-prediction form: ``GreedyCTCDecoder(estimator)(predictions)`` decodes every output of a
    ``Predictions`` object in one launch (``Estimator.greedy_decode``)."""

    def __init__(self, blank_index_or_estimator=0, blank_index: Optional[int] = None):
        self._estimator = None
        if isinstance(blank_index_or_estimator, Estimator):
            self._estimator = blank_index_or_estimator
            self._blank_index = 0 if blank_index is None else int(blank_index)
            if self._blank_index != 0:
                raise ValueError("the CTC blank of model outputs is index 0 (config.py:555)")
        else:
            self._blank_index = int(blank_index_or_estimator)

    def __call__(self, log_emissions, lengths: Optional[Tensor] = None):
        if isinstance(log_emissions, Predictions):
            if self._estimator is None:
                raise ValueError("decoding a Predictions object needs GreedyCTCDecoder(estimator)")
            return self._estimator.greedy_decode(log_emissions)
        if lengths is None:
            raise TypeError("__call__() missing 1 required positional argument: 'lengths'")
        return greedy_ctc_decode(log_emissions, lengths, self._blank_index)


def greedy_ctc_decode(log_emissions: Tensor, lengths: Tensor, blank_index: int = 0) -> List[List[CTCHypothesis]]:
    """``GreedyCTCDecoder.__call__`` (reference predictions.py:194-207) through ``amx_greedy_ctc_emissions``."""
    if log_emissions.dim() != 3:
        raise ValueError("log_emissions must be [N, T, C]")
    if log_emissions.device.type != "cuda":
        raise RuntimeError("allophant_amd decodes on an MI355X only (log_emissions must be a cuda tensor); there is no CPU fallback")
    lib = _lib.load()
    device = log_emissions.device
    if log_emissions.dtype != torch.float32:
        log_emissions = log_emissions.float()
    if log_emissions.stride(2) != 1:
        log_emissions = log_emissions.contiguous()
    N, T, Cn = log_emissions.shape
    if not 0 <= blank_index < Cn:
        raise ValueError("blank_index out of range")
    if N == 0:
        return []
    with torch.cuda.device(device):
        frame_lengths = lengths.detach().to(device=device, dtype=torch.int32).contiguous()
        tokens = torch.empty(N, T, dtype=torch.int64, device=device)
        timesteps = torch.empty_like(tokens)
        counts = torch.empty(N, dtype=torch.int32, device=device)
        scores = torch.empty(N, dtype=torch.float32, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        index = device.index if device.index is not None else torch.cuda.current_device()
        code = lib.amx_greedy_ctc_emissions(
            index, C.c_void_p(log_emissions.data_ptr()), log_emissions.stride(0), log_emissions.stride(1),
            C.c_void_p(frame_lengths.data_ptr()), N, T, Cn, blank_index, C.c_void_p(tokens.data_ptr()),
            C.c_void_p(timesteps.data_ptr()), C.c_void_p(counts.data_ptr()), C.c_void_p(scores.data_ptr()),
            C.c_void_p(stream))
        _lib.check(lib, None, code)
        counts_h, scores_h, tokens_h, timesteps_h = counts.cpu(), scores.cpu(), tokens.cpu(), timesteps.cpu()
    result = []
    for n in range(N):
        k = int(counts_h[n])
        result.append([CTCHypothesis(tokens_h[n, :k].clone(), [], float(scores_h[n]), timesteps_h[n, :k].clone())])
    return result


def feature_decoders(indexer, beam_width: int = 1, feature_names=None, n_best: int = 1) -> Dict[str, GreedyCTCDecoder]:
    """``predictions.feature_decoders`` (reference predictions.py:245-254) for greedy decoding: one decoder per feature
    name of ``indexer`` (an ``AttributeTable`` or anything with ``feature_names``).  Beam search (``beam_width > 1``) is the
    torchaudio/flashlight CPU decoder upstream and is not part of this path."""
    if beam_width != 1 or n_best != 1:
        raise NotImplementedError("only greedy decoding (beam_width=1) runs on the device; beam search is out of scope")
    names = indexer.feature_names if feature_names is None else feature_names
    return {name: GreedyCTCDecoder() for name in names}
