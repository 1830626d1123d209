"""Utterance-level data parallelism over the GPUs of one node (one process per GPU, ``torch.distributed``; backend
``"nccl"`` is RCCL over xGMI on ROCm, ``"gloo"`` on CPU for the tests).

The prediction path is embarrassingly parallel over utterances: no operator mixes batch rows -- for
the wav2vec 2.0 variants without the attention mask or with the group-norm extractor only as long as the blocks keep the
padded length of the global batch (``padding_sensitive`` / ``shard_batch(keep_length=True)``) --, so a batch is cut into contiguous blocks of utterances, every rank runs the full forward pass on its block with a full
weight replica, and the only exchange is one gather of per-frame log-probabilities (plus the frame lengths) to rank 0,
where the reference's decode loop consumes ``Predictions.outputs`` (run.py:765-774).  The reference itself is
single-device (no DDP / NCCL anywhere upstream) -- this module is new functionality with no upstream counterpart.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from .estimator import Batch, CTCHypothesis, Decoded, Predictions


def shard_bounds(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous, balanced utterance blocks: the first ``n % world`` ranks get one utterance more."""
    base, extra = divmod(n, world)
    bounds = []
    start = 0
    for r in range(world):
        size = base + (1 if r < extra else 0)
        bounds.append((start, start + size))
        start += size
    return bounds


def padding_sensitive(spec) -> bool:
    """True for the encoder variants whose results on VALID frames depend on the padded length of the batch tensor: the
    group-norm feature extractor takes its GroupNorm statistics over every frame of the padded tensor, and a model called
    without the attention mask (``use_attention_mask=False``) attends to every padded frame.  For the released Allophant
    checkpoints (XLS-R: layer norm + attention mask) padding never reaches a valid frame and this is False."""
    return spec.get("feat_extract_norm", "layer") == "group" or not spec.get("use_attention_mask", True)


def _keep_length(spec, keep_length: Optional[bool]) -> bool:
    """The padded length a shard keeps: an explicit ``keep_length`` wins, otherwise the model's ``spec`` decides
    (``padding_sensitive``).  Neither given is an error: the silent default used to be "re-pad", which is wrong for the
    padding-sensitive variants by 0.5-0.9 in log-probability (round-5 advisor finding)."""
    if keep_length is not None:
        return bool(keep_length)
    if spec is None:
        raise TypeError("shard_batch / data_parallel_predict need the model's `spec` (or an explicit `keep_length`): whether a "
                        "shard may be re-padded to its own longest utterance depends on the encoder variant (padding_sensitive)")
    return padding_sensitive(spec)


def shard_batch(batch: Batch, rank: int, world: int, spec=None, keep_length: Optional[bool] = None) -> Optional[Batch]:
    """Block ``rank`` of ``batch``.  For the released (XLS-R) form the block is re-padded to its own longest utterance (the
    reference requires ``L == max(lengths)``, utils.py:62-63); for ``padding_sensitive(spec)`` variants it keeps the padded
    length of the GLOBAL batch (the block is marked as a slice of a larger batch and ``Estimator.predict`` runs it with
    ``AMX_FLAG_PADDED``), because their single-device result depends on that length.  ``spec`` is the model's spec
    (``Estimator.spec``); ``keep_length`` overrides what it implies.  Returns ``None`` for an empty block."""
    keep = _keep_length(spec, keep_length)
    lo, hi = shard_bounds(len(batch), world)[rank]
    if hi <= lo:
        return None
    lengths = batch.lengths[lo:hi]
    if keep:
        local = Batch(batch.audio_features[lo:hi].contiguous(), lengths, batch.language_ids[lo:hi])
        local._padded = True
        return local
    local_max = int(lengths.max())
    return Batch(batch.audio_features[lo:hi, :local_max].contiguous(), lengths, batch.language_ids[lo:hi])


class RankError(FloatingPointError):
    """Raised on the destination rank when another rank reported a range error with its shard (and on that rank itself)."""


def predict_reporting(predict: Callable[[Batch], Predictions], batch: Batch, retries: int = 8):
    """``predict(batch)`` for a rank of a data-parallel job: returns ``(predictions, error)``.

    ``Estimator.predict`` raises ``FloatingPointError`` (``AMX_ERANGE``) when an EARLIER pass left the range of the planes, before
    anything of the current call has been enqueued.  Raised on one rank only, in front of a collective, that leaves every
    other rank blocked in the gather (round-5 advisor finding).  Here the report is caught, the pass is issued again (the
    report was consumed by the call that raised it), and the error is handed to the caller, which first joins the step's
    collectives -- its status travels with the frame lengths -- and raises afterwards."""
    error: Optional[BaseException] = None
    for _ in range(retries):
        try:
            return predict(batch), error
        except FloatingPointError as exc:
            error = exc
    raise error  # every retry reported again: nothing sensible can be sent


def _raise_reported(statuses: List[int], own: Optional[BaseException]) -> None:
    bad = [r for r, st in enumerate(statuses) if st]
    if own is not None:
        raise own
    if bad:
        raise RankError(f"rank(s) {bad} reported an error with their shard and raise it themselves: activations beyond the range "
                        "of the 16-bit planes in an earlier pass (AMX_ERANGE; precision='bf16x3' has the range of fp32), or a shard "
                        "with more frames than the stated frame count.  The gathered predictions of that pass are not to be trusted")


def gather_predictions(local: Optional[Predictions], names_and_classes: List[Tuple[str, int]], total_utterances: int,
                       device: torch.device, dst: int = 0, group=None,
                       aliases: Optional[Dict[str, str]] = None, frames: Optional[int] = None,
                       error: Optional[BaseException] = None) -> Optional[Predictions]:
    """Gathers per-rank ``Predictions`` to ``dst``: one ``gather`` of a packed ``[T_max, n_max, sum(C)]`` fp32 block per
    rank and one of the int64 frame lengths.  Returns the assembled ``Predictions`` ([T_max, N, C] per output, frames
    beyond an utterance's length are zero) on ``dst`` and ``None`` elsewhere.  ``frames``: the padded frame count of the
    global batch when the caller knows it (``spec.frame_lengths([L], spec)``): the ranks then need not agree on it with an
    ``all_reduce`` whose result the host reads back.  ``error``: what ``predict_reporting`` caught on this rank -- it travels
    as a status word behind the frame lengths, the rank raises it AFTER the collectives and ``dst`` raises ``RankError``."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bounds = shard_bounds(total_utterances, world)
    n_max = max(hi - lo for lo, hi in bounds)
    widths = [c for _, c in names_and_classes]
    total_c = sum(widths)

    # agree on the padded frame count (ranks may have different local max lengths)
    t_local = 0 if local is None else next(iter(local.outputs.values())).shape[0]
    if frames is not None:
        t_max = int(frames)
        if t_local > t_max and error is None:
            # a caller's mistake -- but raised HERE, on one rank, in front of the collectives, it would strand the others: the
            # shard joins them cropped, with its status set, and raises behind them like a range report
            error = ValueError(f"a shard has {t_local} frames, more than the stated {frames} of the global batch")
    else:
        t_tensor = torch.tensor([t_local], dtype=torch.int64, device=device)
        dist.all_reduce(t_tensor, op=dist.ReduceOp.MAX, group=group)
        t_max = int(t_tensor.item())

    packed = torch.zeros(t_max, n_max, total_c, dtype=torch.float32, device=device)
    lengths = torch.zeros(n_max + 1, dtype=torch.int64, device=device)  # [n_max] = this rank's status
    if error is not None:
        lengths[n_max] = 1
    if local is not None:
        n_local = len(local.lengths)
        col = 0
        for (name, c) in names_and_classes:
            out = local.outputs[name]
            packed[: min(out.shape[0], t_max), :n_local, col: col + c] = out[:t_max]
            col += c
        lengths[:n_local] = local.lengths.to(device)
    gathered_p = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
    gathered_l = [torch.empty_like(lengths) for _ in range(world)] if rank == dst else None
    dist.gather(packed, gathered_p, dst=dst, group=group)
    dist.gather(lengths, gathered_l, dst=dst, group=group)
    if rank != dst:
        _raise_reported([], error)
        return None
    _raise_reported([int(gathered_l[r][n_max]) for r in range(world)], error)
    outputs: Dict[str, Tensor] = {}
    all_lengths = torch.cat([gathered_l[r][: hi - lo] for r, (lo, hi) in enumerate(bounds)])
    col = 0
    for (name, c) in names_and_classes:
        parts = [gathered_p[r][:, : hi - lo, col: col + c] for r, (lo, hi) in enumerate(bounds) if hi > lo]
        outputs[name] = torch.cat(parts, dim=1)
        col += c
    if aliases:
        # e.g. {"phone": "phoneme"}: the allophone pass-through publishes one tensor under two names upstream
        ordered: Dict[str, Tensor] = {}
        for name, tensor in outputs.items():
            for alias, target in aliases.items():
                if target == name:
                    ordered[alias] = tensor
            ordered[name] = tensor
        outputs = ordered
    return Predictions(outputs, all_lengths.cpu())


def data_parallel_predict(predict: Callable[[Batch], Predictions], batch: Batch, outputs: List[Tuple[str, int]],
                          device: torch.device, dst: int = 0, group=None,
                          aliases: Optional[Dict[str, str]] = None, spec=None,
                          keep_length: Optional[bool] = None) -> Optional[Predictions]:
    """One data-parallel ``predict`` over the ranks of ``group``: every rank takes its contiguous block of utterances
    (``shard_batch``), runs ``predict`` on it (e.g. ``lambda b: estimator.predict(b.to(device), tfi)``) and the
    log-probabilities are gathered to ``dst`` (``gather_predictions``), which gets the ``Predictions`` of the whole batch;
    the other ranks get ``None``.  ``outputs`` lists the distinct outputs as (name, classes) in output order and
    ``aliases`` the names that share storage with one of them (see ``unique_outputs``): a rank whose block is empty has no
    local prediction to read them from.  ``spec``: the model's spec (``Estimator.spec``) -- decides whether the blocks keep
    the padded length of the global batch (``padding_sensitive``; ``keep_length`` overrides).  A range report
    (``FloatingPointError``) on one rank does not strand the others: see ``predict_reporting``."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    local_batch = shard_batch(batch, rank, world, spec=spec, keep_length=keep_length)
    local, error = predict_reporting(predict, local_batch) if local_batch is not None else (None, None)
    return gather_predictions(local, outputs, len(batch), device, dst=dst, group=group, aliases=aliases, error=error)


class PendingGather:
    """Handle of an asynchronous ``gather_flat_predictions``: the collectives are in flight on the backend's own stream
    (RCCL: beside the compute stream, so the gather of batch k overlaps the forward pass of batch k + 1); ``wait()`` makes
    the current stream wait for them and returns the assembled ``Predictions`` on the destination rank, ``None`` elsewhere.
    The handle keeps the send buffers alive until then."""

    def __init__(self, works, keep, assemble, error: Optional[BaseException] = None):
        self._works, self._keep, self._assemble, self._error = works, keep, assemble, error

    def wait(self) -> Optional[Predictions]:
        for work in self._works:
            work.wait()
        result = self._assemble() if self._assemble is not None else None
        error, self._error = self._error, None
        self._works, self._keep, self._assemble = [], None, None
        if error is not None:
            raise error  # this rank's own range report: it has joined the collectives of the step, now it says so
        return result


class _CompletedGather:
    """A gather that has already completed (the padded fallback of ``DataParallelRunner``), parked in the runner's pending
    slot so that its result is handed out by the NEXT ``step()`` / ``drain()``, after the result of the step before it."""

    counted = True  # `DataParallelRunner.completed` already includes it

    def __init__(self, result: Optional[Predictions]):
        self._result = result

    def wait(self) -> Optional[Predictions]:
        result, self._result = self._result, None
        return result


def gather_flat_predictions(local: Predictions, device: torch.device, dst: int = 0, group=None, async_op: bool = False,
                            error: Optional[BaseException] = None):
    """Fast path for equal-shaped shards (every rank ran the same ``(N, L)`` geometry, e.g. the weak-scaling benchmark):
    the outputs of ``Estimator.predict`` are views of one flat fp32 block, so a single ``gather`` of that block (plus one
    of the frame lengths) moves everything; rank ``dst`` re-assembles ``[T, world * N, C]`` per output with one strided
    copy each.  Returns the assembled ``Predictions`` on ``dst`` and ``None`` elsewhere -- or, with ``async_op=True``, a
    ``PendingGather`` whose ``wait()`` returns that.  ``error`` (``predict_reporting``): sent as a status word behind the frame
    lengths; this rank raises it when the gather is waited for, ``dst`` raises ``RankError``.  With ``Predictions.lengths`` on the
    host (the façade's default) the status of the OTHER ranks is read on ``dst`` in ``wait()``, which has synchronised anyway."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    flat = local._flat
    if flat is None:
        raise ValueError("gather_flat_predictions needs Predictions produced by Estimator.predict (one flat output block)")
    n = len(local.lengths)
    stacked = torch.empty(world, flat.numel(), dtype=flat.dtype, device=device) if rank == dst else None
    w1 = dist.gather(flat, [stacked[r] for r in range(world)] if rank == dst else None, dst=dst, group=group, async_op=True)
    lens = torch.cat((local.lengths.to(device), torch.tensor([0 if error is None else 1], dtype=local.lengths.dtype, device=device)))
    all_lens = torch.empty(world, n + 1, dtype=lens.dtype, device=device) if rank == dst else None
    w2 = dist.gather(lens, [all_lens[r] for r in range(world)] if rank == dst else None, dst=dst, group=group, async_op=True)

    def assemble() -> Predictions:
        outputs: Dict[str, Tensor] = {}
        done: Dict[int, str] = {}
        base = flat.data_ptr()
        for name, out in local.outputs.items():
            key = out.data_ptr()
            if key in done:  # aliases ("phone" / "phoneme") share storage upstream too (acoustic_model.py:161-167)
                outputs[name] = outputs[done[key]]
                continue
            t, _, c = out.shape
            first = (key - base) // flat.element_size()
            block = stacked[:, first: first + t * n * c].view(world, t, n, c)
            outputs[name] = block.permute(1, 0, 2, 3).reshape(t, world * n, c)
            done[key] = name
        # frame lengths stay on `device`: a host copy here would serialise the caller with the stream every step
        return Predictions(outputs, all_lens[:, :n].reshape(-1), _status=all_lens[:, n])

    pending = PendingGather([w1, w2], (flat, lens, stacked, all_lens), assemble if rank == dst else None, error)
    return pending if async_op else pending.wait()


class DataParallelRunner:
    """The per-step data path of data-parallel prediction as ``bench.py --gpus N`` times it (BASELINE config 3): every rank
    predicts its shard, the log-probabilities go to rank ``dst`` in ONE flat gather per step (equal-shaped shards, e.g.
    32 utterances over 2 / 4 / 8 ranks) or the general padded gather (ragged or empty shards).  With ``overlap`` the gather
    of step k stays in flight on the backend's stream under the forward pass of step k + 1 and is completed before step
    k + 2 is enqueued; ``drain()`` completes the last one.  ``step`` returns, on ``dst``, the assembled global
    ``Predictions`` of the most recently COMPLETED gather (the previous step when overlapping), ``None`` elsewhere / before
    the first completion.

    Every step's predictions come out exactly once and in step order: when an overlapped step has to fall back to the padded
    gather while the previous step's flat gather is still in flight, ``step`` returns the PREVIOUS step's result and parks
    its own (already complete) result, which the next ``step()`` / ``drain()`` returns.

    ``flat=True`` is the fast path for shards of one common ``(N, L)`` geometry; every step first agrees on that with one
    small ``all_reduce`` (a rank whose shard is empty or shaped differently would otherwise leave the others hanging in a
    mismatched ``gather``) and falls back to the padded gather when the shards differ.  The padded gather needs
    ``total_utterances`` (constructor default, or per step: ``step(batch, total_utterances=...)`` -- the last batch of a
    corpus is usually smaller) and, because a rank with an empty shard (``local_batch=None``) has no local prediction to read
    them from, ``outputs`` = the distinct outputs as (name, classes) in output order and ``aliases`` (see
    ``unique_outputs``) -- or at least one earlier non-empty step on this rank to learn them from.  The agreement reads
    two integers back per step, i.e. the host waits for the step it has just enqueued; a caller that has established equal
    shards on the host already (``bench.py``: ``shard_bounds`` of equal-length utterances) passes ``verify_shapes=False``
    and keeps the host running ahead of the GPU."""

    def __init__(self, predict: Callable[[Batch], Predictions], device: torch.device, dst: int = 0, group=None,
                 overlap: bool = True, flat: bool = True, total_utterances: Optional[int] = None,
                 outputs: Optional[List[Tuple[str, int]]] = None, aliases: Optional[Dict[str, str]] = None,
                 verify_shapes: bool = True):
        self._predict, self._device, self._dst, self._group = predict, device, dst, group
        self._verify = verify_shapes
        self._overlap, self._flat, self._total = overlap, flat, total_utterances
        self._outputs, self._aliases = outputs, aliases
        self._pending: Optional[PendingGather] = None
        self.completed = 0  # gathers completed so far
        if not flat and total_utterances is None:
            raise ValueError("the padded gather (flat=False) needs total_utterances, the size of the global batch")

    def _same_geometry(self, local: Optional[Predictions]) -> bool:
        """True when every rank holds a flat block of one common size and utterance count (one tiny all_reduce)."""
        size = -1 if local is None or local._flat is None else local._flat.numel()
        count = -1 if local is None else len(local.lengths)
        probe = torch.tensor([size, -size, count, -count], dtype=torch.int64, device=self._device)
        dist.all_reduce(probe, op=dist.ReduceOp.MAX, group=self._group)
        lo_size, lo_count = -int(probe[1]), -int(probe[3])
        return lo_size >= 0 and lo_size == int(probe[0]) and lo_count == int(probe[2])

    def _padded(self, local: Optional[Predictions], total: Optional[int], error: Optional[BaseException] = None) -> Optional[Predictions]:
        if local is not None and self._outputs is None:
            self._outputs, self._aliases = unique_outputs(local)
        if self._outputs is None:
            raise ValueError("a rank with an empty shard needs `outputs` (and `aliases`): it has no local prediction to take "
                             "the output names and widths from")
        if total is None:
            raise ValueError("shards of different shapes need the padded gather: pass total_utterances")
        result = gather_predictions(local, self._outputs, total, self._device, dst=self._dst, group=self._group,
                                    aliases=self._aliases, error=error)
        self.completed += 1
        return result

    def _finish(self, pending) -> Optional[Predictions]:
        if not getattr(pending, "counted", False):
            self.completed += 1
        return pending.wait()

    def step(self, local_batch: Optional[Batch], total_utterances: Optional[int] = None) -> Optional[Predictions]:
        """``total_utterances``: size of THIS step's global batch (default: the constructor's), used by the padded gather."""
        total = self._total if total_utterances is None else int(total_utterances)
        # (a range report of an earlier pass -- FloatingPointError -- must not keep this rank out of the step's collectives)
        local, error = predict_reporting(self._predict, local_batch) if local_batch is not None else (None, None)
        if not self._flat:
            return self._padded(local, total, error)
        if self._verify and not self._same_geometry(local):
            # the flat gather cannot take this step: complete what is in flight first (order on rank `dst`), then gather
            # padded.  The in-flight gather is the PREVIOUS step's result and is what this call returns; this step's own
            # result is complete as well and is parked for the next step() / drain().
            had_pending = self._pending is not None
            previous = self.drain()
            result = self._padded(local, total, error)
            if not had_pending:
                return result
            self._pending = _CompletedGather(result)
            return previous
        handle = gather_flat_predictions(local, self._device, dst=self._dst, group=self._group, async_op=True, error=error)
        if not self._overlap:
            self.completed += 1
            return handle.wait()
        previous, self._pending = self._pending, handle
        if previous is None:
            return None
        return self._finish(previous)

    def drain(self) -> Optional[Predictions]:
        if self._pending is None:
            return None
        pending, self._pending = self._pending, None
        return self._finish(pending)


def gather_decoded(local: Optional[Decoded], names: List[str], total_utterances: int, device: torch.device, dst: int = 0,
                   group=None, capacity: Optional[int] = None) -> Optional[Dict[str, List[List[CTCHypothesis]]]]:
    """Gathers greedy CTC alignments instead of log-probabilities (only token ids cross xGMI).
    ``local`` is the shard's ``Estimator.greedy_decode_device`` result (``None`` for an empty shard), ``names`` the outputs
    to move (e.g. ``["phoneme"]``; run.py:767-774 decodes the phoneme output and, on request, attribute outputs).  One
    ``all_reduce(MAX)`` agrees on the longest alignment K, then ONE ``gather`` moves a packed int32 block per rank:
    ``[O, n_max]`` counts, ``[O, n_max]`` score bits, ``[O, n_max, K]`` tokens and ``[O, n_max, K]`` timesteps -- for config 3
    and the phoneme output at most 4 x 2 x 499 x 4 B = 16 KB per rank against 1.4 MB of log-probabilities.  With ``capacity``
    (an alignment length no rank can exceed, e.g. the frame count of the global batch's padded length) the agreement -- an
    ``all_reduce`` whose result the host reads back, i.e. a synchronisation with the step just enqueued -- is skipped and
    every rank sends ``capacity`` slots per utterance: the call then only enqueues work.  Returns, on ``dst``,
    the hypotheses of the whole batch in the reference's form (per output, per utterance
    ``[CTCHypothesis(tokens, [], score, timesteps)]``); ``None`` elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    bounds = shard_bounds(total_utterances, world)
    n_max = max(hi - lo for lo, hi in bounds)
    n_out = len(names)
    if local is not None:
        local = local.select(names)
        if local.counts.shape[1] != bounds[rank][1] - bounds[rank][0]:
            raise ValueError("the local alignments do not cover this rank's block of utterances")
    if capacity is not None:
        k_max = int(capacity)
        # an alignment has at most one entry per frame: a token tensor no wider than the capacity cannot overflow it (a
        # static shape -- no read-back); a wider one is checked on `dst` after the gather
        needs_check = local is not None and local.tokens.shape[2] > k_max
    else:
        needs_check = False
        k_tensor = torch.zeros(1, dtype=torch.int32, device=device)
        if local is not None and local.counts.numel():
            k_tensor = local.counts.max().to(device=device, dtype=torch.int32).reshape(1)
        dist.all_reduce(k_tensor, op=dist.ReduceOp.MAX, group=group)
        k_max = int(k_tensor.item())

    head = n_out * n_max
    packed = torch.zeros(2 * head + 2 * head * k_max, dtype=torch.int32, device=device)
    if local is not None:
        n_local = local.counts.shape[1]
        k_local = min(k_max, local.tokens.shape[2])
        packed[:head].view(n_out, n_max)[:, :n_local] = local.counts.to(device)
        packed[head: 2 * head].view(n_out, n_max)[:, :n_local] = local.scores.to(device).contiguous().view(torch.int32)
        body = packed[2 * head:].view(2, n_out, n_max, k_max)
        body[0, :, :n_local, :k_local] = local.tokens[:, :, :k_local].to(device=device, dtype=torch.int32)
        body[1, :, :n_local, :k_local] = local.timesteps[:, :, :k_local].to(device=device, dtype=torch.int32)
    gathered = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
    dist.gather(packed, gathered, dst=dst, group=group)
    if needs_check and int(local.counts.max()) > k_max:
        raise ValueError(f"an alignment of this rank holds more than capacity={k_max} tokens: it would be truncated")
    if rank != dst:
        return None
    result: Dict[str, List[List[CTCHypothesis]]] = {name: [] for name in names}
    for r, (lo, hi) in enumerate(bounds):
        if hi <= lo:
            continue
        block = gathered[r].cpu()
        counts = block[:head].view(n_out, n_max)
        if counts.numel() and int(counts.max()) > k_max:  # the block is on the host already: no extra synchronisation
            raise ValueError(f"rank {r} sent an alignment of {int(counts.max())} tokens, more than capacity={k_max}")
        scores = block[head: 2 * head].view(torch.float32).view(n_out, n_max)
        body = block[2 * head:].view(2, n_out, n_max, k_max).to(torch.int64)
        for o, name in enumerate(names):
            for n in range(hi - lo):
                k = int(counts[o, n])
                result[name].append([CTCHypothesis(body[0, o, n, :k].clone(), [], float(scores[o, n]), body[1, o, n, :k].clone())])
    return result


def unique_outputs(predictions: Predictions) -> Tuple[List[Tuple[str, int]], Dict[str, str]]:
    """Splits ``predictions.outputs`` into the distinct tensors (name, classes) and the aliases that share storage with
    one of them (``"phone"`` -> ``"phoneme"`` for allophone models, acoustic_model.py:161-167)."""
    seen: Dict[int, str] = {}
    unique: List[Tuple[str, int]] = []
    aliases: Dict[str, str] = {}
    items = list(predictions.outputs.items())
    # the later name wins as the canonical one so that the alias precedes it like upstream ("phone" before "phoneme")
    for name, tensor in reversed(items):
        key = tensor.data_ptr()
        if key in seen:
            aliases[name] = seen[key]
        else:
            seen[key] = name
            unique.append((name, tensor.shape[-1]))
    unique.reverse()
    return unique, aliases
