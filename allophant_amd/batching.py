"""Host-side batching for the prediction path.

Counterparts of the pieces of the reference's data path that decide what one ``Estimator.predict`` call sees:

* ``max_frame_batches``  -- ``MaxFrameBatchSampler.__iter__`` (reference allophant/batching.py:94-139): greedy packing in
  sampler order; a batch is closed as soon as ``(utterances + 1) * longest`` would exceed the frame budget;
* ``utterance_batches``  -- the ``"utterances"`` batching mode (``torch.utils.data.BatchSampler`` with ``drop_last=False``,
  batching.py:267-300);
* ``collate``            -- ``_build_batch`` for unlabeled batches (batching.py:162-177): zero right-padding to the longest
  utterance (``rnn.pad_sequence(..., True)``), int64 lengths and language ids -- exactly the padding contract
  ``zero_mean_unit_var_norm`` relies on (acoustic_model.py:762-767);
* ``split_by_language``  -- ``RawLabeledBatch.split_by_language`` (dataset_processing.py:103-126): runs of equal language
  id become separate batches re-padded to their own longest utterance; ``run.predict`` does this for composition models
  because one ``composition_feature_matrix`` applies to a whole batch (run.py:558-568, 710-719).

New functionality without an upstream counterpart (the reference feeds the GPU from DataLoader workers and a blocking
``batch.to(device, non_blocking=True)`` of pageable memory, run.py:742-743):

* ``length_sorted_order`` -- sampler order that sorts utterances by length (optionally inside language runs), which cuts the
  padding a max-frames batch carries;
* ``Prefetcher``          -- stages the next batch in pinned host memory and copies it to the device on a side HIP stream
  while the current batch is being computed.
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional, Sequence, Tuple

import torch
from torch import Tensor

from .estimator import Batch


def max_frame_batches(order: Iterable[int], frame_lengths: Sequence[int], max_frames: int) -> Iterator[List[int]]:
    """Index lists whose padded size ``len(batch) * max(length)`` stays within ``max_frames``.  Bit-exact with upstream,
    including its corner case: an utterance longer than the whole budget forms a batch of its own and, when it is the first
    of the stream, is preceded by an empty batch (callers skip empty lists)."""
    batch: List[int] = []
    longest = 0
    for index in order:
        length = int(frame_lengths[index])
        longest = max(longest, length)
        if (len(batch) + 1) * longest > max_frames:
            yield batch
            batch = [index]
            longest = length
        else:
            batch.append(index)
    if batch:
        yield batch


def bucketed_frame_batches(order: Iterable[int], frame_lengths: Sequence[int], max_frames: int,
                           bucket: int) -> Iterator[Tuple[List[int], int]]:
    """``max_frame_batches`` on a grid of batch GEOMETRIES (no upstream counterpart): yields ``(indices, padded_length)`` where the
    padded length is the batch's longest utterance rounded up to a multiple of ``bucket`` samples and a batch holds at most
    ``max_frames // padded_length`` utterances (every batch of a bucket but its last: exactly that many).  Batches of one bucket
    therefore share one ``(N, L)``, which is what a recording of a forward pass is keyed on (``amx_forward``, ABI 6): in
    length-sorted order the corpus becomes runs of equal geometry and the passes of a run replay one HIP graph.  The cost is the
    padding up to the bucket boundary -- ``collate(..., padded_length=...)`` marks such a batch as padded beyond its longest
    utterance (``AMX_FLAG_PADDED``), which the XLS-R form's results do not depend on (``parallel.padding_sensitive``)."""
    if bucket < 1:
        raise ValueError("bucket must be a positive number of samples")
    batch: List[int] = []
    longest = 0

    def padded(length: int) -> int:
        return -(-length // bucket) * bucket

    for index in order:
        length = int(frame_lengths[index])
        grown = max(longest, length)
        if batch and (len(batch) + 1) * padded(grown) > max_frames:
            yield batch, padded(longest)
            batch, longest = [index], length
        else:
            batch.append(index)
            longest = grown
    if batch:
        yield batch, padded(longest)


def utterance_batches(order: Iterable[int], batch_size: int) -> Iterator[List[int]]:
    batch: List[int] = []
    for index in order:
        batch.append(index)
        if len(batch) == batch_size:
            yield batch
            batch = []
    if batch:
        yield batch


def collate(audio: Sequence[Tensor], language_ids: Optional[Sequence[int]] = None, pin: bool = False) -> Batch:
    """Dense zero-right-padded batch ``[N, max(len)]`` from 1-D waveforms."""
    lengths = torch.tensor([int(a.numel()) for a in audio], dtype=torch.int64)
    longest = int(lengths.max()) if len(audio) else 0
    features = torch.zeros(len(audio), longest, dtype=torch.float32)
    if pin:
        features = features.pin_memory()
    for i, a in enumerate(audio):
        features[i, : a.numel()] = a.reshape(-1)
    ids = torch.tensor(list(language_ids) if language_ids is not None else [0] * len(audio), dtype=torch.int64)
    return Batch(features, lengths, ids)


def split_by_language(batch: Batch) -> Iterator[Tuple[int, Batch]]:
    """``(language_id, sub-batch)`` for every run of consecutive equal language ids."""
    ids = batch.language_ids
    n = len(batch)
    start = 0
    while start < n:
        stop = start + 1
        while stop < n and int(ids[stop]) == int(ids[start]):
            stop += 1
        lengths = batch.lengths[start:stop]
        yield int(ids[start]), Batch(batch.audio_features[start:stop, : int(lengths.max())], lengths, ids[start:stop])
        start = stop


def length_sorted_order(frame_lengths: Sequence[int], language_ids: Optional[Sequence[int]] = None,
                        descending: bool = True) -> List[int]:
    """Indices sorted by length (inside each language when ids are given, languages in first-appearance order) -- ties keep
    corpus order."""
    idx = list(range(len(frame_lengths)))
    sign = -1 if descending else 1
    if language_ids is None:
        return sorted(idx, key=lambda i: (sign * int(frame_lengths[i]), i))
    first_seen = {}
    for i in idx:
        first_seen.setdefault(int(language_ids[i]), len(first_seen))
    return sorted(idx, key=lambda i: (first_seen[int(language_ids[i])], sign * int(frame_lengths[i]), i))


def padding_efficiency(batches: Iterable[Sequence[int]], frame_lengths: Sequence[int]) -> float:
    """valid samples / padded samples over a batch sequence."""
    valid = padded = 0
    for b in batches:
        ls = [int(frame_lengths[i]) for i in b]
        valid += sum(ls)
        padded += len(ls) * max(ls)
    return valid / padded if padded else 1.0


class PinnedCollator:
    """``collate`` into a small ring of reusable pinned host buffers (``hipHostMalloc`` per batch costs milliseconds).  A
    slot is refilled ``depth`` batches later; the asynchronous host-to-device copy that read it may not have run yet (the
    host runs ahead of the GPU), so whoever issues that copy hands the slot an event recorded behind it
    (``mark_in_flight``; ``Prefetcher`` does) and the refill waits for that event first."""

    def __init__(self, max_samples: int, depth: int = 3):
        self._buffers = [torch.empty(max_samples, dtype=torch.float32).pin_memory() for _ in range(depth)]
        self._events: List[Optional["torch.cuda.Event"]] = [None] * depth
        self._handed_out = [False] * depth  # slot given to a caller since its last refill
        self._next = 0

    def mark_in_flight(self, batch: Batch, event: "torch.cuda.Event") -> None:
        """``event`` completes once the device copy of ``batch`` (a batch this collator returned) has read its slot.
        ``Prefetcher`` and ``Batch.to(device, non_blocking=True)`` call this."""
        slot = getattr(batch, "_pinned_slot", None)
        if slot is not None and slot[0] is self:
            self._events[slot[1]] = event

    def __call__(self, audio: Sequence[Tensor], language_ids: Optional[Sequence[int]] = None,
                 padded_length: Optional[int] = None) -> Batch:
        """``padded_length`` (``bucketed_frame_batches``): pad to that many samples instead of the longest utterance; the batch
        is then marked as padded beyond ``max(lengths)`` (``Estimator.predict`` passes ``AMX_FLAG_PADDED``)."""
        lengths = torch.tensor([int(a.numel()) for a in audio], dtype=torch.int64)
        longest = int(lengths.max())
        over_padded = padded_length is not None and int(padded_length) > longest
        if padded_length is not None:
            if int(padded_length) < longest:
                raise ValueError("padded_length is shorter than the longest utterance of the batch")
            longest = int(padded_length)
        n = len(audio)
        slot = self._next
        buf = self._buffers[slot]
        if n * longest > buf.numel():
            if over_padded:
                raise ValueError("the pinned ring is smaller than this padded batch")
            return collate(audio, language_ids, pin=True)  # over-long single utterance: one-off buffer, the ring stays put
        self._next = (self._next + 1) % len(self._buffers)
        if self._events[slot] is not None:
            self._events[slot].synchronize()  # the copy out of this slot has finished
            self._events[slot] = None
        elif self._handed_out[slot] and torch.cuda.is_available() and torch.cuda.is_initialized():
            # the slot went out and nobody reported a copy event (a hand-rolled asynchronous copy): wait for everything
            # rather than refill memory a host-to-device copy may still be reading
            torch.cuda.synchronize()
        features = buf[: n * longest].view(n, longest)
        for i, a in enumerate(audio):
            k = a.numel()
            features[i, :k].copy_(a.reshape(-1))
            if k < longest:
                features[i, k:].zero_()
        ids = torch.tensor(list(language_ids) if language_ids is not None else [0] * n, dtype=torch.int64)
        batch = Batch(features, lengths, ids)
        batch._pinned_slot = (self, slot)
        if over_padded:
            batch._padded = True
        self._handed_out[slot] = True
        return batch


class Prefetcher:
    """Iterates device-resident batches.  Each ``next()`` collates one batch into pinned memory and copies it to the device
    on a side stream; because ``Estimator.predict`` only *launches* work, the caller's call for batch k+1 arrives while the
    GPU is still computing batch k, so the host-side collation and the host-to-device copy of k+1 overlap the compute of k
    (the compute stream waits for the copy only when it actually consumes the batch).  ``batches`` yields CPU ``Batch``
    objects, or anything ``fetch`` turns into one (e.g. index lists).  Batches that come out of a ``PinnedCollator`` carry
    their slot with them; the collator is told the event behind which the copy out of that slot has completed."""

    def __init__(self, batches: Iterable, device: torch.device, fetch=None, ring: int = 0):
        """``ring`` > 0: device audio buffers are REUSED -- ``ring`` of them per batch shape, handed out in turn -- instead of coming
        fresh from the caching allocator: a recording of a forward pass holds the addresses of its buffers (``amx_forward``), so a
        loop that wants its passes replayed has to present the same ones again.  The copy into a slot waits (on the copy stream)
        for everything the compute stream had been given when the slot's previous batch had been consumed, i.e. one ``next()``
        later -- the loop ``for batch in prefetcher: predict(batch)`` is the contract (``ring`` >= 2)."""
        self._source = iter(batches)
        self._device = torch.device(device)
        self._fetch = fetch
        self._stream = torch.cuda.Stream(self._device)
        self._ring = int(ring)
        self._slots = {}       # shape -> [device tensors]
        self._turn = {}        # shape -> next slot
        self._history = []     # (shape, slot) of the batches handed out, newest last
        self._events = {}      # (shape, slot) -> event on the compute stream behind the forward pass that read the slot

    @property
    def copy_stream(self) -> "torch.cuda.Stream":
        return self._stream

    def __iter__(self):
        return self

    def __next__(self) -> Batch:
        item = next(self._source)
        batch = self._fetch(item) if self._fetch is not None else item
        audio = batch.audio_features if batch.audio_features.is_pinned() else batch.audio_features.pin_memory()
        ring_key = None
        if self._ring > 0:
            current = torch.cuda.current_stream(self._device)
            if self._history:
                # everything enqueued so far includes the forward pass of the batch handed out last: its slot is free behind this
                event = torch.cuda.Event()
                event.record(current)
                self._events[self._history[-1]] = event
            shape = tuple(audio.shape)
            slots = self._slots.setdefault(shape, [])
            turn = self._turn.get(shape, 0)
            if len(slots) < self._ring:
                slots.append(torch.empty(shape, dtype=audio.dtype, device=self._device))
                turn = len(slots) - 1
            self._turn[shape] = (turn + 1) % self._ring
            ring_key = (shape, turn)
            if ring_key in self._events:
                self._stream.wait_event(self._events.pop(ring_key))
            elif ring_key in self._history[-1:]:
                self._stream.wait_stream(current)  # (ring of one: the batch just handed out still owns the slot)
        with torch.cuda.stream(self._stream):
            if ring_key is not None:
                dev = self._slots[ring_key[0]][ring_key[1]]
                dev.copy_(audio, non_blocking=True)
                self._history = (self._history + [ring_key])[-2:]
            else:
                dev = audio.to(self._device, non_blocking=True)
            slot = getattr(batch, "_pinned_slot", None)
            if slot is not None:
                event = torch.cuda.Event()
                event.record(self._stream)
                slot[0].mark_in_flight(batch, event)
        current = torch.cuda.current_stream(self._device)
        current.wait_stream(self._stream)  # stream-ordered: later launches on the compute stream see the copied batch
        if ring_key is None:
            dev.record_stream(current)
        moved = Batch(dev, batch.lengths, batch.language_ids)
        if getattr(batch, "_padded", False):
            moved._padded = True
        return moved


class Batcher:
    """``Batcher`` with the reference's constructor and ``batches()`` signature (allophant/batching.py:229-342) for the
    prediction path: ``batching_mode`` ``"utterances"`` (at most ``batch_size`` utterances per batch) or ``"frames"`` (at
    most ``batch_size`` padded samples per batch, needs ``data_lengths``), sequential or seeded-shuffled sampler order,
    ``skip_batches``.  ``data`` is any indexable of 1-D waveforms (or of ``(waveform, language_id)`` pairs); batches are
    collated like upstream's ``_build_batch`` (zero right-padding).  Language oversampling is a training-side feature
    (batching.py:32-91) and not part of this path; ``data_workers`` is accepted for signature compatibility -- collation
    is a memcpy here, decoding audio files is the caller's job."""

    def __init__(self, batch_size: int, batching_mode: str = "utterances", language_oversampling_factor: Optional[float] = None,
                 data_workers: Optional[int] = 0, collate_fn=None):
        mode = getattr(batching_mode, "value", batching_mode)
        if mode not in ("utterances", "frames"):
            raise ValueError(f"unknown batching mode {batching_mode!r}")
        if language_oversampling_factor is not None:
            raise NotImplementedError("language oversampling is a training-time sampler (batching.py:32-91), out of scope")
        self._batch_size = int(batch_size)
        self._mode = mode
        self._collate = collate_fn or collate

    @property
    def batch_size(self) -> int:
        return self._batch_size

    def index_batches(self, n: int, data_lengths: Optional[Sequence[int]] = None, shuffle: bool = False,
                      seed: Optional[int] = None, skip_batches: int = 0, order: Optional[Sequence[int]] = None) -> Iterator[List[int]]:
        """The index lists behind ``batches`` (``order`` overrides the sampler, e.g. ``length_sorted_order``)."""
        if order is None:
            if shuffle:
                generator = torch.Generator()
                if seed is not None:
                    generator.manual_seed(seed)
                order = torch.randperm(n, generator=generator).tolist()  # RandomSampler's permutation (batching.py:319)
            else:
                order = range(n)
        if self._mode == "utterances":
            it = utterance_batches(order, self._batch_size)
        else:
            if data_lengths is None:
                raise ValueError("Frame Lengths for each utterance are required for using max frame batching")
            it = max_frame_batches(order, data_lengths, self._batch_size)
        for i, indices in enumerate(it):  # SkipBatchSampler (batching.py:142-153)
            if i >= skip_batches:
                yield indices

    def batches(self, data, data_lengths: Optional[Sequence[int]] = None, shuffle: bool = False, seed: Optional[int] = None,
                skip_batches: int = 0, order: Optional[Sequence[int]] = None) -> Iterator[Batch]:
        for indices in self.index_batches(len(data), data_lengths, shuffle, seed, skip_batches, order):
            if not indices:
                continue  # upstream's empty first batch for an over-long first utterance collates to nothing useful
            items = [data[i] for i in indices]
            if items and isinstance(items[0], (tuple, list)):
                yield self._collate([a for a, _ in items], [int(l) for _, l in items])
            else:
                yield self._collate(items)
