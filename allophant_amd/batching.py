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
    buffer is reused ``depth`` batches later, i.e. after its host-to-device copy has long been consumed."""

    def __init__(self, max_samples: int, depth: int = 3):
        self._buffers = [torch.empty(max_samples, dtype=torch.float32).pin_memory() for _ in range(depth)]
        self._next = 0

    def __call__(self, audio: Sequence[Tensor], language_ids: Optional[Sequence[int]] = None) -> Batch:
        lengths = torch.tensor([int(a.numel()) for a in audio], dtype=torch.int64)
        longest = int(lengths.max())
        n = len(audio)
        buf = self._buffers[self._next]
        self._next = (self._next + 1) % len(self._buffers)
        if n * longest > buf.numel():
            return collate(audio, language_ids, pin=True)  # over-long single utterance: one-off buffer
        features = buf[: n * longest].view(n, longest)
        for i, a in enumerate(audio):
            k = a.numel()
            features[i, :k].copy_(a.reshape(-1))
            if k < longest:
                features[i, k:].zero_()
        ids = torch.tensor(list(language_ids) if language_ids is not None else [0] * n, dtype=torch.int64)
        return Batch(features, lengths, ids)


class Prefetcher:
    """Iterates device-resident batches.  Each ``next()`` collates one batch into pinned memory and copies it to the device
    on a side stream; because ``Estimator.predict`` only *launches* work, the caller's call for batch k+1 arrives while the
    GPU is still computing batch k, so the host-side collation and the host-to-device copy of k+1 overlap the compute of k
    (the compute stream waits for the copy only when it actually consumes the batch).  ``batches`` yields CPU ``Batch``
    objects, or anything ``fetch`` turns into one (e.g. index lists)."""

    def __init__(self, batches: Iterable, device: torch.device, fetch=None):
        self._source = iter(batches)
        self._device = torch.device(device)
        self._fetch = fetch
        self._stream = torch.cuda.Stream(self._device)

    def __iter__(self):
        return self

    def __next__(self) -> Batch:
        item = next(self._source)
        batch = self._fetch(item) if self._fetch is not None else item
        audio = batch.audio_features if batch.audio_features.is_pinned() else batch.audio_features.pin_memory()
        with torch.cuda.stream(self._stream):
            dev = audio.to(self._device, non_blocking=True)
        current = torch.cuda.current_stream(self._device)
        current.wait_stream(self._stream)  # stream-ordered: later launches on the compute stream see the copied batch
        dev.record_stream(current)
        return Batch(dev, batch.lengths, batch.language_ids)
