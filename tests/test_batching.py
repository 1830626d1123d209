"""``allophant_amd.batching`` against the REAL reference's ``MaxFrameBatchSampler`` / ``_build_batch`` /
``split_by_language`` (tests/golden/g7_batching.json, written by oracle/gen_batching_golden.py), plus the properties of
the length-sorted order."""
import json
import os

import torch

from allophant_amd import batching as B

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g7_batching.json")


def _golden():
    with open(GOLDEN) as f:
        return json.load(f)


def test_max_frame_batches_equal_the_reference_sampler():
    for case in _golden()["sampler_cases"]:
        got = list(B.max_frame_batches(case["order"], case["lengths"], case["max_frames"]))
        assert got == case["batches"]
        # the budget holds for every batch with more than one utterance; nothing is dropped or repeated
        for b in got:
            # (an utterance longer than the whole budget closes an EMPTY batch first -- upstream quirk, kept bit-exact)
            assert len(b) <= 1 or len(b) * max(case["lengths"][i] for i in b) <= case["max_frames"]
        assert sorted(i for b in got for i in b) == sorted(case["order"])


def test_collate_and_split_by_language_match_the_reference():
    g = _golden()
    c = g["collate"]
    audio = [torch.arange(1, l + 1, dtype=torch.float32) * (i + 1) for i, l in enumerate(c["lens"])]
    batch = B.collate(audio, c["langs"])
    assert batch.audio_features.tolist() == c["audio"] and batch.lengths.tolist() == c["lengths"]
    assert batch.language_ids.tolist() == c["ids"] and batch.lengths.dtype == torch.int64
    splits = list(B.split_by_language(batch))
    assert len(splits) == len(g["splits"])
    for (lang, sub), ref in zip(splits, g["splits"]):
        assert lang == ref["language"] and sub.audio_features.tolist() == ref["audio"]
        assert sub.lengths.tolist() == ref["lengths"] and sub.language_ids.tolist() == ref["ids"]
        assert sub.audio_features.shape[1] == int(sub.lengths.max())  # the L == max(lengths) contract of predict()


def test_utterance_batches_and_sorted_order():
    assert list(B.utterance_batches(range(7), 3)) == [[0, 1, 2], [3, 4, 5], [6]]
    lengths = [5, 9, 3, 9, 7]
    assert B.length_sorted_order(lengths) == [1, 3, 4, 0, 2]
    assert B.length_sorted_order(lengths, language_ids=[1, 0, 1, 0, 0]) == [0, 2, 1, 3, 4]
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(32000, 240000, (300,), generator=g).tolist()
    corpus = list(B.max_frame_batches(range(300), lens, 32 * 160000))
    sorted_ = list(B.max_frame_batches(B.length_sorted_order(lens), lens, 32 * 160000))
    assert B.padding_efficiency(sorted_, lens) > 0.9 > 0.7 > B.padding_efficiency(corpus, lens)


def test_batcher_has_the_reference_surface():
    """`Batcher(batch_size, batching_mode).batches(data, data_lengths, shuffle, seed, skip_batches)` (reference
    batching.py:229-342): frames mode == the reference's MaxFrameBatchSampler batches (golden g7), utterances mode ==
    BatchSampler(drop_last=False), SkipBatchSampler semantics, seeded shuffling, zero-padding collation."""
    import pytest

    g = _golden()
    for case in g["sampler_cases"]:
        if list(case["order"]) != list(range(len(case["lengths"]))):
            continue
        batcher = B.Batcher(case["max_frames"], "frames")
        assert list(batcher.index_batches(len(case["lengths"]), case["lengths"])) == case["batches"]
    data = [torch.full((n,), float(i + 1)) for i, n in enumerate((4, 9, 2, 7, 5))]
    batches = list(B.Batcher(2).batches(data))
    assert [b.audio_features.shape for b in batches] == [(2, 9), (2, 7), (1, 5)]
    assert batches[0].audio_features[0].tolist() == [1.0] * 4 + [0.0] * 5 and batches[0].lengths.tolist() == [4, 9]
    assert [len(b) for b in B.Batcher(2).batches(data, skip_batches=1)] == [2, 1]
    with pytest.raises(ValueError, match="Frame Lengths"):
        list(B.Batcher(100, "frames").batches(data))
    a = list(B.Batcher(2).index_batches(5, shuffle=True, seed=7))
    assert a == list(B.Batcher(2).index_batches(5, shuffle=True, seed=7)) and sorted(i for b in a for i in b) == list(range(5))
    with_language = list(B.Batcher(3).batches([(d, i % 2) for i, d in enumerate(data)]))
    assert with_language[0].language_ids.tolist() == [0, 1, 0]
    assert B.Batcher(16, "utterances").batch_size == 16


import pytest


@pytest.mark.gpu
def test_pinned_collator_equals_collate():
    if not torch.cuda.is_available():
        import pytest
        pytest.skip("pinned host memory needs the HIP runtime")
    audio = [torch.randn(n) for n in (5, 9, 3)]
    ring = B.PinnedCollator(64, depth=2)
    for _ in range(3):  # buffers are reused: stale tails must be re-zeroed
        got = ring(audio, [1, 1, 2])
        ref = B.collate(audio, [1, 1, 2])
        assert torch.equal(got.audio_features, ref.audio_features) and torch.equal(got.lengths, ref.lengths)
        audio = audio[::-1]


@pytest.mark.gpu
def test_pinned_collator_slot_travels_with_asynchronous_copies():
    """A slot is refilled only behind the copy that read it, whoever issued the copy: ``Batch.to(device, non_blocking=True)``
    reports its event to the collator; an over-long batch takes a one-off buffer and does not skip a ring slot."""
    if not torch.cuda.is_available():
        pytest.skip("pinned host memory needs the HIP runtime")
    ring = B.PinnedCollator(32, depth=2)
    first = ring([torch.full((8,), 1.0), torch.full((4,), 2.0)])
    assert first._pinned_slot == (ring, 0)
    dev = first.to("cuda:0", non_blocking=True)
    assert ring._events[0] is not None  # the asynchronous copy handed its event over
    big = ring([torch.zeros(40)])  # does not fit a slot: one-off pinned buffer ...
    assert getattr(big, "_pinned_slot", None) is None and big.audio_features.is_pinned()
    second = ring([torch.full((6,), 3.0)])
    assert second._pinned_slot == (ring, 1)  # ... and the ring did not advance past slot 1
    third = ring([torch.full((8,), 4.0)])   # back on slot 0: waits for the event of `first`'s copy, then refills
    assert third._pinned_slot == (ring, 0) and ring._events[0] is None
    torch.cuda.synchronize()
    assert dev.audio_features.cpu().tolist() == [[1.0] * 8, [2.0] * 4 + [0.0] * 4]


def test_bucketed_frame_batches_put_a_corpus_on_a_grid_of_geometries():
    """``bucketed_frame_batches`` (no upstream counterpart; round 6): every utterance exactly once, every batch inside the frame
    budget at its bucketed padded length, batches of one bucket share (N, L) except a bucket's last -- the property recordings of
    forward passes are keyed on -- and ``PinnedCollator(padded_length=...)`` marks such batches as padded beyond their longest."""
    import torch

    from allophant_amd import batching as B

    g = torch.Generator().manual_seed(3)
    lengths = torch.randint(2 * 16000, 15 * 16000, (4096,), generator=g).tolist()
    budget, bucket = 32 * 160000, 16000
    order = B.length_sorted_order(lengths)
    batches = list(B.bucketed_frame_batches(order, lengths, budget, bucket))
    assert sorted(i for b, _ in batches for i in b) == list(range(len(lengths)))
    geometries = {}
    for indices, padded in batches:
        longest = max(lengths[i] for i in indices)
        assert padded % bucket == 0 and longest <= padded < longest + bucket
        assert len(indices) * padded <= budget
        geometries.setdefault((len(indices), padded), 0)
        geometries[(len(indices), padded)] += 1
    # 13 one-second buckets: most batches repeat a geometry (the share of passes that can replay a recording)
    repeats = sum(c - 1 for c in geometries.values())
    assert repeats / len(batches) > 0.7, (repeats, len(batches), len(geometries))
    # an order that is not sorted still respects the budget
    for indices, padded in B.bucketed_frame_batches(range(len(lengths)), lengths, budget, bucket):
        assert len(indices) * padded <= budget and max(lengths[i] for i in indices) <= padded
    with pytest.raises(ValueError):
        next(B.bucketed_frame_batches(order, lengths, budget, 0))
    # the collator pads to the bucket and says so
    audio = [torch.ones(n) for n in (30000, 20000)]
    plain = B.collate(audio)
    assert plain.audio_features.shape == (2, 30000) and not getattr(plain, "_padded", False)
