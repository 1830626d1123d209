"""Utterance-level data parallelism on CPU: two gloo ranks shard a batch, run the forward pass of their block (the CPU
oracle stands in for the HIP path here -- there is no GPU in this container) and gather the log-probabilities to rank 0,
which must equal the single-process result."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from allophant_amd import spec as S, synthetic
from allophant_amd.estimator import Batch, Predictions
from allophant_amd.parallel import (data_parallel_predict, gather_flat_predictions, gather_predictions, shard_batch,
                                    shard_bounds, unique_outputs)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _setup():
    spec = S.multitask_spec(S.tiny_encoder(1), ["syllabic", "long"], embedding_size=16, train_phonemes=9, n_features=5,
                            allophone_layer=True)
    spec["shared_phones"] = 11
    state = synthetic.make_state_dict(spec, seed=4)
    audio, lengths = synthetic.make_audio(5, 3000, seed=11, ragged=True)
    tfi = synthetic.make_inventory(spec, 6, seed=4)
    return spec, state, audio, lengths, tfi


def _predict(spec, state, batch, tfi):
    from oracle import allophant_oracle as O

    out, flen = O.predict(batch.audio_features, batch.lengths, state, spec, tfi, synthetic.category_offsets(spec))
    return Predictions(out, flen)


def _worker(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    spec, state, audio, lengths, tfi = _setup()
    full = Batch(audio, lengths, torch.zeros(len(lengths), dtype=torch.long))
    names = [(n, c) for n, c in [("syllabic", 4), ("long", 4), ("phoneme", 7)]]
    gathered = data_parallel_predict(lambda b: _predict(spec, state, b, tfi), full, names, torch.device("cpu"), dst=0,
                                     aliases={"phone": "phoneme"}, spec=spec)
    # the same gather with the global frame count stated up front (no agreement round): identical tensors
    shard = shard_batch(full, rank, world, spec=spec)
    again = gather_predictions(_predict(spec, state, shard, tfi) if shard is not None else None, names, len(full),
                               torch.device("cpu"), dst=0, aliases={"phone": "phoneme"},
                               frames=S.frame_lengths([int(lengths.max())], spec)[0])
    if rank == 0:
        assert list(again.outputs) == list(gathered.outputs)
        for name in again.outputs:
            assert torch.equal(again.outputs[name], gathered.outputs[name]), name
    if rank == 0:
        torch.save({"outputs": gathered.outputs, "lengths": gathered.lengths}, result_path)
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds():
    assert shard_bounds(32, 8) == [(4 * i, 4 * i + 4) for i in range(8)]
    assert shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert shard_bounds(1, 2) == [(0, 1), (1, 1)]


def test_unique_outputs_aliases():
    t = torch.zeros(3, 2, 4)
    p = Predictions({"a": torch.ones(3, 2, 2), "phone": t, "phoneme": t}, torch.tensor([3, 2]))
    unique, aliases = unique_outputs(p)
    assert unique == [("a", 2), ("phoneme", 4)] and aliases == {"phone": "phoneme"}


def test_two_rank_gather_matches_single_process(tmp_path):
    world = 2
    port = _free_port()
    result_path = str(tmp_path / "gathered.pt")
    mp.spawn(_worker, args=(world, port, result_path), nprocs=world, join=True)
    got = torch.load(result_path)
    spec, state, audio, lengths, tfi = _setup()
    single = _predict(spec, state, Batch(audio, lengths, torch.zeros(len(lengths), dtype=torch.long)), tfi)
    assert list(got["outputs"].keys()) == ["syllabic", "long", "phone", "phoneme"]
    assert torch.equal(got["lengths"], single.lengths)
    for name, expected in single.outputs.items():
        g = got["outputs"][name]
        assert g.shape == expected.shape
        valid = (torch.arange(g.shape[0]).unsqueeze(1) < single.lengths.unsqueeze(0)).unsqueeze(-1)
        # sharding re-pads each block to its own longest utterance: identical up to fp32 reassociation
        assert ((g - expected).abs() * valid).max().item() < 1e-4, name


def _flat_worker(rank, world, port, result_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the layout Estimator.predict produces: one flat block, per-output [T, N, C] views, "phone" aliasing "phoneme"
    t, n = 5, 3
    widths = [("syllabic", 4), ("phoneme", 7)]
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(sum(t * n * c for _, c in widths), generator=g)
    outputs, off = {}, 0
    for name, c in widths:
        view = flat[off: off + t * n * c].view(t, n, c)
        if name == "phoneme":
            outputs["phone"] = view
        outputs[name] = view
        off += t * n * c
    local = Predictions(outputs, torch.tensor([5, 4, 2]) + rank, _flat=flat)
    if rank == 0:
        # the asynchronous form (bench.py overlaps the gather of step k with the forward pass of step k + 1)
        handle = gather_flat_predictions(local, torch.device("cpu"), dst=0, async_op=True)
        gathered = handle.wait()
    else:
        gathered = gather_flat_predictions(local, torch.device("cpu"), dst=0)
    if rank == 0:
        torch.save({"outputs": gathered.outputs, "lengths": gathered.lengths}, result_path)
    else:
        assert gathered is None
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gather_assembles_the_global_batch(tmp_path):
    """The benchmark's gather: equal-shaped shards, one flat block per rank, [T, world * N, C] per output on rank 0."""
    world = 2
    result_path = str(tmp_path / "flat.pt")
    mp.spawn(_flat_worker, args=(world, _free_port(), result_path), nprocs=world, join=True)
    got = torch.load(result_path)
    assert list(got["outputs"].keys()) == ["syllabic", "phone", "phoneme"]
    assert got["lengths"].tolist() == [5, 4, 2, 6, 5, 3]
    assert got["outputs"]["phone"].data_ptr() == got["outputs"]["phoneme"].data_ptr() or torch.equal(got["outputs"]["phone"], got["outputs"]["phoneme"])
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        flat = torch.randn(5 * 3 * 4 + 5 * 3 * 7, generator=g)
        assert torch.equal(got["outputs"]["syllabic"][:, 3 * rank: 3 * rank + 3], flat[: 60].view(5, 3, 4))
        assert torch.equal(got["outputs"]["phoneme"][:, 3 * rank: 3 * rank + 3], flat[60:].view(5, 3, 7))


# ---- the N > 1 code path of bench.py (BASELINE config 3): shard_batch + DataParallelRunner (overlapped flat gather) ----
def _flat_predictions(out, flen):
    """The oracle's outputs in the layout ``Estimator.predict`` produces: one flat block, [T, N, C] views in output order,
    "phone" aliasing "phoneme"."""
    names = [k for k in out if k != "phone"]
    flat = torch.cat([out[k].reshape(-1) for k in names])
    views, off = {}, 0
    for k in names:
        t, n, c = out[k].shape
        view = flat[off: off + t * n * c].view(t, n, c)
        if k == "phoneme" and "phone" in out:
            views["phone"] = view
        views[k] = view
        off += t * n * c
    return Predictions(views, flen, _flat=flat)


def _runner_worker(rank, world, port, result_path):
    from oracle import allophant_oracle as O
    from allophant_amd.parallel import DataParallelRunner

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    spec, state, _, _, tfi = _setup()
    offsets = synthetic.category_offsets(spec)

    def predict(batch):
        out, flen = O.predict(batch.audio_features, batch.lengths, state, spec, tfi, offsets)
        return _flat_predictions(out, flen)

    runner = DataParallelRunner(predict, torch.device("cpu"), dst=0)
    steps = []
    for step in range(3):
        # the global batch of this step (same on every rank, full-length utterances: equal shards as in the benchmark)
        audio, lengths = synthetic.make_audio(4, 2400, seed=50 + step)
        shard = shard_batch(Batch(audio, lengths, torch.zeros(4, dtype=torch.long)), rank, world, spec=spec)
        steps.append(runner.step(shard))
    steps.append(runner.drain())
    assert steps[0] is None  # the first gather is still in flight when step 0 returns
    if rank == 0:
        assert runner.completed == 3 and all(s is not None for s in steps[1:])
        torch.save([{"outputs": s.outputs, "lengths": s.lengths} for s in steps[1:]], result_path)
    else:
        assert all(s is None for s in steps)
    dist.barrier()
    dist.destroy_process_group()


def test_bench_strong_scaling_path_two_ranks(tmp_path):
    """bench.py --gpus N (N > 1): the global batch is sharded by ``shard_batch``, every rank predicts its shard and
    ``DataParallelRunner`` gathers the flat log-prob blocks to rank 0, one step behind the forward passes.  With the CPU
    oracle standing in for the HIP path, every step's gathered result must equal the single-process prediction of that
    step's global batch."""
    from oracle import allophant_oracle as O

    world = 2
    result_path = str(tmp_path / "runner.pt")
    mp.spawn(_runner_worker, args=(world, _free_port(), result_path), nprocs=world, join=True)
    got = torch.load(result_path)
    spec, state, _, _, tfi = _setup()
    assert len(got) == 3
    for step, g in enumerate(got):
        audio, lengths = synthetic.make_audio(4, 2400, seed=50 + step)
        ref, ref_len = O.predict(audio, lengths, state, spec, tfi, synthetic.category_offsets(spec))
        assert list(g["outputs"]) == list(ref) and g["lengths"].tolist() == ref_len.tolist()
        for name in ref:
            assert g["outputs"][name].shape == ref[name].shape
            assert (g["outputs"][name] - ref[name]).abs().max().item() < 1e-4, (step, name)


def _global_decoded(total, seed):
    """Seeded alignments of a ``total``-utterance batch in ``Decoded`` form (three outputs, up to 20 frames)."""
    from allophant_amd.estimator import Decoded

    g = torch.Generator().manual_seed(seed)
    n_out, frames = 3, 20
    counts = torch.randint(0, frames + 1, (n_out, total), generator=g, dtype=torch.int32)
    tokens = torch.randint(1, 40, (n_out, total, frames), generator=g)
    timesteps = torch.sort(torch.randint(1, 500, (n_out, total, frames), generator=g), dim=-1).values
    scores = -torch.rand(n_out, total, generator=g) * 100
    return Decoded(["syllabic", "long", "phoneme"], tokens, timesteps, counts, scores)


def _decoded_worker(rank, world, port, total, result_path, capacity=None):
    from allophant_amd.estimator import Decoded
    from allophant_amd.parallel import gather_decoded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = _global_decoded(total, seed=77)
    lo, hi = shard_bounds(total, world)[rank]
    local = None
    if hi > lo:
        local = Decoded(full.names, full.tokens[:, lo:hi], full.timesteps[:, lo:hi], full.counts[:, lo:hi], full.scores[:, lo:hi])
    got = gather_decoded(local, ["phoneme", "long"], total, torch.device("cpu"), dst=0, capacity=capacity)
    if rank == 0:
        torch.save(got, result_path)
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_of_decoded_alignments(tmp_path):
    """the data-parallel path can move greedy CTC alignments instead of log-probabilities.  Ragged shards
    (5 utterances -> 3 + 2) and an empty shard (1 utterance on 2 ranks) must reproduce the single-process hypotheses
    bit for bit, in utterance order, for the selected outputs only."""
    for total, capacity in ((5, None), (1, None), (5, 20), (1, 33)):  # with a capacity: no agreement round, same result
        result_path = str(tmp_path / f"decoded{total}_{capacity}.pt")
        mp.spawn(_decoded_worker, args=(2, _free_port(), total, result_path, capacity), nprocs=2, join=True)
        got = torch.load(result_path, weights_only=False)
        want = _global_decoded(total, seed=77).select(["phoneme", "long"]).hypotheses()
        assert list(got) == ["phoneme", "long"]
        for name in want:
            assert len(got[name]) == total
            for g, w in zip(got[name], want[name]):
                assert len(g) == 1 and g[0].words == []
                assert g[0].tokens.dtype == torch.int64 and g[0].tokens.tolist() == w[0].tokens.tolist()
                assert g[0].timesteps.tolist() == w[0].timesteps.tolist()
                assert g[0].score == w[0].score


# ---- DataParallelRunner with an empty shard and with shards of different shapes (round-2 advisor finding) ----
def _uneven_runner_worker(rank, world, port, result_path):
    from oracle import allophant_oracle as O
    from allophant_amd.parallel import DataParallelRunner

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    spec, state, _, _, tfi = _setup()
    offsets = synthetic.category_offsets(spec)

    def predict(batch):
        out, flen = O.predict(batch.audio_features, batch.lengths, state, spec, tfi, offsets)
        return _flat_predictions(out, flen)

    names = [("syllabic", 4), ("long", 4), ("phoneme", 7)]
    results = []
    # (utterances, ragged): 1 utterance on 2 ranks = an empty shard on rank 1; 3 ragged = shards of different (N, L);
    # 4 equal = the flat fast path again, after the fallbacks
    for step, (total, ragged) in enumerate([(1, False), (3, True), (4, False)]):
        runner = DataParallelRunner(predict, torch.device("cpu"), dst=0, total_utterances=total, outputs=names,
                                    aliases={"phone": "phoneme"})
        audio, lengths = synthetic.make_audio(total, 2400, seed=70 + step, ragged=ragged)
        shard = shard_batch(Batch(audio, lengths, torch.zeros(total, dtype=torch.long)), rank, world, spec=spec)
        if total == 1 and rank == 1:
            assert shard is None
        got = runner.step(shard)
        got = got if got is not None else runner.drain()
        results.append(None if got is None else {"outputs": dict(got.outputs), "lengths": got.lengths.cpu()})
    if rank == 0:
        assert all(r is not None for r in results)
        torch.save(results, result_path)
    else:
        assert all(r is None for r in results)
    # the padded gather without the size of the global batch is refused at construction, not in the middle of a collective
    try:
        DataParallelRunner(predict, torch.device("cpu"), flat=False)
        raise AssertionError("flat=False without total_utterances must be refused")
    except ValueError:
        pass
    dist.barrier()
    dist.destroy_process_group()


def test_runner_with_empty_and_uneven_shards(tmp_path):
    from oracle import allophant_oracle as O

    result_path = str(tmp_path / "uneven.pt")
    mp.spawn(_uneven_runner_worker, args=(2, _free_port(), result_path), nprocs=2, join=True)
    got = torch.load(result_path)
    spec, state, _, _, tfi = _setup()
    for step, (total, ragged) in enumerate([(1, False), (3, True), (4, False)]):
        audio, lengths = synthetic.make_audio(total, 2400, seed=70 + step, ragged=ragged)
        ref, ref_len = O.predict(audio, lengths, state, spec, tfi, synthetic.category_offsets(spec))
        g = got[step]
        assert list(g["outputs"]) == list(ref), step
        assert g["lengths"].tolist() == ref_len.tolist(), step
        for name in ref:
            valid = (torch.arange(ref[name].shape[0]).unsqueeze(1) < ref_len.unsqueeze(0)).unsqueeze(-1)
            assert g["outputs"][name].shape == ref[name].shape, (step, name)
            assert ((g["outputs"][name] - ref[name]).abs() * valid).max().item() < 1e-4, (step, name)


# ---- one runner across steps [equal, equal, ragged, smaller last batch, equal] (round-3 advisor finding) ----
_MIXED_STEPS = [(4, False), (4, False), (3, True), (2, False), (4, False)]


def _mixed_runner_worker(rank, world, port, result_path):
    from oracle import allophant_oracle as O
    from allophant_amd.parallel import DataParallelRunner

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    spec, state, _, _, tfi = _setup()
    offsets = synthetic.category_offsets(spec)

    def predict(batch):
        out, flen = O.predict(batch.audio_features, batch.lengths, state, spec, tfi, offsets)
        return _flat_predictions(out, flen)

    # total_utterances of the constructor is the usual batch size; the ragged and the smaller batch pass their own
    runner = DataParallelRunner(predict, torch.device("cpu"), dst=0, total_utterances=4,
                                outputs=[("syllabic", 4), ("long", 4), ("phoneme", 7)], aliases={"phone": "phoneme"})
    delivered = []
    for step, (total, ragged) in enumerate(_MIXED_STEPS):
        audio, lengths = synthetic.make_audio(total, 2400, seed=90 + step, ragged=ragged)
        shard = shard_batch(Batch(audio, lengths, torch.zeros(total, dtype=torch.long)), rank, world, spec=spec)
        delivered.append(runner.step(shard, total_utterances=total))
    delivered.append(runner.drain())
    assert runner.drain() is None  # nothing is handed out twice
    assert delivered[0] is None    # step 0's gather is still in flight when step 0 returns
    if rank == 0:
        results = [d for d in delivered if d is not None]
        assert runner.completed == len(_MIXED_STEPS) == len(results), (runner.completed, len(results))
        torch.save([{"outputs": dict(r.outputs), "lengths": r.lengths.cpu()} for r in results], result_path)
    else:
        assert all(d is None for d in delivered)
    dist.barrier()
    dist.destroy_process_group()


def test_runner_delivers_every_step_once_and_in_order(tmp_path):
    """An overlapped runner that has to fall back to the padded gather while the previous step's flat gather is in flight
    must not drop that step: across [equal, equal, ragged, smaller, equal] every step's global predictions come out exactly
    once, in step order."""
    from oracle import allophant_oracle as O

    result_path = str(tmp_path / "mixed.pt")
    mp.spawn(_mixed_runner_worker, args=(2, _free_port(), result_path), nprocs=2, join=True)
    got = torch.load(result_path)
    spec, state, _, _, tfi = _setup()
    assert len(got) == len(_MIXED_STEPS)
    for step, (total, ragged) in enumerate(_MIXED_STEPS):
        audio, lengths = synthetic.make_audio(total, 2400, seed=90 + step, ragged=ragged)
        ref, ref_len = O.predict(audio, lengths, state, spec, tfi, synthetic.category_offsets(spec))
        g = got[step]
        assert list(g["outputs"]) == list(ref), step
        assert g["lengths"].tolist() == ref_len.tolist(), step
        for name in ref:
            valid = (torch.arange(ref[name].shape[0]).unsqueeze(1) < ref_len.unsqueeze(0)).unsqueeze(-1)
            assert g["outputs"][name].shape == ref[name].shape, (step, name)
            assert ((g["outputs"][name] - ref[name]).abs() * valid).max().item() < 1e-4, (step, name)


def _overflow_worker(rank, world, port):
    from allophant_amd.estimator import Decoded
    from allophant_amd.parallel import gather_decoded

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full = _global_decoded(4, seed=78)
    full.counts[:, 3] = 20  # an alignment of 20 tokens on rank 1
    lo, hi = shard_bounds(4, world)[rank]
    local = Decoded(full.names, full.tokens[:, lo:hi], full.timesteps[:, lo:hi], full.counts[:, lo:hi], full.scores[:, lo:hi])
    try:
        gather_decoded(local, ["phoneme"], 4, torch.device("cpu"), dst=0, capacity=8)
        raised = False
    except ValueError:
        raised = True
    # rank 0 sees the over-long count in the gathered block, rank 1 in its own alignments
    assert raised, rank
    dist.barrier()
    dist.destroy_process_group()


def test_decoded_gather_refuses_alignments_beyond_the_capacity():
    mp.spawn(_overflow_worker, args=(2, _free_port()), nprocs=2, join=True)


# ---- variants whose valid frames depend on the padded length (group norm over time / no attention mask) ----
def _variant_setup():
    encoder = S.tiny_encoder(1)
    encoder.update(feat_extract_norm="group", conv_bias=False, stable_layer_norm=False, use_attention_mask=False)
    spec = S.multitask_spec(encoder, ["syllabic"], embedding_size=16, train_phonemes=9, n_features=5)
    state = synthetic.make_state_dict(spec, seed=7)
    audio, lengths = synthetic.make_audio(4, 3000, seed=13, ragged=True)
    tfi = synthetic.make_inventory(spec, 6, seed=7)
    return spec, state, audio, lengths, tfi


def _variant_predict(spec, state, batch, tfi):
    from oracle import allophant_oracle as O

    out, flen = O.predict(batch.audio_features, batch.lengths, state, spec, tfi, synthetic.category_offsets(spec),
                          padded=bool(getattr(batch, "_padded", False)))
    return Predictions(out, flen)


def _variant_worker(rank, world, port, result_path):
    from allophant_amd.parallel import padding_sensitive

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    spec, state, audio, lengths, tfi = _variant_setup()
    assert padding_sensitive(spec)
    full = Batch(audio, lengths, torch.zeros(len(lengths), dtype=torch.long))
    names = [("syllabic", 4), ("phoneme", 7)]
    # the spec decides (round-5 advisor finding: the default used to be "re-pad", silently wrong for these variants) ...
    kept = data_parallel_predict(lambda b: _variant_predict(spec, state, b, tfi), full, names, torch.device("cpu"), dst=0, spec=spec)
    # ... an explicit keep_length overrides it, and giving neither is refused
    repadded = data_parallel_predict(lambda b: _variant_predict(spec, state, b, tfi), full, names, torch.device("cpu"), dst=0,
                                     keep_length=False)
    try:
        shard_batch(full, rank, world)
        raise AssertionError("shard_batch without spec / keep_length must be refused")
    except TypeError:
        pass
    if rank == 0:
        torch.save({"kept": kept.outputs, "repadded": repadded.outputs, "lengths": kept.lengths}, result_path)
    dist.barrier()
    dist.destroy_process_group()


def test_padding_sensitive_variants_shard_with_the_global_length(tmp_path):
    """Group-norm / unmasked wav2vec 2.0: the single-device result depends on the padded length of the batch tensor, so the
    shards keep it (``shard_batch(keep_length=True)`` -> AMX_FLAG_PADDED); with re-padded shards the valid frames differ."""
    from allophant_amd.parallel import padding_sensitive

    assert not padding_sensitive(_setup()[0])  # XLS-R form: layer norm + attention mask
    world = 2
    result_path = str(tmp_path / "variant.pt")
    mp.spawn(_variant_worker, args=(world, _free_port(), result_path), nprocs=world, join=True)
    got = torch.load(result_path)
    spec, state, audio, lengths, tfi = _variant_setup()
    single = _variant_predict(spec, state, Batch(audio, lengths, torch.zeros(len(lengths), dtype=torch.long)), tfi)
    assert torch.equal(got["lengths"], single.lengths)
    worst_repadded = 0.0
    for name, expected in single.outputs.items():
        valid = (torch.arange(expected.shape[0]).unsqueeze(1) < single.lengths.unsqueeze(0)).unsqueeze(-1)
        assert got["kept"][name].shape == expected.shape
        assert ((got["kept"][name] - expected).abs() * valid).max().item() < 1e-4, name
        g = got["repadded"][name]
        worst_repadded = max(worst_repadded, ((g - expected[: g.shape[0]]).abs() * valid[: g.shape[0]]).max().item())
    assert worst_repadded > 1e-2  # what the advisor measured: re-padded shards are a different function for these specs


# ---- a range report (AMX_ERANGE -> FloatingPointError) on ONE rank must not strand the others in the gather ----
def _range_report_worker(rank, world, port, result_path):
    from oracle import allophant_oracle as O
    from allophant_amd.parallel import DataParallelRunner, RankError

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    spec, state, _, _, tfi = _setup()
    offsets = synthetic.category_offsets(spec)
    calls = {"n": 0}

    def predict(batch):
        # what Estimator.predict does when an EARLIER pass overflowed: it raises before enqueuing anything of this call
        calls["n"] += 1
        if rank == 1 and calls["n"] == 2:
            raise FloatingPointError("an EARLIER forward pass left the range of the planes")
        out, flen = O.predict(batch.audio_features, batch.lengths, state, spec, tfi, offsets)
        return _flat_predictions(out, flen)

    names = [("syllabic", 4), ("long", 4), ("phoneme", 7)]
    log = []
    # (a) the padded gather of data_parallel_predict: the reporting rank joins the collectives, then raises; rank 0 raises RankError
    audio, lengths = synthetic.make_audio(4, 2400, seed=21)
    full = Batch(audio, lengths, torch.zeros(4, dtype=torch.long))
    ok = data_parallel_predict(predict, full, names, torch.device("cpu"), dst=0, aliases={"phone": "phoneme"}, spec=spec)
    log.append("first ok" if (ok is not None) == (rank == 0) else "first wrong")
    try:
        data_parallel_predict(predict, full, names, torch.device("cpu"), dst=0, aliases={"phone": "phoneme"}, spec=spec)
        log.append("second returned")
    except RankError as exc:
        log.append("RankError " + str(exc)[:20])
    except FloatingPointError:
        log.append("FloatingPointError")
    # (c) a shard with more frames than the caller stated for the global batch (rank 1's utterances are the long ones): the same
    # route -- cropped into the collectives with its status set, ValueError behind them, RankError on the destination
    from allophant_amd.parallel import gather_predictions

    ragged_lengths = torch.tensor([1200, 1200, 2400, 2400])
    ragged = Batch(audio * (torch.arange(2400).unsqueeze(0) < ragged_lengths.unsqueeze(1)), ragged_lengths, torch.zeros(4, dtype=torch.long))
    local = predict(shard_batch(ragged, rank, world, spec=spec))
    try:
        gather_predictions(local, names, 4, torch.device("cpu"), dst=0, aliases={"phone": "phoneme"}, frames=S.frame_lengths([1200], spec)[0])
        log.append("third returned")
    except RankError:
        log.append("RankError")
    except ValueError as exc:
        log.append("ValueError " + str(exc)[:13])
    # (b) the flat, overlapped gather of DataParallelRunner (bench.py --gpus N): the status travels behind the frame lengths
    calls["n"] = 0
    runner = DataParallelRunner(predict, torch.device("cpu"), dst=0, verify_shapes=False)
    statuses = []
    for step in range(3):
        shard = shard_batch(full, rank, world, spec=spec)
        try:
            got = runner.step(shard)
            if got is not None:
                statuses.append(got._status.tolist())
        except FloatingPointError:
            statuses.append("raised")
    try:
        got = runner.drain()
        if got is not None:
            statuses.append(got._status.tolist())
            got.check_ranks()
    except FloatingPointError:
        statuses.append("raised")
    torch.save({"log": log, "statuses": statuses}, f"{result_path}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_a_range_report_on_one_rank_does_not_hang_the_gather(tmp_path):
    """Round-5 advisor finding: ``Estimator.predict`` raises ``FloatingPointError`` for an earlier pass BEFORE the collective, on
    one rank only -- the others then blocked in the gather forever.  Now the rank joins the step's collectives (its status word
    travels with the frame lengths) and raises afterwards; the destination raises ``RankError`` (padded gather) or finds the
    status in the assembled predictions (flat gather, ``Predictions.check_ranks``).  The test finishing at all is the point."""
    path = str(tmp_path / "range")
    mp.spawn(_range_report_worker, args=(2, _free_port(), path), nprocs=2, join=True)
    r0, r1 = torch.load(path + ".0"), torch.load(path + ".1")
    assert r0["log"][0] == "first ok" and r1["log"][0] == "first ok"
    assert r0["log"][1].startswith("RankError")       # the destination learns which rank reported
    assert r1["log"][1] == "FloatingPointError"       # the reporting rank raises its own error -- after the gather
    assert r0["log"][2] == "RankError" and r1["log"][2] == "ValueError a shard has 7", (r0["log"], r1["log"])
    # flat gather: step 1 of rank 1 reported; rank 0 sees status [0, 1] for that step and 0 otherwise; rank 1 raised once
    assert r1["statuses"].count("raised") == 1
    seen = [s for s in r0["statuses"] if s != "raised"]
    assert seen.count([0, 1]) == 1 and seen.count([0, 0]) == 2, r0["statuses"]
