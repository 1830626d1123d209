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
                                     aliases={"phone": "phoneme"})
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
