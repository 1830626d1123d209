"""Two ranks through ``amx_forward`` + ``amx_gather_outputs`` on ONE GPU (the pool has no multi-GPU box): two processes play the
ranks of a 2-GPU data-parallel job on BASELINE config 2's batch -- 16 x 10 s each, XLS-R-300m shape -- and the test-only RCCL
stand-in (tests/fake_rccl/fake_rccl.c, transport mode: ncclSend / ncclRecv move the bytes rank's HBM -> host file -> root's HBM
at ncclGroupEnd) carries the exchange.  What runs for real: every line of ``amx_gather_outputs`` on a non-root and on a root
rank with a peer > 0 (``recv + peer * count`` offsets, the lengths exchange, one group), behind a real forward pass on the same
stream.  What does not: RCCL's own transport over xGMI.

Checked: the root's block r is BITWISE the flat output of a single-process pass over shard r; the assembled [T, 32, C] tensors
meet the single-process 32 x 10 s pass (bitwise where the products of the two batch sizes take the same K split, else within
1e-4 -- logged); frame lengths equal.  Contract: include/allophant_amx.h (amx_gather_outputs), SURVEY.md section 8(e).
"""
import os
import subprocess
import sys

import pytest
import torch

from allophant_amd import synthetic

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_RANK = r"""
import ctypes as C, os, sys
root_dir, rank, world, fake_so, msg_dir, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
sys.path.insert(0, root_dir)
os.environ["AMX_RCCL_LIBRARY"] = fake_so
import torch
import bench
from allophant_amd import lib as L, parallel, synthetic
from allophant_amd.estimator import Batch, Estimator
spec = bench.build_spec()
est = Estimator(spec, synthetic.make_state_dict(spec, seed=0), "cuda:0", "f16x3")
tfi = synthetic.make_inventory(spec, 27, seed=0)
audio, lengths = synthetic.make_audio(32, 160000, seed=1234)
shard = parallel.shard_batch(Batch(audio, lengths, torch.zeros(32, dtype=torch.long)), rank, world, spec=spec)
n_local = len(shard)
pred = est.predict(Batch(shard.audio_features.cuda(), shard.lengths, shard.language_ids), tfi)
fake = C.CDLL(fake_so, mode=C.RTLD_GLOBAL)
fake.fake_rccl_set_mode(1)
fake.fake_comm_create.restype = C.c_void_p
fake.fake_comm_create.argtypes = [C.c_int, C.c_int, C.c_char_p]
comm = C.c_void_p(fake.fake_comm_create(rank, world, msg_dir.encode()))
lib = L.load()
flat = pred._flat
frames = pred.lengths.to("cuda:0")
root = 0
recv = torch.full((world * flat.numel(),), float("nan"), device="cuda") if rank == root else None
recv_len = torch.full((world * n_local,), -1, dtype=torch.int64, device="cuda") if rank == root else None
stream = torch.cuda.current_stream().cuda_stream
rc = lib.amx_gather_outputs(comm, rank, world, root, C.c_void_p(flat.data_ptr()), flat.numel(),
                            C.c_void_p(recv.data_ptr()) if recv is not None else None, C.c_void_p(frames.data_ptr()), n_local,
                            C.c_void_p(recv_len.data_ptr()) if recv_len is not None else None, C.c_void_p(stream))
assert rc == 0, lib.amx_dist_last_error().decode()
torch.cuda.synchronize()
if rank == root:
    torch.save({"recv": recv.cpu(), "lengths": recv_len.cpu(), "count": flat.numel(), "n_local": n_local}, out_path)
est.close()
print("rank", rank, "done")
"""


def test_two_ranks_on_one_gpu_reproduce_the_single_process_batch(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import bench
    from allophant_amd import parallel
    from allophant_amd.estimator import Batch, Estimator

    fake_so = str(tmp_path / "libfake_rccl.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O2", "-o", fake_so, os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.c"), "-ldl"],
                   check=True)
    msg_dir = tmp_path / "messages"
    msg_dir.mkdir()
    out_path = str(tmp_path / "root.pt")
    script = tmp_path / "rank.py"
    script.write_text(_RANK)
    world = 2
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), fake_so, str(msg_dir), out_path],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    # meanwhile, the single-process results in this process: the two shards and the whole batch
    spec = bench.build_spec()
    est = Estimator(spec, synthetic.make_state_dict(spec, seed=0), "cuda:0", "f16x3")
    tfi = synthetic.make_inventory(spec, 27, seed=0)
    audio, lengths = synthetic.make_audio(32, 160000, seed=1234)
    whole_batch = Batch(audio, lengths, torch.zeros(32, dtype=torch.long))
    shard_flat, shard_len = [], []
    for r in range(world):
        sh = parallel.shard_batch(whole_batch, r, world, spec=spec)
        p = est.predict(Batch(sh.audio_features.cuda(), sh.lengths, sh.language_ids), tfi)
        torch.cuda.synchronize()
        shard_flat.append(p._flat.cpu().clone())
        shard_len.append(p.lengths.cpu().clone())
        layout = [(name, out.shape, (out.data_ptr() - p._flat.data_ptr()) // 4) for name, out in p.outputs.items()]
    whole = est.predict(Batch(audio.cuda(), lengths, whole_batch.language_ids), tfi)
    torch.cuda.synchronize()
    whole_out = {k: v.cpu() for k, v in whole.outputs.items()}
    whole_len = whole.lengths.cpu()
    est.close()
    logs = []
    for p in procs:
        out, _ = p.communicate(timeout=900)
        logs.append(out)
        assert p.returncode == 0, out[-3000:]
    got = torch.load(out_path)
    count, n_local = got["count"], got["n_local"]
    assert n_local == 16 and got["recv"].numel() == world * count
    for r in range(world):
        block = got["recv"][r * count: (r + 1) * count]
        assert torch.equal(block, shard_flat[r]), f"block of rank {r} is not the single-process shard output"
        assert torch.equal(got["lengths"][r * n_local: (r + 1) * n_local], shard_len[r])
    assert torch.equal(got["lengths"], whole_len)
    # the [T, N, C] tensors of the whole batch from the gathered blocks, at the offsets amx_output_layout reports per shard
    worst, bitwise = 0.0, True
    for name, shape, offset in layout:
        t, n, c = shape
        parts = [got["recv"][r * count + offset: r * count + offset + t * n * c].view(t, n, c) for r in range(world)]
        assembled = torch.cat(parts, dim=1)
        assert assembled.shape == whole_out[name].shape
        bitwise &= torch.equal(assembled, whole_out[name])
        worst = max(worst, float((assembled - whole_out[name]).abs().max()))
    print(f"[two-rank gather] assembled vs single-process 32 x 10 s: max abs {worst:.3g}, bitwise {bitwise}")
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "two_rank_gather.log"), "w") as f:
            f.write(f"two ranks x 16 x 10 s on one GPU through amx_forward + amx_gather_outputs (fake RCCL transport):\n"
                    f"  blocks bitwise equal to the single-process shard outputs: True\n"
                    f"  assembled [T, 32, C] vs the single-process 32 x 10 s pass: max abs {worst:.3g}, bitwise {bitwise}\n")
    assert worst < 1e-4
