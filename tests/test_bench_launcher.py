"""``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE starts its own ranks (round-2 review, item 2): the parent never
touches the GPU, runs ``python -m torch.distributed.run`` on the same file as a child process, relays rank 0's single JSON
line and exits non-zero when a rank fails.  Here on CPU: a stub rank script (gloo) stands in for the GPU ranks through the
test switch ``AMX_BENCH_CHILD_SCRIPT``; the real ranks must fail loudly without an MI355X.

The on-device form (`AMX_BENCH_FORCE_LAUNCH=1 AMX_BENCH_FORCE_DIST=1 python3 bench.py --gpus 1`: launcher -> one rank -> RCCL
gather on a one-rank group) is run from a fresh shell on the GPU box, not from pytest: a pytest process that has already
initialised the GPU must not start other programs on this pool (log: profiles/r03_launcher_on_box.log)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = '''
import json, os, sys
import torch
import torch.distributed as dist

world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
assert os.environ["MASTER_ADDR"] == "127.0.0.1"
dist.init_process_group("gloo")
t = torch.tensor([float(rank + 1)])
dist.all_reduce(t)
print(f"chatter from rank {rank}", flush=True)      # must not reach the launcher's stdout
if os.environ.get("STUB_FAIL_RANK") == str(rank):
    sys.exit(3)
dist.barrier()
if rank == 0:
    print(json.dumps({"n_gpus": world, "sum": t.item(), "argv": sys.argv[1:]}), flush=True)
dist.destroy_process_group()
'''


def _run(tmp_path, extra_env, args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=600, cwd=str(tmp_path))


def test_self_launch_relays_rank0_line(tmp_path):
    stub = tmp_path / "stub_rank.py"
    stub.write_text(STUB)
    done = _run(tmp_path, {"AMX_BENCH_CHILD_SCRIPT": str(stub)}, ["--gpus", "2", "--steps", "2", "--warmup", "1"])
    assert done.returncode == 0, done.stderr.decode()[-2000:]
    lines = [ln for ln in done.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["sum"] == 3.0
    assert line["argv"] == ["--gpus", "2", "--steps", "2", "--warmup", "1"]  # the ranks get the launcher's arguments
    assert "chatter from rank" in done.stderr.decode()


def test_self_launch_reports_a_failing_rank(tmp_path):
    stub = tmp_path / "stub_rank.py"
    stub.write_text(STUB)
    done = _run(tmp_path, {"AMX_BENCH_CHILD_SCRIPT": str(stub), "STUB_FAIL_RANK": "1"}, ["--gpus", "2"])
    assert done.returncode != 0
    assert not [ln for ln in done.stdout.decode().splitlines() if ln.strip().startswith("{")]


def test_real_ranks_fail_loudly_without_a_gpu(tmp_path):
    """No CPU fallback: without an MI355X every rank exits with the 'needs an MI355X' message and so does the launcher."""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("this is the no-GPU behaviour")
    done = _run(tmp_path, {}, ["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert done.returncode != 0
    assert "needs an MI355X" in done.stderr.decode()
    assert done.stdout.decode().strip() == ""

