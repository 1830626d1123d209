"""The multi-rank branches of ``amx_gather_outputs`` (csrc/amx_dist.hip; contract include/allophant_amx.h) without multi-GPU
hardware: a test-only stand-in for librccl (tests/fake_rccl/fake_rccl.c, loaded through ``AMX_RCCL_LIBRARY``) records every
``ncclGroupStart/End`` / ``ncclSend`` / ``ncclRecv`` the library issues, for any (rank, world, root).  No GPU needed: the
gather itself makes no HIP call, and the recorder never dereferences a pointer.  The reference has nothing to mirror
(run.py:576-580 is single-device); the contract is SURVEY.md section 8(e).

Each case runs in a subprocess: the library binds its RCCL once per process.
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE_SRC = os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.c")

_DRIVER = r"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.environ["AMX_ROOT"])
from allophant_amd import lib as L
lib = L.load()
fake = C.CDLL(os.environ["AMX_RCCL_LIBRARY"], mode=C.RTLD_GLOBAL) if os.path.exists(os.environ["AMX_RCCL_LIBRARY"]) else None
class Call(C.Structure):
    _fields_ = [("kind", C.c_int), ("buffer", C.c_uint64), ("count", C.c_uint64), ("dtype", C.c_int), ("peer", C.c_int),
                ("comm", C.c_uint64), ("stream", C.c_uint64), ("group_depth", C.c_int)]
cases = json.loads(os.environ["AMX_CASES"])
results = []
for case in cases:
    if fake is not None:
        fake.fake_rccl_set_mode(0)
        fake.fake_rccl_reset()
    vp = C.c_void_p
    rc = lib.amx_gather_outputs(vp(case["comm"]), case["rank"], case["world"], case["root"], vp(case["send"]), case["count"],
                                vp(case["recv"]) if case["recv"] else None, vp(case["send_lengths"]), case["n_local"],
                                vp(case["recv_lengths"]) if case["recv_lengths"] else None, vp(case["stream"]))
    calls = []
    if fake is not None:
        fake.fake_rccl_get.argtypes = [C.c_int, C.POINTER(Call)]
        for i in range(fake.fake_rccl_count()):
            c = Call()
            fake.fake_rccl_get(i, C.byref(c))
            calls.append({k: getattr(c, k) for k, _ in Call._fields_})
    results.append({"rc": rc, "error": lib.amx_dist_last_error().decode(), "calls": calls})
print("RESULT " + json.dumps(results))
"""


@pytest.fixture(scope="module")
def fake_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O2", "-o", out, FAKE_SRC, "-ldl"], check=True)
    return out


def _run(cases, rccl_library):
    env = dict(os.environ, AMX_ROOT=ROOT, AMX_RCCL_LIBRARY=rccl_library, AMX_CASES=json.dumps(cases))
    proc = subprocess.run([sys.executable, "-c", _DRIVER], env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    line = [l for l in proc.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def _case(rank, world, root, count=1000, n_local=4, recv=True):
    # distinct, recognisable "addresses" (never dereferenced by the library or the recorder)
    return {"comm": 0x1234, "rank": rank, "world": world, "root": root, "send": 0x10000000, "count": count,
            "recv": 0x40000000 if recv else 0, "send_lengths": 0x20000000, "n_local": n_local,
            "recv_lengths": 0x30000000 if recv else 0, "stream": 0x77}


KIND = {0: "start", 1: "end", 2: "send", 3: "recv"}
F32, I64 = 7, 4  # ncclFloat32, ncclInt64 (rccl.h)


def _check_calls(case, calls):
    kinds = [KIND[c["kind"]] for c in calls]
    assert kinds[0] == "start" and kinds[-1] == "end" and kinds.count("start") == 1 and kinds.count("end") == 1, kinds  # ONE group
    inner = calls[1:-1]
    assert all(c["group_depth"] == 1 and c["comm"] == case["comm"] and c["stream"] == case["stream"] for c in inner)
    recvs = [c for c in inner if KIND[c["kind"]] == "recv"]
    sends = [c for c in inner if KIND[c["kind"]] == "send"]
    # every rank: its block and its lengths to the root, in that order
    assert [(c["buffer"], c["count"], c["dtype"], c["peer"]) for c in sends] == \
        [(case["send"], case["count"], F32, case["root"]), (case["send_lengths"], case["n_local"], I64, case["root"])]
    if case["rank"] == case["root"]:
        # the root: one pair of receives per peer, rank order, block r at recv + r * count floats, lengths at + r * n_local
        want = []
        for peer in range(case["world"]):
            want.append((case["recv"] + 4 * peer * case["count"], case["count"], F32, peer))
            want.append((case["recv_lengths"] + 8 * peer * case["n_local"], case["n_local"], I64, peer))
        assert [(c["buffer"], c["count"], c["dtype"], c["peer"]) for c in recvs] == want
        # receives are posted before the root's own sends (a send to self inside the group is a copy)
        assert kinds.index("send") > max(i for i, k in enumerate(kinds) if k == "recv")
    else:
        assert recvs == []


def test_every_rank_of_an_eight_gpu_gather_issues_the_contracted_calls(fake_lib):
    """world = 8 (BASELINE config 3), root 0 and a root in the middle: root and non-root branches, peers > 0."""
    cases = [_case(rank, 8, root, recv=(rank == root)) for root in (0, 5) for rank in range(8)]
    results = _run(cases, fake_lib)
    for case, res in zip(cases, results):
        assert res["rc"] == 0, res["error"]
        _check_calls(case, res["calls"])
    # config 3's real sizes: 4 x 10 s per rank, 38 outputs -> count = 499 * 4 * sum(C); offsets stay exact beyond 2^31 bytes
    big = _case(0, 8, 0, count=3 * (1 << 28), n_local=4)
    res = _run([big], fake_lib)[0]
    assert res["rc"] == 0
    _check_calls(big, res["calls"])


def test_two_ranks_and_degenerate_worlds(fake_lib):
    cases = [_case(0, 2, 0), _case(1, 2, 0, recv=False), _case(0, 2, 1, recv=False), _case(1, 2, 1), _case(0, 1, 0),
             _case(2, 4, 3, count=0, recv=False), _case(3, 4, 3, n_local=0)]
    results = _run(cases, fake_lib)
    for case, res in zip(cases, results):
        assert res["rc"] == 0, res["error"]
        calls = res["calls"]
        if case["count"] == 0 or case["n_local"] == 0:
            # an empty block or no lengths: that half of the exchange is skipped on every rank, the group still brackets
            kinds = [KIND[c["kind"]] for c in calls]
            assert kinds[0] == "start" and kinds[-1] == "end"
            dtypes = {c["dtype"] for c in calls[1:-1]}
            assert dtypes == ({I64} if case["count"] == 0 else {F32})
            continue
        _check_calls(case, calls)


def test_argument_errors_leave_rccl_untouched(fake_lib):
    bad = [dict(_case(0, 2, 0), comm=0), dict(_case(2, 2, 0)), dict(_case(0, 2, 2)), dict(_case(0, 2, 0), send=0),
           dict(_case(0, 2, 0), recv=0), dict(_case(0, 2, 0), recv_lengths=0), dict(_case(0, 2, 0), count=-1)]
    for res in _run(bad, fake_lib):
        assert res["rc"] == -1 and res["calls"] == [], res  # AMX_EINVAL before any RCCL call


def test_a_named_rccl_library_that_does_not_load_is_an_error_not_a_fallback(tmp_path):
    """a failed dlopen of AMX_RCCL_LIBRARY must not silently bind whatever librccl the process holds."""
    missing = str(tmp_path / "no_such_librccl.so")
    res = _run([_case(0, 2, 0)], missing)[0]
    assert res["rc"] == -3 and "could not be loaded" in res["error"] and "no_such_librccl.so" in res["error"], res
