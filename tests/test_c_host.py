"""A compiled C host of the boundary (round-5 review, item 7; the reference's own FFI convention is a compiled binding,
src/lib.rs:9-18): ``tests/c_host/host.c`` is C99, includes ``include/allophant_amx.h``, links ``liballophant_amx.so`` and runs

    amx_create -> amx_set_inventory -> amx_output_layout -> amx_forward(AMX_FLAG_HOST_IO) -> amx_greedy_ctc

on golden g1 (a tiny multitask model whose outputs the REAL reference produced, ``oracle/gen_golden.py``).  CPU part: the host
compiles with ``gcc -std=c99 -pedantic -Wall -Werror`` against the header, links every declared entry point and reports the struct
sizes the ctypes binding assumes.  GPU part: its log-probabilities meet the golden within 1e-3 and its alignments are the
reference decoder's."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

from allophant_amd import lib as L
from tests.golden_util import Golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c_host", "host.c")
ROCM_LIB = "/opt/rocm/lib"


def _build(tmp_path):
    exe = str(tmp_path / "amx_c_host")
    lib_dir = os.path.dirname(L.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
           "-L", lib_dir, "-l:" + os.path.basename(L.LIB_PATH), "-L", ROCM_LIB, "-lamdhip64",
           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + ROCM_LIB]
    done = subprocess.run(cmd, capture_output=True, text=True)
    assert done.returncode == 0, done.stderr
    return exe


def test_c_host_compiles_and_links_every_entry_point(tmp_path):
    if not os.path.exists(L.LIB_PATH):
        pytest.skip("liballophant_amx.so not built")
    exe = _build(tmp_path)
    done = subprocess.run([exe, "--link-check"], capture_output=True, text=True)
    assert done.returncode == 0, done.stderr
    words = done.stdout.replace(",", " ").split()
    assert int(words[1]) == L.AMX_ABI_VERSION
    assert int(words[2]) == len(L.EXPORTS)  # the host names every symbol the header declares
    assert int(words[words.index("sizeof(amx_config)") + 1]) == C.sizeof(L.AmxConfig)
    assert int(words[words.index("sizeof(amx_class_desc)") + 1]) == C.sizeof(L.AmxClassDesc)
    assert int(words[words.index("sizeof(amx_output_desc)") + 1]) == C.sizeof(L.AmxOutputDesc)


def _write_model(path, spec, state, tfi, offsets, audio, lengths):
    from allophant_amd.estimator import _spec_to_structs

    cfg, descs = _spec_to_structs(spec, "f16x3")
    with open(path, "wb") as f:
        f.write(b"AMXH")
        f.write(bytes(cfg))
        f.write(struct.pack("<i", len(descs)))
        f.write(bytes(descs))
        f.write(struct.pack("<i", len(state)))
        for name, tensor in state.items():
            data = np.ascontiguousarray(tensor.detach().cpu().numpy(), dtype=np.float32)
            encoded = name.encode()
            f.write(struct.pack("<i", len(encoded)))
            f.write(encoded)
            f.write(struct.pack("<q", data.size))
            f.write(data.tobytes())
        if tfi is None:
            f.write(struct.pack("<ii", 0, 0))
        else:
            t = np.ascontiguousarray(tfi.numpy(), dtype=np.int64)
            f.write(struct.pack("<ii", t.shape[0], t.shape[1]))
            f.write(t.tobytes())
            f.write(np.ascontiguousarray(offsets.numpy(), dtype=np.int64).tobytes())
        a = np.ascontiguousarray(audio.numpy(), dtype=np.float32)
        f.write(struct.pack("<iq", a.shape[0], a.shape[1]))
        f.write(a.tobytes())
        f.write(np.ascontiguousarray(lengths.numpy(), dtype=np.int64).tobytes())


def _read_result(path, n):
    with open(path, "rb") as f:
        blob = f.read()
    assert blob[:4] == b"AMXR"
    n_out, t, total = struct.unpack_from("<iqq", blob, 4)
    at = 4 + 4 + 8 + 8
    desc_size = C.sizeof(L.AmxOutputDesc)
    layout = []
    for _ in range(n_out):
        d = L.AmxOutputDesc.from_buffer_copy(blob[at: at + desc_size])
        layout.append((d.name.decode(), d.classes, d.offset))
        at += desc_size
    out = np.frombuffer(blob, dtype=np.float32, count=total, offset=at)
    at += 4 * total
    frames = np.frombuffer(blob, dtype=np.int64, count=n, offset=at)
    at += 8 * n
    decoded = {}
    for name, _, _ in layout:
        for i in range(n):
            (count,) = struct.unpack_from("<i", blob, at)
            at += 4
            tokens = np.frombuffer(blob, dtype=np.int64, count=count, offset=at)
            at += 8 * count
            timesteps = np.frombuffer(blob, dtype=np.int64, count=count, offset=at)
            at += 8 * count
            (score,) = struct.unpack_from("<f", blob, at)
            at += 4
            decoded[(name, i)] = (tokens, timesteps, score)
    assert at == len(blob)
    return layout, t, out, frames, decoded


@pytest.mark.gpu
def test_c_host_runs_a_reference_golden(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    g = Golden("g1_tiny_multitask")
    exe = _build(tmp_path)
    model, result = str(tmp_path / "model.bin"), str(tmp_path / "result.bin")
    _write_model(model, g.spec, g.state_dict(), g.tfi, g.category_offsets, g.audio, g.lengths)
    done = subprocess.run([exe, model, result], capture_output=True, text=True)
    assert done.returncode == 0, done.stderr + done.stdout
    n = g.audio.shape[0]
    layout, t, out, frames, decoded = _read_result(result, n)
    assert frames.tolist() == g.frame_lengths.tolist()
    assert [name for name, _, _ in layout] == g.output_names
    for name, classes, offset in layout:
        got = torch.from_numpy(out[offset: offset + t * n * classes].reshape(t, n, classes).copy())
        want = g.logprobs(name)
        assert got.shape == want.shape
        valid = (torch.arange(t).unsqueeze(1) < g.frame_lengths.unsqueeze(0)).unsqueeze(-1)
        assert ((got - want).abs() * valid).max().item() < 1e-3, name
        for i in range(n):
            tokens, timesteps, score = g.tokens(name, i)
            got_tokens, got_timesteps, got_score = decoded[(name, i)]
            assert got_tokens.tolist() == tokens.tolist(), (name, i)
            assert got_timesteps.tolist() == timesteps.tolist(), (name, i)
            assert abs(got_score - score) < 1e-3 * max(1, int(g.frame_lengths[i])), (name, i)
