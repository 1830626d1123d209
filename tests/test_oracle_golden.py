"""The CPU oracle (oracle/allophant_oracle.py, oracle/oracle_int.c) against the golden vectors generated from the REAL
reference by oracle/gen_golden.py.  CPU only."""
import ctypes as C
import json
import os
import subprocess

import numpy as np
import pytest
import torch

from allophant_amd import spec as S, synthetic
from golden_util import GOLDEN_DIR, Golden, max_abs_valid_bm, max_abs_valid_tm
from oracle import allophant_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TINY = ["g1_tiny_multitask", "g2_tiny_hierarchical", "g2b_tiny_hierarchical_blanks", "g5_tiny_baseline",
        "g8_tiny_time_layer",  # g8: time-layer (ProjectingMultiheadAttention) classifiers, with and without positions
        # the group-norm / post-LN wav2vec 2.0 variant (wav2vec2-base / -large): g11 as the reference runs such a model
        # (attention_mask=None), g11b with the attention mask and a conv bias
        "g11_tiny_groupnorm_postln", "g11b_tiny_groupnorm_masked",
        # head dimensions other than 64 (round 6): 80 (XLS-R 1B's) and 32
        "g13_tiny_head_dim_80", "g13b_tiny_head_dim_32",
        # `add_adapter=True` (round 6): the reference owns and runs a Wav2Vec2Adapter and reads nothing of it -- the state dict holds
        # its weights, the outputs and frame counts are those of the encoder alone
        "g15_tiny_adapter"]


@pytest.mark.parametrize("name", TINY)
def test_oracle_matches_reference_tiny(name):
    g = Golden(name)
    out, flen, inter = O.predict(g.audio, g.lengths, g.state_dict(), g.spec, g.tfi, g.category_offsets, True,
                                 keep_intermediates=True)
    assert list(out.keys()) == g.output_names == S.output_names(g.spec)
    assert torch.equal(flen, g.frame_lengths)
    for k in g.output_names:
        assert max_abs_valid_tm(out[k], g.logprobs(k), g.frame_lengths) < 2e-5, k
    raw, _ = O.predict(g.audio, g.lengths, g.state_dict(), g.spec, g.tfi, g.category_offsets, False)
    for k in g.output_names:
        assert max_abs_valid_tm(raw[k], g.logits(k), g.frame_lengths) < 2e-5, k
    assert max_abs_valid_bm(inter["conv_out"], g.conv_out(), g.frame_lengths) < 1e-5
    for i in g.hidden_indices():
        assert max_abs_valid_bm(inter["hidden_states"][i], g.hidden(i), g.frame_lengths) < 2e-5, i
    # greedy decode of the oracle's own log-probs reproduces the reference decoder's alignments (integer-exact)
    for k in g.output_names:
        hyps = O.greedy_ctc(out[k].transpose(0, 1).contiguous(), flen)
        for i, (tokens, timesteps, score) in enumerate(hyps):
            et, es, esc = g.tokens(k, i)
            assert torch.equal(tokens, et) and torch.equal(timesteps, es), (k, i)
            assert abs(float(score) - esc) < 1e-3


def test_oracle_matches_reference_xlsr_shape():
    """Full XLS-R-300m shape, procedural weights, 2 x 3 s ragged (sub-sampled golden tensors)."""
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    g = Golden("g3_xlsr_multitask")
    out, flen, inter = O.predict(g.audio, g.lengths, g.state_dict(), g.spec, g.tfi, g.category_offsets, True,
                                 keep_intermediates=True)
    assert list(out.keys()) == g.output_names
    assert len(g.output_names) == 38 and g.output_names[-2:] == ["phone", "phoneme"]
    assert torch.equal(flen, g.frame_lengths)
    worst = max(max_abs_valid_tm(out[k], g.logprobs(k), g.frame_lengths) for k in g.output_names)
    assert worst < 2e-4, worst
    assert max_abs_valid_bm(inter["conv_out"][:, :, ::8], g.conv_out(), g.frame_lengths) < 1e-5
    for i in g.hidden_indices():
        assert max_abs_valid_bm(inter["hidden_states"][i][:, :, ::8], g.hidden(i), g.frame_lengths) < 1e-4, i
    assert torch.equal(out["phone"], out["phoneme"])


def test_oracle_matches_reference_wav2vec2_base_shape():
    """Full wav2vec2-base shape (768 / 12 / 12 / 3072) of the group-norm / post-LN variant, attention_mask=None, procedural
    weights, 2 x 3 s ragged (sub-sampled golden tensors)."""
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    g = Golden("g12_w2v2base_multitask")
    assert g.spec["feat_extract_norm"] == "group" and not g.spec["stable_layer_norm"] and not g.spec["use_attention_mask"]
    out, flen, inter = O.predict(g.audio, g.lengths, g.state_dict(), g.spec, g.tfi, g.category_offsets, True,
                                 keep_intermediates=True)
    assert list(out.keys()) == g.output_names and len(g.output_names) == 38
    assert torch.equal(flen, g.frame_lengths)
    worst = max(max_abs_valid_tm(out[k], g.logprobs(k), g.frame_lengths) for k in g.output_names)
    assert worst < 2e-4, worst
    assert max_abs_valid_bm(inter["conv_out"][:, :, ::8], g.conv_out(), g.frame_lengths) < 1e-5
    for i in g.hidden_indices():
        assert max_abs_valid_bm(inter["hidden_states"][i][:, :, ::8], g.hidden(i), g.frame_lengths) < 1e-4, i


def _g4():
    return np.load(os.path.join(GOLDEN_DIR, "g4_integer.npz"))


def test_frame_lengths_golden():
    z = _g4()
    enc = S.xlsr_300m_encoder()
    lin = torch.from_numpy(z["lengths_in"])
    expected = torch.from_numpy(z["lengths_out"])
    assert torch.equal(O.downsampled_lengths(lin, enc["conv_kernel"], enc["conv_stride"]), expected)
    assert S.frame_lengths(lin.tolist(), enc) == expected.tolist()
    # known values: 400 samples = one frame, 3 s -> 149, 10 s -> 499, 60 s -> 2999
    table = dict(zip(lin.tolist(), expected.tolist()))
    assert table[400] == 1 and table[48000] == 149 and table[160000] == 499 and table[960000] == 2999


def test_mask_golden():
    z = _g4()
    assert torch.equal(O.mask_sequence(torch.from_numpy(z["mask_lengths"])), torch.from_numpy(z["mask"]))


def test_greedy_ctc_golden():
    z = _g4()
    for ci in z["ctc_cases"]:
        lp = torch.from_numpy(z[f"ctc/{ci}/logprobs"])
        ln = torch.from_numpy(z[f"ctc/{ci}/lengths"])
        hyps = O.greedy_ctc(lp, ln)
        for i, (tokens, timesteps, score) in enumerate(hyps):
            assert torch.equal(tokens, torch.from_numpy(z[f"ctc/{ci}/tokens/{i}"]))
            assert torch.equal(timesteps, torch.from_numpy(z[f"ctc/{ci}/timesteps/{i}"]))
            assert abs(float(score) - float(z[f"ctc/{ci}/score/{i}"])) < 1e-3


def test_evaluation_order_golden():
    graphs = json.loads(bytes(_g4()["graphs_json"]).decode())
    assert len(graphs) >= 10
    for g in graphs:
        names = [c["name"] for c in g["classes"]]
        assert [names[i] for i in O.topological_order(g["classes"])] == g["order"]
        assert [names[i] for i in S.evaluation_order(g["classes"])] == g["order"]


@pytest.fixture(scope="module")
def oracle_c():
    out_dir = os.path.join(ROOT, "oracle", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "liboracle_int.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "oracle", "oracle_int.c")])
    return C.CDLL(so)


def test_c_oracle_frame_lengths(oracle_c):
    z = _g4()
    enc = S.xlsr_300m_encoder()
    lin = np.ascontiguousarray(z["lengths_in"], dtype=np.int64)
    out = np.zeros_like(lin)
    k = np.array(enc["conv_kernel"], dtype=np.int32)
    s = np.array(enc["conv_stride"], dtype=np.int32)
    oracle_c.oracle_frame_lengths(lin.ctypes.data_as(C.c_void_p), len(lin), k.ctypes.data_as(C.c_void_p),
                                  s.ctypes.data_as(C.c_void_p), len(k), out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, z["lengths_out"])


def test_c_oracle_greedy_ctc(oracle_c):
    z = _g4()
    for ci in z["ctc_cases"]:
        lp = np.ascontiguousarray(z[f"ctc/{ci}/logprobs"], dtype=np.float32)
        ln = np.ascontiguousarray(z[f"ctc/{ci}/lengths"], dtype=np.int64)
        n, t, c = lp.shape
        tokens = np.zeros((n, t), dtype=np.int64)
        timesteps = np.zeros((n, t), dtype=np.int64)
        counts = np.zeros(n, dtype=np.int32)
        scores = np.zeros(n, dtype=np.float64)
        oracle_c.oracle_greedy_ctc(lp.ctypes.data_as(C.c_void_p), ln.ctypes.data_as(C.c_void_p), n, t, c, C.c_int64(0),
                                   tokens.ctypes.data_as(C.c_void_p), timesteps.ctypes.data_as(C.c_void_p),
                                   counts.ctypes.data_as(C.c_void_p), scores.ctypes.data_as(C.c_void_p))
        for i in range(n):
            k = counts[i]
            assert np.array_equal(tokens[i, :k], z[f"ctc/{ci}/tokens/{i}"])
            assert np.array_equal(timesteps[i, :k], z[f"ctc/{ci}/timesteps/{i}"])
            assert abs(scores[i] - float(z[f"ctc/{ci}/score/{i}"])) < 1e-3


def test_oracle_matches_reference_xlsr_1b_width():
    """Golden g14 (round 6): hidden 1280 / 16 heads of 80 / 80 channels per positional-convolution group -- the width of XLS-R 1B --
    on two layers; sub-sampled golden tensors."""
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    g = Golden("g14_xlsr1b_width")
    assert g.spec["hidden"] == 1280 and g.spec["hidden"] // g.spec["heads"] == 80
    out, flen, inter = O.predict(g.audio, g.lengths, g.state_dict(), g.spec, g.tfi, g.category_offsets, True, keep_intermediates=True)
    assert list(out.keys()) == g.output_names and torch.equal(flen, g.frame_lengths)
    worst = max(max_abs_valid_tm(out[k], g.logprobs(k), g.frame_lengths) for k in g.output_names)
    assert worst < 5e-5, worst
    for i in g.hidden_indices():
        assert max_abs_valid_bm(inter["hidden_states"][i][:, :, ::8], g.hidden(i), g.frame_lengths) < 5e-5, i
