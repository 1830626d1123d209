"""Developer probe: max-abs log-prob error against the CPU oracle for the weight families of tests/test_gpu_range.py, with
and without the per-tensor power-of-two pack scales (AMX_NO_PACK_SCALE=1), per precision mode.

    python tools/range_probe.py            # run twice: plain and with AMX_NO_PACK_SCALE=1
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

import test_gpu_range as R
from allophant_amd import spec as S, synthetic
from allophant_amd.estimator import Batch, Estimator
from oracle import allophant_oracle as O

spec = S.multitask_spec(S.xlsr_300m_encoder(), allophone_layer=True)
spec["shared_phones"] = 80
tfi = synthetic.make_inventory(spec, 27, seed=3)
audio, lengths = synthetic.make_audio(2, 48000, seed=777, ragged=True)
batch = Batch(audio.cuda(), lengths, torch.zeros(2, dtype=torch.long))
tag = "unscaled planes" if os.environ.get("AMX_NO_PACK_SCALE") == "1" else "pack scales"
for kind in ["plain", "scales", "student_t", "ln_gain", "outlier", "tiny_weights"]:
    if kind == "plain":
        state = synthetic.make_state_dict(spec, seed=0)
    elif kind == "tiny_weights":
        # every GEMM weight and bias of the encoder layers x 2^-10: each layer contributes ~nothing, but the planes hold
        # values around 3e-5 -- below the fp16 normal range
        state = synthetic.make_state_dict(spec, seed=0)
        for frag in ("attention.q_proj", "attention.k_proj", "attention.v_proj", "attention.out_proj", "feed_forward.intermediate_dense",
                     "feed_forward.output_dense"):
            R._scale(state, frag, -10)
    else:
        state = R._variant(spec, 0, kind)
    ref, ref_len = O.predict(audio, lengths, state, spec, tfi, synthetic.category_offsets(spec))
    row = []
    for precision in ("f16x3", "bf16x3", "f16"):
        est = Estimator(spec, state, "cuda:0", precision)
        pred = est.predict(batch, tfi)
        row.append(f"{precision} {R._worst(pred, ref, ref_len):.2e}")
        est.close()
    print(f"RANGE [{tag}] {kind:13s} " + "   ".join(row), flush=True)
