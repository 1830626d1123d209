"""Developer diagnostic (see diag_long.py): does a preceding solo run change the result of the 24 x 60 s batch?"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from allophant_amd import synthetic
from allophant_amd.estimator import Batch, Estimator

mode = sys.argv[1] if len(sys.argv) > 1 else "driver"
spec = bench.build_spec()
state = synthetic.make_state_dict(spec, seed=0)
tfi = synthetic.make_inventory(spec, 27, seed=0)
L = 960000
n_batch = 24
audio, lengths = synthetic.make_audio(n_batch, L, seed=17, ragged=True)
i = n_batch - 1
n_i = int(lengths[i])
if mode == "oracle":
    from oracle import allophant_oracle as O
    ref, ref_len = O.predict(audio[i:i + 1, :n_i].contiguous(), lengths[i:i + 1], state, spec, tfi, synthetic.category_offsets(spec))
    torch.save({k: v[:, 0] for k, v in ref.items()}, "/tmp/diag_ref.pt")
    sys.exit(0)
if mode == "driver":
    subprocess.check_call([sys.executable, __file__, "oracle"])
    for variant in ("batch_only", "solo_first", "last4_only", "solo_first_last4"):
        subprocess.check_call([sys.executable, __file__, variant])
    sys.exit(0)
ref = torch.load("/tmp/diag_ref.pt")
est = Estimator(spec, state, "cuda:0", "f16x3")
t_i = ref["phoneme"].shape[0]
if mode.startswith("solo_first"):
    est.predict(Batch(audio[i:i + 1, :n_i].contiguous().cuda(), lengths[i:i + 1], torch.zeros(1, dtype=torch.long)), tfi)
if mode.endswith("last4") or mode == "last4_only":
    # the second slice of the chunked run, called directly
    from allophant_amd import lib as L_
    import ctypes as C
    sub = audio[20:24].contiguous().cuda()
    sub_len = lengths[20:24].contiguous()
    est._set_inventory(tfi)
    n_out, T, total = C.c_int(), C.c_int64(), C.c_int64()
    est._lib.amx_output_layout(est._handle, 4, L, None, C.byref(n_out), C.byref(T), C.byref(total))
    descs = (L_.AmxOutputDesc * n_out.value)()
    est._lib.amx_output_layout(est._handle, 4, L, descs, C.byref(n_out), C.byref(T), C.byref(total))
    flat = torch.empty(total.value, dtype=torch.float32, device="cuda")
    out_len = torch.empty(4, dtype=torch.int64)
    code = est._lib.amx_forward(est._handle, C.c_void_p(sub.data_ptr()), C.cast(sub_len.data_ptr(), C.POINTER(C.c_int64)), 4, L,
                                C.c_void_p(flat.data_ptr()), C.cast(out_len.data_ptr(), C.POINTER(C.c_int64)), L_.FLAG_PADDED, None)
    assert code == 0, est._lib.amx_last_error(est._handle)
    torch.cuda.synchronize()
    outs = {d.name.decode(): flat[d.offset: d.offset + T.value * 4 * d.classes].view(T.value, 4, d.classes) for d in descs}
    col = 3
else:
    full = est.predict(Batch(audio.cuda(), lengths, torch.zeros(n_batch, dtype=torch.long)), tfi)
    outs, col = full.outputs, i
errs = {k: (outs[k][:t_i, col].cpu() - ref[k][:t_i]).abs().max().item() for k in ref}
worst = sorted(errs.items(), key=lambda kv: -kv[1])[:3]
diff = (outs["phoneme"][:t_i, col].cpu() - ref["phoneme"][:t_i]).abs().max(-1).values
top = torch.topk(diff, 4)
print(f"{mode:18s} vs oracle:", " ".join(f"{k}={v:.2e}" for k, v in worst), "| worst frames", top.indices.tolist())
est.close()
