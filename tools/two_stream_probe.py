"""Developer probe: one 32 x 10 s batch on one stream against two 16 x 10 s half batches on two streams (two handles), to see
whether independent launches de-phase the HBM-bound phases (epilogue bursts, row kernels) from the MFMA-bound main loops.

    python tools/two_stream_probe.py [precision] [mask pattern: none | lo-hi | i8 | i16 | even]

With a mask pattern the two streams are CU-masked (hipExtStreamCreateWithCUMask) so that each owns half of the CUs and the
kernels of the two half batches really run side by side (without masks two persistent 128-KiB-LDS workgroups cannot share a
CU, so the launches serialise); AMX_FORCE_CUS=128 then sizes the persistent grids for half a chip.  The un-split reference
run is measured in a separate process (AMX_FORCE_CUS unset)."""
import ctypes as C
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
pattern = sys.argv[2] if len(sys.argv) > 2 else "driver"
if pattern == "driver":
    for pat, cus in (("full", ""), ("none", ""), ("lo-hi", "128"), ("i8", "128"), ("i16", "128"), ("even", "128")):
        env = dict(os.environ)
        if cus:
            env["AMX_FORCE_CUS"] = cus
        subprocess.call([sys.executable, __file__, prec, pat], env=env)
    sys.exit(0)

import torch

import bench
from allophant_amd import synthetic
from allophant_amd.estimator import Batch, Estimator

spec = bench.build_spec()
state = synthetic.make_state_dict(spec, seed=0)
tfi = synthetic.make_inventory(spec, 27, seed=0)
dev = torch.device("cuda", 0)
torch.cuda.init()
TOTAL = int(os.environ.get("PROBE_UTTERANCES", "32"))  # utterances of the whole batch (even)
HALF = TOTAL // 2
audio, lengths = synthetic.make_audio(TOTAL, 160000, seed=1234)


def timed(fn, name):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    print(f"{prec} {name}: {(time.perf_counter() - t0) / 10 * 1e3:.3f} ms per {TOTAL} utterances", flush=True)


if pattern == "full":
    est = Estimator(spec, state, dev, prec)
    full = Batch(audio.cuda(), lengths, torch.zeros(TOTAL, dtype=torch.long))
    timed(lambda: est.predict(full, tfi, True), f"one stream, {TOTAL} x 10 s")
    sys.exit(0)


def masked_streams(pat):
    if pat == "none":
        return [torch.cuda.Stream(dev) for _ in range(2)]
    hip = C.CDLL("libamdhip64.so")
    masks = []
    for half in range(2):
        words = [0] * 8
        for i in range(256):
            sel = {"lo-hi": i // 128, "i8": (i // 8) % 2, "i16": (i // 16) % 2, "even": i % 2}[pat]
            if sel == half:
                words[i // 32] |= 1 << (i % 32)
        masks.append(words)
    streams = []
    for words in masks:
        handle = C.c_void_p()
        arr = (C.c_uint32 * 8)(*words)
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(handle), 8, arr)
        if rc != 0:
            raise SystemExit(f"hipExtStreamCreateWithCUMask failed: {rc}")
        streams.append(torch.cuda.ExternalStream(handle.value, device=dev))
    return streams


halves = [Batch(audio[i * HALF:(i + 1) * HALF].cuda(), lengths[i * HALF:(i + 1) * HALF], torch.zeros(HALF, dtype=torch.long)) for i in range(2)]
ests = [Estimator(spec, state, dev, prec) for _ in range(2)]
streams = masked_streams(pattern)


def run_halves():
    for est, stream, half in zip(ests, streams, halves):
        with torch.cuda.stream(stream):
            est.predict(half, tfi, True)


timed(run_halves, f"two streams ({pattern}, AMX_FORCE_CUS={os.environ.get('AMX_FORCE_CUS', '-')}), 2 x ({HALF} x 10 s)")
# one half alone on its masked stream: what a half chip does with a half batch
def run_one():
    with torch.cuda.stream(streams[0]):
        ests[0].predict(halves[0], tfi, True)
timed(run_one, f"   one masked stream alone, {HALF} x 10 s")
