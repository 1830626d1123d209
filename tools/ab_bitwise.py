"""Developer check: SHA-256 of the flat output of predict() per batch geometry, for comparing two builds of the library bit for bit
(a change that claims "same values" -- a store pattern, a cheaper instruction sequence for the same arithmetic):

    AMX_LIB_PATH=$PWD/build/ab/head.so python tools/ab_bitwise.py f16x3 32:10 8:60 > a.txt
    AMX_LIB_PATH=$PWD/build/liballophant_amx_dev.so python tools/ab_bitwise.py f16x3 32:10 8:60 > b.txt; diff a.txt b.txt

Ragged variants (packed rows, the masked last key tile) are hashed as well.  AB_ENCODER selects the encoder (bench.py --encoder)."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from allophant_amd import synthetic
from allophant_amd.estimator import Batch, Estimator

args = sys.argv[1:]
prec = args[0] if args else "f16x3"
geometries = [tuple(a.split(":")) for a in args[1:]] or [("4", "10"), ("32", "10"), ("8", "60")]
spec = bench.build_spec(encoder_name=os.environ.get("AB_ENCODER", "xlsr"))
est = Estimator(spec, synthetic.make_state_dict(spec, seed=0), torch.device("cuda", 0), prec)
tfi = synthetic.make_inventory(spec, 27, seed=0)
for n, seconds in geometries:
    n, length = int(n), int(float(seconds) * 16000)
    for ragged in (False, True):
        audio, lengths = synthetic.make_audio(n, length, seed=1234, ragged=ragged)
        pred = est.predict(Batch(audio.cuda(), lengths, torch.zeros(n, dtype=torch.long)), tfi, True, _no_graph=True)
        torch.cuda.synchronize()
        digest = hashlib.sha256(pred._flat.cpu().numpy().tobytes()).hexdigest()
        print(f"{prec} {n} x {seconds} s {'ragged' if ragged else 'padded'}: {digest}")
est.close()
