"""Developer tool: run `steps` predict() steps of one batch geometry and nothing else, for rocprofv3:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $ROOT/tools/step_trace.py f16x3 4 10 20

prints the wall time per step; the per-kernel durations come from the profiler's kernel_stats.csv (divide the call counts by
steps + 3 warm-up steps).  Optional 5th argument: "base" = the wav2vec2-base (group-norm / post-LN) variant, "xlsr-1b" / "xlsr-2b" = the wide XLS-R shapes."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from allophant_amd import spec as S, synthetic
from allophant_amd.estimator import Batch, Estimator

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
seconds = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
variant = sys.argv[5] if len(sys.argv) > 5 else "xlsr"
if variant == "base":
    spec = S.multitask_spec(S.wav2vec2_base_encoder(), allophone_layer=True)
    spec["shared_phones"] = 80
elif variant in ("xlsr-1b", "xlsr-2b"):
    spec = bench.build_spec(encoder_name=variant)
else:
    spec = bench.build_spec()
state = synthetic.make_state_dict(spec, seed=0)
est = Estimator(spec, state, torch.device("cuda", 0), prec)
tfi = synthetic.make_inventory(spec, 27, seed=0)
audio, lengths = synthetic.make_audio(n, int(seconds * 16000), seed=1234)
batch = Batch(audio.cuda(), lengths, torch.zeros(n, dtype=torch.long))
for _ in range(3):
    pred = est.predict(batch, tfi, True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    pred = est.predict(batch, tfi, True)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / steps
print(f"{variant} {prec} {n} x {seconds:g} s: {wall * 1e3:.3f} ms/step over {steps} steps (+3 warm-up), {int(pred.lengths.sum()) / wall:.0f} frames/s")
est.close()
