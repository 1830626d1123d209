"""Row f3 measurement on the GPU box: throughput on a synthetic ragged "corpus" (utterance lengths ~ U[2 s, 15 s]) fed
through max-frames batching -- corpus order (what the reference harness does) vs length-sorted order -- with the next
batch staged in pinned memory and copied on a side stream while the current one is computed (batching.Prefetcher).
Reports valid (unpadded) output frames per second, the number that matters on real data."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from allophant_amd import batching as B, spec as S, synthetic
from allophant_amd.estimator import Estimator
import bench

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
n_utt = int(sys.argv[2]) if len(sys.argv) > 2 else 512
spec = bench.build_spec()
state = synthetic.make_state_dict(spec, seed=0)
device = torch.device("cuda", 0)
est = Estimator(spec, state, device, prec)
tfi = synthetic.make_inventory(spec, 27, seed=0)
g = torch.Generator().manual_seed(5)
lengths = torch.randint(2 * 16000, 15 * 16000, (n_utt,), generator=g).tolist()
audio = [torch.randn(l, generator=g) * 0.1 for l in lengths]
budget = 32 * 160000  # the padded size of BASELINE config 2


def run(order, label):
    batches = [b for b in B.max_frame_batches(order, lengths, budget) if b]
    collator = B.PinnedCollator(budget)
    fetch = lambda idx: collator([audio[i] for i in idx])
    for warm in (True, False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frames = 0
        # the first pass over the corpus also warms the caching allocators for every batch shape
        for batch in B.Prefetcher(batches, device, fetch):
            pred = est.predict(batch, tfi)
            frames += int(pred.lengths.sum())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{prec} {label}: {len(batches)} batches, padding efficiency {B.padding_efficiency(batches, lengths):.3f}, "
          f"{frames / dt:.0f} valid frames/s ({dt:.2f} s for {sum(lengths) / 16000:.0f} s of audio)", flush=True)


run(range(n_utt), "corpus order   ")
run(B.length_sorted_order(lengths), "length-sorted  ")
