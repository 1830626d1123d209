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


def run_bucketed(order, label, bucket):
    """The same corpus on a grid of batch geometries (batching.bucketed_frame_batches): batches of one bucket share (N, L), so the
    passes of a run of equal geometry replay ONE recorded HIP graph (ABI 6: recordings are keyed on geometry where no length
    travels by value) -- reported: the share of passes that were replays."""
    batches = [b for b in B.bucketed_frame_batches(order, lengths, budget, bucket)]
    collator = B.PinnedCollator(budget)
    fetch = lambda item: collator([audio[i] for i in item[0]], padded_length=item[1])
    out_ring, out_turn = {}, {}
    for warm in (True, False):
        torch.cuda.synchronize()
        c0, r0 = est.graph_info()
        t0 = time.perf_counter()
        frames, modes = 0, {0: 0, 1: 0, 2: 0}
        # buffers of a geometry are reused (two audio and two output blocks per shape, in turn): a recording holds their addresses
        for k, batch in enumerate(B.Prefetcher(batches, device, fetch, ring=2)):
            shape = tuple(batch.audio_features.shape)
            if shape not in out_ring:
                pred = est.predict(batch, tfi)  # first batch of this geometry: learn the size of its output block
                out_ring[shape] = [torch.empty(pred._flat.numel(), dtype=torch.float32, device=device) for _ in range(2)]
            else:
                turn = out_turn.get(shape, 0)
                out_turn[shape] = turn ^ 1
                pred = est.predict(batch, tfi, _out=out_ring[shape][turn])
            modes[est.pass_info()["graph"]] += 1
            frames += int(pred.lengths.sum())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        c1, r1 = est.graph_info()
    padded = sum(len(b) * l for b, l in batches)
    geometries = len({(len(b), l) for b, l in batches})
    print(f"{prec} {label}: {len(batches)} batches of {geometries} geometries (bucket {bucket / 16000:.2f} s), padding efficiency "
          f"{sum(lengths) / padded:.3f}, {frames / dt:.0f} valid frames/s ({dt:.2f} s); passes of the timed sweep: {modes[2]} replayed, "
          f"{modes[1]} recorded, {modes[0]} eager = {modes[2] / max(1, len(batches)):.0%} replays", flush=True)


run(range(n_utt), "corpus order   ")
run(B.length_sorted_order(lengths), "length-sorted  ")
run_bucketed(B.length_sorted_order(lengths), "length-sorted, bucketed", 8000)
run_bucketed(B.length_sorted_order(lengths), "length-sorted, bucketed", 16000)
