#!/usr/bin/env python3
"""Benchmark of the Allophant acoustic-encoder forward path (``Estimator.predict``) on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): encoder output frames/s over the whole node.  A *step* is one ``predict`` call over one batch of
synthetic 16 kHz audio already resident in HBM: input normalisation, the 7-layer conv feature extractor, the 24-layer
transformer encoder, all 36 attribute heads + the composed phoneme head, per-head log-softmax, and (N > 1) the RCCL gather
of the log-probabilities to rank 0.  Workload at every N: BASELINE config 2 per GPU (multitask checkpoint schema,
32 x 10 s utterances, 27-phone synthetic inventory standing in for 'es'), i.e. weak scaling: utterances are sharded
across ranks with no data-path collective other than the final gather.  Weights are procedural (seed 0): real checkpoints
are not reachable offline.

Rank 0 prints ONE JSON line (see DESIGN.md section "Measurement" for the definition of every field).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

# dmabuf IPC for RCCL / cross-process device memory on this driver; must be in the environment before HIP initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from allophant_amd import spec as S, synthetic  # noqa: E402

MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/f16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
HBM_PEAK_GBS = 8000.0


def build_spec():
    spec = S.multitask_spec(S.xlsr_300m_encoder(), allophone_layer=True)
    spec["shared_phones"] = 80
    return spec


def gemm_flops(spec, n, length, planes):
    """Algorithmic FLOPs per step of the launches of each GEMM kernel (SURVEY.md Appendix D formulas).  The routing rule
    mirrors ``pp_eligible`` in allophant_amd/csrc/amx_gemm.hip: the 256 x 256 ping-pong kernel takes every product with
    N >= 256, N % 4 == 0, M >= 384 and K a multiple of 128 / planes; the generic tile kernel takes the rest."""
    C, D, F = spec["conv_dim"], spec["hidden"], spec["ffn"]
    ts = [length]
    for k, s in zip(spec["conv_kernel"], spec["conv_stride"]):
        ts.append((ts[-1] - k) // s + 1)
    T = ts[-1]
    M = n * T
    products = []  # (M, N, K, launches)
    ln = ln_launches = 0
    last_conv = len(spec["conv_kernel"]) - 1
    for i in range(1, len(spec["conv_kernel"])):
        m, k = n * ts[i + 1], C * spec["conv_kernel"][i]
        if i < last_conv and C == 512 and m >= 1024 and k % (128 // planes) == 0:
            # `ln_eligible`: the row-complete 128 x 512 kernel with fused LayerNorm + GELU (conv layers before the last)
            ln += 2 * m * C * k
            ln_launches += 1
        else:
            products.append((m, C, k, 1))
    products.append((M, D, C, 1))
    for shape in ((3 * D, D), (D, D), (F, D), (D, F)):
        products.append((M, shape[0], shape[1], spec["layers"]))
    attr_cols = sum(c["size"] + 1 for c in spec["classes"] if c["name"] != "phoneme")
    products.append((M, attr_cols, D, 1))
    products.append((M, spec["embedding_size"], D, 1))
    pp = tile = 0
    pp_launches = 0
    pp_bytes = 0  # algorithmic HBM bytes of the ping-pong launches: operands once (16-bit planes), outputs once
    b16 = 2 * planes
    for idx, (m, nn, k, cnt) in enumerate(products):
        fl = 2 * m * nn * k * cnt
        if nn >= 256 and nn % 4 == 0 and m >= 384 and k % (128 // planes) == 0:
            pp += fl
            pp_launches += cnt
            a_bytes = m * k * b16
            if nn == C and k > C:  # conv layer: overlapping windows over n * T_in channels-last rows, read once
                a_bytes = n * ts[last_conv] * C * b16
            if (nn, k) in ((D, D), (D, F)) and cnt > 1:  # out-proj / FFN2: fp32 residual in, fp32 out
                out_bytes, extra_in = m * nn * 4, m * nn * 4
            elif (nn, k) == (D, C) or nn == C:  # feature projection / conv: fp32 out
                out_bytes, extra_in = m * nn * 4, 0
            else:  # QKV, FFN1, phoneme head: 16-bit planes out
                out_bytes, extra_in = m * nn * b16, 0
            pp_bytes += cnt * (a_bytes + nn * k * b16 + extra_in + out_bytes)
        else:
            tile += fl
    tile += 2 * M * D * (D // spec["pos_groups"]) * spec["pos_kernel"]  # grouped positional conv
    attention = spec["layers"] * 4 * M * T * D
    total = pp + ln + tile + attention + 2 * n * ts[1] * C * spec["conv_kernel"][0]
    return {"gemm_pp": pp, "gemm_pp_launches": pp_launches, "gemm_pp_bytes": pp_bytes, "gemm_ln": ln, "gemm_ln_launches": ln_launches, "gemm_tile": tile,
            "attention": attention, "total": total, "frames_per_utt": T}


def cpu_baseline(spec, state, tfi, n_sample, length):
    """Times the CPU oracle (oracle/allophant_oracle.py: the restatement pinned against the reference) on the host
    cores of this box on a bounded sample of the same workload.  Reported baseline, not the target."""
    from oracle import allophant_oracle as O

    cores = max(1, (os.cpu_count() or 2) // 2)
    torch.set_num_threads(cores)
    audio, lengths = synthetic.make_audio(n_sample, length, seed=1234)
    offsets = synthetic.category_offsets(spec)
    times = []
    frames = 0
    # warm-up on two utterances (thread pool, allocator), then two timed runs of the sample: ~30 s of CPU work in all
    O.predict(audio[:2], lengths[:2], state, spec, tfi, offsets, True)
    for _ in range(2):
        t0 = time.perf_counter()
        out, flen = O.predict(audio, lengths, state, spec, tfi, offsets, True)
        times.append(time.perf_counter() - t0)
        frames = int(flen.sum())
    med = min(times)
    return {"value": frames / med, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_sample} x {length / 16000:.0f} s utterances of the same synthetic workload, fp32 torch CPU oracle, "
                      f"warm-up + best of {len(times)} runs ({med:.2f} s)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "bf16x3", "f16", "bf16"])
    ap.add_argument("--utterances", type=int, default=32)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--phones", type=int, default=27)
    ap.add_argument("--also", default="bf16", choices=["", "f16x3", "bf16x3", "f16", "bf16"],
                    help="second precision mode reported under throughput_mode (N=1 only; empty string to skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist

    # AMX_BENCH_FORCE_DIST=1 (developer switch): run the RCCL gather path with a one-rank group, so that the collective
    # code is exercised on a single-GPU box (launch under torch.distributed.run --nproc-per-node 1)
    use_dist = world > 1 or os.environ.get("AMX_BENCH_FORCE_DIST") == "1"
    if use_dist:
        dist.init_process_group("nccl", device_id=device)

    from allophant_amd import parallel
    from allophant_amd.estimator import Batch, Estimator

    spec = build_spec()
    state = synthetic.make_state_dict(spec, seed=0)
    tfi = synthetic.make_inventory(spec, args.phones, seed=0)
    length = int(args.seconds * 16000)
    n = args.utterances
    # every rank gets its own block of a notional global batch of n * world utterances
    audio, lengths = synthetic.make_audio(n, length, seed=1234 + rank)

    def measure(precision, steps, warmup):
        """K timed steps (no per-kernel events: recording ~370 events costs 0.3-1.2 ms per step) bracketed by barrier +
        synchronize, then a second pass of K steps with HIP events around every launch for the per-kernel numbers."""
        est = Estimator(spec, state, device, precision)
        batch = Batch(audio.to(device), lengths, torch.zeros(n, dtype=torch.long))

        pending = [None]

        def step(timing=False):
            pred = est.predict(batch, tfi, True, _timing=timing)
            if use_dist:
                # RCCL gather of the per-frame log-probabilities (one flat fp32 block per rank) + frame lengths to rank 0,
                # which re-assembles `Predictions` of the global batch: [T, n * world, C] per output.  The gather of step k
                # is asynchronous and overlaps the forward pass of step k + 1; it is completed (and assembled on rank 0)
                # before step k + 2 is enqueued, and `drain()` completes the last one inside the timed region.
                previous, pending[0] = pending[0], parallel.gather_flat_predictions(pred, device, dst=0, async_op=True)
                if previous is not None:
                    gathered = previous.wait()
                    if rank == 0:
                        return gathered.outputs, gathered.lengths
            return pred.outputs, pred.lengths

        def drain():
            if pending[0] is not None:
                pending[0].wait()
                pending[0] = None

        for _ in range(warmup):
            step()
        drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        drain()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        t_tensor = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if use_dist:
            dist.all_reduce(t_tensor, op=dist.ReduceOp.MAX)
        elapsed = float(t_tensor.item())
        # instrumented pass: same steps, HIP events on the launch stream around every kernel
        est.timing_fetch()
        for _ in range(steps):
            step(timing=True)
        drain()
        torch.cuda.synchronize()
        timing = est.timing_fetch()
        est.close()
        return elapsed, timing

    def summarize(precision, steps, elapsed, timing):
        planes = 2 if precision.endswith("x3") else 1
        fl = gemm_flops(spec, n, length, planes)
        frames_per_rank = fl["frames_per_utt"] * n
        gemm_ms, gemm_launches = timing["gemm_pp"]
        achieved = fl["gemm_pp"] * steps / (gemm_ms * 1e-3) / 1e12 if gemm_ms > 0 else None
        return fl, frames_per_rank, {
            "kernel": "gemm_pp_kernel<T16, planes, MI>: persistent ping-pong GEMM, 256x256 tiles (MI = 8: feature projection, "
                      "QKV / out-proj / FFN of the 24 encoder layers, phoneme head) or 128x256 (MI = 4: last conv layer)",
            "bound": "mfma",
            "achieved": achieved,
            "peak": MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / MFMA_PEAK_TFLOPS if achieved else None,
            "traffic": load_traffic(precision),
            "flops_per_launch": fl["gemm_pp"] / fl["gemm_pp_launches"],
            "algorithmic_bytes_per_launch": fl["gemm_pp_bytes"] / fl["gemm_pp_launches"],
            "avg_launch_ms": gemm_ms / gemm_launches if gemm_launches else None,
            "launches_per_step": gemm_launches // max(1, steps),
            "mfma_issue_factor": 3 if planes == 2 else 1,
            "issued_frac": (3 if planes == 2 else 1) * achieved / MFMA_PEAK_TFLOPS if achieved else None,
            "conv_ln_gemm": {
                "kernel": "gemm_ln_kernel<T16, planes>: 128x512 row-complete GEMM + LayerNorm + GELU (conv layers 1-5)",
                "achieved": fl["gemm_ln"] * steps / (timing["gemm_ln"][0] * 1e-3) / 1e12 if timing["gemm_ln"][0] > 0 else None,
                "avg_launch_ms": timing["gemm_ln"][0] / timing["gemm_ln"][1] if timing["gemm_ln"][1] else None,
            },
            "timing": "HIP events around every launch in a second pass of the same K steps (recording them costs 0.3-1.2 ms per "
                      "step, so the timed region that yields `value` runs without them)",
        }

    def load_traffic(precision):
        """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/, collected with
        tools/profile_bench.sh as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate passes)."""
        pmc_path = os.path.join(ROOT, "profiles", "r01_gemm_traffic.json")
        try:
            with open(pmc_path) as f:
                return json.load(f)[precision]["hbm_bytes_per_launch"]
        except Exception:
            return None

    elapsed, timing = measure(args.precision, args.steps, args.warmup)
    fl, frames_per_rank, roofline = summarize(args.precision, args.steps, elapsed, timing)
    total_frames = frames_per_rank * world * args.steps
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        kernel_breakdown = {k: {"ms_per_step": round(v[0] / args.steps, 4), "launches_per_step": v[1] // max(1, args.steps)}
                            for k, v in timing.items()}
        result = {
            "metric": "encoder frames/sec (whole node), 10s x 32 utterances @16kHz",
            "value": total_frames / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {
                "workload": f"BASELINE config 2: multitask checkpoint schema (36 attribute heads + composed phoneme head, "
                            f"allophone pass-through), {n} x {args.seconds:.0f} s synthetic 16 kHz utterances per GPU, "
                            f"{args.phones}-phone synthetic inventory ('es'-sized), procedural weights seed 0",
                "global_batch": n * world,
                "frames_per_step": frames_per_rank * world,
                "parallelism": f"dp{world} (utterance shards + RCCL gather of log-probs to rank 0)" if world > 1 else "single GPU",
                "precision_mode": args.precision,
            },
            "roofline": roofline,
            "kernels": kernel_breakdown,
            "whole_step_tflops": fl["total"] * args.steps * world / elapsed / 1e12,
        }
    # the single-plane 16-bit throughput mode of the same workload (error measured and bounded in tests/, not a parity
    # mode): reported beside the parity-mode headline, never as `value`
    if args.also and args.also != args.precision and world == 1:
        e2, t2 = measure(args.also, args.steps, args.warmup)
        fl2, fpr2, roof2 = summarize(args.also, args.steps, e2, t2)
        if rank == 0:
            result["throughput_mode"] = {
                "dtype": args.also, "value": fpr2 * args.steps / e2, "unit": "frames/s", "ms_per_step": e2 / args.steps * 1e3,
                "roofline": roof2,
                "kernels": {k: {"ms_per_step": round(v[0] / args.steps, 4)} for k, v in t2.items()},
                "note": "single 16-bit plane per operand (1 MFMA per product); max-abs log-prob error vs the reference "
                        "2.3e-1 (bf16) / 3.0e-2 (f16) at XLS-R shape, see DESIGN.md section 3",
            }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(spec, state, tfi, args.cpu_sample, length)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
