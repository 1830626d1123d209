#!/usr/bin/env python3
"""Benchmark of the Allophant acoustic-encoder forward path (``Estimator.predict``) on MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N --steps K --warmup W          (N > 1 without WORLD_SIZE: starts its own N ranks, see `self_launch`)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Metric (BASELINE.json): encoder output frames/s over the whole node.  A *step* is one ``predict`` over one batch of
synthetic 16 kHz audio already resident in HBM: input normalisation, the 7-layer conv feature extractor, the 24-layer
transformer encoder, all 36 attribute heads + the composed phoneme head, per-head log-softmax.

* N = 1: BASELINE config 2 -- 32 x 10 s utterances on one GPU (multitask checkpoint schema, 27-phone synthetic inventory
  standing in for 'es').
* N > 1: BASELINE config 3 -- the SAME global batch of 32 x 10 s utterances sharded 32 / N per rank
  (``parallel.shard_batch``: contiguous utterance blocks, no data-path collective) with the RCCL gather of the per-frame
  log-probabilities to rank 0 inside the timed region: strong scaling (``"scaling": "strong"``).  The weak-scaling figure
  (32 x 10 s per GPU, gathered the same way) is reported beside it under ``weak_scaling``.

Weights are procedural (seed 0): real checkpoints are not reachable offline.  Rank 0 prints ONE JSON line (DESIGN.md
section "Measurement" defines every field).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import math
import os
import socket
import statistics
import subprocess
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
MEASURED_MFMA_CEILING_TFLOPS = 2100.0  # 16x16x32 f16, two waves per SIMD (profiles/r02_mfma_only_ceiling.log)
CONV0_PATTERN_STORE_GBS = 5660.0  # conv0's own store pattern without its arithmetic (tools/store_bw_probe.hip, profiles/r03_store_bw.log)
TRAFFIC_FILES = ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r04b_traffic.json", "r03il_traffic.json", "r03_traffic.json", "r02_traffic.json")  # newest first; written by tools/collect_profiles.py


# BASELINE.json configs this file can time on one GPU (config 3 = config 2 under --gpus N; config 1 is CPU plumbing):
# (utterances, seconds, phones of the synthetic inventory, hierarchical graph)
CONFIG_PRESETS = {
    1: (1, 3.0, 64, "baseline"),  # the reference's CPU-runnable case on the device: baseline schema (one phoneme classifier, no composition)
    2: (32, 10.0, 27, False),   # multitask, 'es'-sized inventory -- the configuration `metric` is quoted on (default)
    4: (64, 5.0, 48, True),     # hierarchical: the phoneme head sees cat(OUTPUT, softmax(attribute logits)), ['es','it']-sized
    5: (8, 60.0, 200, False),   # long-form stress: 200-phone inventory (its stated fp16 single-plane mode: --also f16)
}


def build_encoder(name):
    """`xlsr` (default: every released Allophant checkpoint), or the group-norm / post-LN family the reference also accepts as
    `model_id` (acoustic_model.py:775-826): `w2v2-base` (768 / 12 / 12 / 3072), `w2v2-large` (1024 / 24 / 16 / 4096)."""
    if name == "w2v2-base":
        return S.wav2vec2_base_encoder()
    if name in ("xlsr-1b", "xlsr-2b"):  # hidden 1280 / 1920, 48 layers, heads of 80 / 120 (round 6)
        return S.xlsr_1b_encoder() if name == "xlsr-1b" else S.xlsr_2b_encoder()
    encoder = S.xlsr_300m_encoder()
    if name == "w2v2-large":
        encoder.update(feat_extract_norm="group", conv_bias=False, stable_layer_norm=False, use_attention_mask=False)
    return encoder


def build_spec(hierarchical=False, encoder_name="xlsr", phones=64):
    encoder = build_encoder(encoder_name)
    if hierarchical == "baseline":  # BASELINE config 1: `kgnlp/allophant-baseline` schema -- a single phoneme head Linear(hidden -> P + 1)
        return S.baseline_spec(encoder, phonemes=phones)
    spec = S.hierarchical_spec(encoder, allophone_layer=True) if hierarchical else S.multitask_spec(encoder, allophone_layer=True)
    spec["shared_phones"] = 80
    return spec


def work_model(spec, n, length, planes):
    """Algorithmic FLOPs and HBM bytes per step of each kernel class (SURVEY.md Appendix D formulas).  The routing rule
    mirrors ``pp_eligible`` / ``ln_eligible`` in allophant_amd/csrc/amx_gemm.hip: conv layers 1-5 run on the row-complete
    kernel with fused LayerNorm + GELU (``gemm_ln``), the last conv layer is timed as ``conv_tail``, the ping-pong kernel
    takes every other product with N >= 256, N % 4 == 0, M >= 384 and K a multiple of 128 / planes, the tile kernels the
    rest.  Bytes: every operand read once and every output written once; 16-bit planes x `planes`, fp32 where the path
    keeps fp32 (audio, residual stream)."""
    C, D, F = spec["conv_dim"], spec["hidden"], spec["ffn"]
    ts = [length]
    for k, s in zip(spec["conv_kernel"], spec["conv_stride"]):
        ts.append((ts[-1] - k) // s + 1)
    T = ts[-1]
    M = n * T
    b16 = 2 * planes
    n_conv = len(spec["conv_kernel"])
    last_conv = n_conv - 1
    ln_flops = ln_bytes = 0
    ln_launches = 0
    tail_flops = tail_bytes = 0
    products = []  # (M, N, K, launches)
    for i in range(1, n_conv):
        m, k = n * ts[i + 1], C * spec["conv_kernel"][i]
        flops = 2 * m * C * k
        io = n * ts[i] * C * b16 + C * k * b16 + m * C * b16  # input planes once (overlapping windows), weights, output planes
        if i == last_conv:
            tail_flops, tail_bytes = flops, io
        elif C == 512 and m >= 1024 and k % (128 // planes) == 0 and spec.get("feat_extract_norm", "layer") == "layer":
            ln_flops += flops
            ln_bytes += io
            ln_launches += 1
        else:
            products.append((m, C, k, 1))
    products.append((M, D, C, 1))
    for shape in ((3 * D, D), (D, D), (F, D), (D, F)):
        products.append((M, shape[0], shape[1], spec["layers"]))
    attr_cols = sum(c["size"] + 1 for c in spec["classes"] if c["name"] != "phoneme")
    products.append((M, attr_cols, D, 1))
    phoneme = next(c for c in spec["classes"] if c["name"] == "phoneme")
    k_phoneme = synthetic.head_input_size(spec, phoneme)
    # (baseline schema: the phoneme head is a plain classifier of P + 1 columns, no embedding composition)
    products.append((M, spec["embedding_size"] or phoneme["size"] + 1, (k_phoneme + 31) // 32 * 32, 1))  # K padded to the 32-element row blocks
    pp = tile = 0
    pp_launches = 0
    pp_bytes = 0
    for (m, nn, k, cnt) in products:
        fl = 2 * m * nn * k * cnt
        if nn >= 256 and nn % 4 == 0 and m >= 384 and k % (128 // planes) == 0:
            pp += fl
            pp_launches += cnt
            a_bytes = m * k * b16
            if (nn, k) in ((D, D), (D, F)) and cnt > 1:  # out-proj / FFN2: fp32 residual in, fp32 out
                out_bytes, extra_in = m * nn * 4, m * nn * 4
            elif (nn, k) == (D, C):  # feature projection: fp32 out
                out_bytes, extra_in = m * nn * 4, 0
            else:  # QKV, FFN1, phoneme head: 16-bit planes out
                out_bytes, extra_in = m * nn * b16, 0
            pp_bytes += cnt * (a_bytes + nn * k * b16 + extra_in + out_bytes)
        else:
            tile += fl
    tile += 2 * M * D * (D // spec["pos_groups"]) * spec["pos_kernel"]  # grouped positional conv
    attention = spec["layers"] * 4 * M * T * D
    conv0_flops = 2 * n * ts[1] * C * spec["conv_kernel"][0]
    conv0_bytes = n * length * 4 + n * ts[1] * C * b16  # fp32 audio in, planes out
    total = pp + ln_flops + tail_flops + tile + attention + conv0_flops
    return {"gemm_pp": pp, "gemm_pp_launches": pp_launches, "gemm_pp_bytes": pp_bytes,
            "gemm_ln": ln_flops, "gemm_ln_bytes": ln_bytes, "gemm_ln_launches": ln_launches,
            "conv_tail": tail_flops, "conv_tail_bytes": tail_bytes, "conv0": conv0_flops, "conv0_bytes": conv0_bytes,
            "gemm_tile": tile, "attention": attention, "total": total, "frames_per_utt": T}


def physical_cores():
    """(physical cores, logical CPUs) this process may run on: distinct (physical id, core id) pairs of /proc/cpuinfo among
    the CPUs of the affinity mask (sockets x cores, SMT siblings counted once)."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    pairs, cpu, phys = set(), None, None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                key, _, val = line.partition(":")
                key, val = key.strip(), val.strip()
                if key == "processor":
                    cpu, phys = int(val), None
                elif key == "physical id":
                    phys = val
                elif key == "core id" and cpu in allowed:
                    pairs.add((phys, val))
    except OSError:
        pass
    return (len(pairs) or len(allowed)), len(allowed)


def cpu_baseline(spec, state, tfi, audio, lengths):
    """Times the CPU oracle (oracle/allophant_oracle.py: the restatement pinned against the reference) on the host cores
    of this box on the sample it is given -- by default the WHOLE benchmark batch.  The thread count
    is chosen first: a sweep on a two-utterance slice (physical cores, a half, a quarter, and 8 -- the survey's container
    figure was taken on 8 threads) ranks the counts -- fp32 GEMMs of 499-row utterances stop scaling long before a 2-socket
    box runs out of cores, and oversubscribing them is slower than a few cores -- and the two best are timed on the first 8
    utterances (more utterances = more parallel work, so the slice's winner is not always the sample's).  The winner then
    runs the whole sample ONCE, timed: that run is `value` (the 8-utterance trial doubles as its warm-up).  Reported baseline,
    not the target.  Returns (record, oracle outputs of the sample, frame lengths) -- the outputs double as the parity spot
    check of the timed path."""
    from oracle import allophant_oracle as O

    cores, logical = physical_cores()
    offsets = synthetic.category_offsets(spec) if spec.get("composition_categories") else None
    candidates = sorted({max(1, cores), max(1, cores // 2), max(1, cores // 4), min(8, max(1, cores))}, reverse=True)
    sweep = {}
    probe_a, probe_l = audio[:2].contiguous(), lengths[:2].contiguous()
    for threads in candidates:
        torch.set_num_threads(threads)
        O.predict(probe_a, probe_l, state, spec, tfi, offsets, True)  # warm-up of the pool at this size
        t0 = time.perf_counter()
        _, flen = O.predict(probe_a, probe_l, state, spec, tfi, offsets, True)
        sweep[threads] = int(flen.sum()) / (time.perf_counter() - t0)
    ranked = sorted(sweep, key=sweep.get, reverse=True)[:2]
    n_trial = min(8, len(lengths))
    trial_a, trial_l = audio[:n_trial].contiguous(), lengths[:n_trial].contiguous()
    trial = {}
    for threads in ranked:
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        _, flen = O.predict(trial_a, trial_l, state, spec, tfi, offsets, True)
        trial[threads] = int(flen.sum()) / (time.perf_counter() - t0)
    best = max(trial, key=trial.get)
    torch.set_num_threads(best)
    # `value`: the MEDIAN of three timed runs on the bounded sample (the first n_trial utterances: ~6 s per run on the box, the
    # trial above was their warm-up), with the spread -- BASELINE.md section 4: "1 warm-up + median of >= 3"
    runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        out, flen = O.predict(trial_a, trial_l, state, spec, tfi, offsets, True)
        runs.append(int(flen.sum()) / (time.perf_counter() - t0))
    ordered = sorted(runs)
    whole = None
    if len(lengths) > n_trial:
        # one run over everything it was given (the whole benchmark batch by default): its outputs are the parity spot check of
        # the timed path; its rate is reported beside the median
        t0 = time.perf_counter()
        out, flen = O.predict(audio, lengths, state, spec, tfi, offsets, True)
        seconds = time.perf_counter() - t0
        whole = {"value": int(flen.sum()) / seconds, "seconds": seconds, "utterances": len(lengths)}
    record = {"value": ordered[1], "unit": "frames/s", "cores": best, "kind": "port",
              "runs_frames_per_s": [round(v, 1) for v in runs], "min": ordered[0], "max": ordered[2],
              "spread_rel": (ordered[2] - ordered[0]) / ordered[1] if ordered[1] > 0 else None,
              "whole_batch_run": whole,
              "physical_cores": cores, "logical_cpus": logical,
              "thread_sweep_frames_per_s": {str(k): round(v, 1) for k, v in sweep.items()},
              "trial_frames_per_s": {str(k): round(v, 1) for k, v in trial.items()},
              "sample": f"median of 3 timed runs on the first {n_trial} x {audio.shape[1] / 16000:.0f} s utterances of the benchmark "
                        f"batch (one earlier run on them = warm-up), fp32 torch CPU oracle on {best} threads (a sweep over "
                        f"{candidates} on a 2-utterance slice ranks the counts, the two best are timed on the sample); "
                        f"whole_batch_run = one run over all {len(lengths)} utterances, whose outputs are the parity spot check; "
                        f"{cores} physical cores, {logical} logical CPUs"}
    return record, out, flen


# host-only translation units of the library: no device code, no launch plan (amx_api.hip orchestrates launches that the other
# files plan and implement; amx_dist.hip calls RCCL)
HOST_ONLY_SOURCES = ("amx_api.hip", "amx_dist.hip")


def kernel_source_hash():
    """Identifies the kernels a measurement belongs to: sha256 over the device sources and launch plans of liballophant_amx
    (names + bytes of csrc/*.hip, *.inc, *.h except the host-only files).  Round 5: the host-only files left the hash -- an edit of
    the range-report bookkeeping in amx_api.hip must not void PMC passes of kernels it cannot change."""
    digest = hashlib.sha256()
    csrc = os.path.join(ROOT, "allophant_amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".inc", ".h")) and name not in HOST_ONLY_SOURCES:
            digest.update(name.encode())
            with open(os.path.join(csrc, name), "rb") as f:
                digest.update(f.read())
    return digest.hexdigest()[:16]


def load_traffic(precision):
    """Measured HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/, collected with
    tools/profile_bench.sh exactly as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate passes,
    FETCH_SIZE doubled).  NOT measured in this run: the file carries the hash of the kernel sources it was measured on
    (tools/collect_profiles.py) and is used only when that equals the hash of the sources in this tree -- a kernel change
    nulls `traffic` until the PMC passes are repeated.  Returns (data or None, description of the source)."""
    current = kernel_source_hash()
    stale = None
    for name in TRAFFIC_FILES:
        path = os.path.join(ROOT, "profiles", name)
        try:
            with open(path) as f:
                blob = json.load(f)
            data = blob[precision]
        except Exception:
            continue
        measured_on = blob.get("kernel_source_hash")
        if measured_on == current:
            return data, (f"profiles/{name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on kernel sources "
                          f"{current}, FETCH_SIZE doubled per MI355X_MICROARCH.md; not a live counter)")
        stale = stale or f"profiles/{name} was measured on kernel sources {measured_on}, this tree is {current}: traffic not reported"
    return None, stale


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(argv, gpus):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: this process has not touched the GPU (no
    HIP call, no torch.cuda.is_available()), so it starts N fresh ranks -- `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` on this same file, as a CHILD process, never a re-exec
    -- relays rank 0's single JSON line on stdout and exits with the children's status.  AMX_BENCH_CHILD_SCRIPT (tests) names
    another script for the ranks to run."""
    script = os.environ.get("AMX_BENCH_CHILD_SCRIPT") or os.path.abspath(__file__)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), script] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or gpus) // gpus)))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE)
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.strip()]
    json_lines = [ln for ln in lines if ln.lstrip().startswith("{")]
    for ln in lines:
        if ln not in json_lines:
            print(ln, file=sys.stderr)
    if proc.returncode != 0:
        print(f"bench.py: a rank failed (torch.distributed.run exited with {proc.returncode})", file=sys.stderr)
        return proc.returncode or 1
    if len(json_lines) != 1:
        print(f"bench.py: expected one JSON line from rank 0, got {len(json_lines)}", file=sys.stderr)
        return 1
    print(json_lines[0], flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "bf16x3", "f16", "bf16"])
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIG_PRESETS),
                    help="BASELINE.json config to time on one GPU: 2 (default, the configuration `metric` is quoted on; config 3 "
                         "under --gpus N), 4 (hierarchical, 64 x 5 s), 5 (8 x 60 s, 200 phones) or 1 (baseline schema, 1 x 3 s: the "
                         "reference's CPU-runnable case, here on the device)")
    ap.add_argument("--encoder", default="xlsr", choices=["xlsr", "w2v2-base", "w2v2-large", "xlsr-1b", "xlsr-2b"],
                    help="wav2vec 2.0 shape and variant: xlsr (default; `metric` is quoted on it) or the group-norm / post-LN family "
                         "(informational lines: other work per frame)")
    ap.add_argument("--utterances", type=int, default=None, help="utterances of the (global) batch (default: the config's)")
    ap.add_argument("--seconds", type=float, default=None)
    ap.add_argument("--phones", type=int, default=None)
    ap.add_argument("--also", default="bf16", choices=["", "f16x3", "bf16x3", "f16", "bf16"],
                    help="second precision mode reported under throughput_mode (N=1 only; empty string to skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="utterances of the batch the CPU baseline (and the parity spot check) runs on; 0 = the whole batch "
                         "(config 2: 32 x 10 s, ~25 s per oracle run on the box)")
    ap.add_argument("--no-spot-check", action="store_true", help="skip the oracle spot check of the timed path")
    ap.add_argument("--decoded-gather", action="store_true",
                    help="N > 1: also time the step that gathers greedy CTC alignments of the phoneme output instead of log-probs")
    ap.add_argument("--no-ragged", action="store_true", help="N = 1: skip the informational ragged-batch leg")
    ap.add_argument("--no-weak", action="store_true", help="N > 1: skip the weak-scaling leg (32 x 10 s per GPU)")
    ap.add_argument("--no-graph", action="store_true",
                    help="enqueue every pass launch by launch (AMX_FLAG_NO_GRAPH) instead of replaying its HIP graph: the same kernels; "
                         "used under rocprofv3 counter collection (tools/profile_bench.sh)")
    args = ap.parse_args()
    preset = CONFIG_PRESETS[args.config]
    if args.utterances is None:
        args.utterances = preset[0]
    if args.seconds is None:
        args.seconds = preset[1]
    if args.phones is None:
        args.phones = preset[2]
    if args.config != 2 and args.gpus > 1:
        raise SystemExit("--gpus N > 1 times BASELINE config 3 (config 2 sharded); --config 4 / 5 are single-GPU lines")

    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("AMX_BENCH_FORCE_LAUNCH") == "1"):
        # nothing above touched the GPU: this process only starts the ranks and relays rank 0's line
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))
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
    # stdout carries exactly ONE JSON line: RCCL prints a version banner to stdout when its first communicator comes up, so
    # everything before the final print goes to stderr (file descriptor 1 is pointed at stderr until then)
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if use_dist:
        dist.init_process_group("nccl", device_id=device)

    from allophant_amd import parallel
    from allophant_amd.estimator import Batch, Estimator

    spec = build_spec(hierarchical=preset[3], encoder_name=args.encoder, phones=args.phones)
    state = synthetic.make_state_dict(spec, seed=0)
    tfi = synthetic.make_inventory(spec, args.phones, seed=0) if spec.get("embedding_size") else None
    length = int(args.seconds * 16000)

    graph_stats = {}
    pass_stats = {}

    def measure(precision, steps, warmup, batch, timing_pass=True):
        """K timed steps (no per-kernel events: recording ~370 events costs 0.3-1.2 ms per step) bracketed by barrier +
        synchronize, max over ranks; then a second pass of K steps with HIP events around every launch for the per-kernel
        numbers.  `batch` is this rank's device-resident shard."""
        est = Estimator(spec, state, device, precision)
        # equal shards are established on the host below (shard_bounds of equal-length utterances): no per-step agreement
        runner = parallel.DataParallelRunner(lambda b: est.predict(b, tfi, True, _no_graph=args.no_graph), device, dst=0,
                                             verify_shapes=False) if use_dist else None

        def step(timing=False):
            if runner is None:
                return est.predict(batch, tfi, True, _timing=timing, _no_graph=args.no_graph)
            if timing:
                est.predict(batch, tfi, True, _timing=True)
                return None
            # RCCL gather of the per-frame log-probabilities (one flat fp32 block per rank) + frame lengths to rank 0,
            # which re-assembles `Predictions` of the global batch: [T, n_global, C] per output
            return runner.step(batch)

        for _ in range(warmup):
            step()
        if runner is not None:
            runner.drain()
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        last = None
        if runner is not None:
            last = runner.drain()  # the last gather completes (and is assembled on rank 0) inside the timed region
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if last is not None:
            last.check_ranks()  # (outside the timed region: a rank that reported a range error with its shard fails the run here)
        t_tensor = torch.tensor([elapsed], dtype=torch.float64, device=device)
        if use_dist:
            dist.all_reduce(t_tensor, op=dist.ReduceOp.MAX)
        elapsed = float(t_tensor.item())
        graph_stats[precision] = est.graph_info()  # (passes recorded into HIP graphs, passes replayed) up to the end of the timed region
        pass_stats[precision] = est.pass_info()    # what the last timed pass did: LayerNorm fold, row layout, eager / recorded / replayed
        timing = None
        if timing_pass:
            # instrumented pass: same steps, HIP events on the launch stream around every kernel
            est.timing_fetch()
            for _ in range(steps):
                step(timing=True)
            torch.cuda.synchronize()
            timing = est.timing_fetch()
        est.close()
        return elapsed, timing

    def summarize(precision, steps, timing, n_local):
        planes = 2 if precision.endswith("x3") else 1
        issue = 3 if planes == 2 else 1
        w = work_model(spec, n_local, length, planes)
        traffic, traffic_source = load_traffic(precision)
        if args.config != 2 or args.utterances != CONFIG_PRESETS[2][0] or args.seconds != CONFIG_PRESETS[2][1] or world != 1 or args.encoder != "xlsr":
            # the committed PMC passes are those of config 2 on one GPU: another workload has other launches
            traffic, traffic_source = None, "the committed PMC passes (profiles/) are those of BASELINE config 2 on one GPU: not reported for this workload"
        traffic = traffic or {}  # {} when no PMC pass of THESE kernel sources is committed: every traffic field is null

        def rate(flops, cls):
            ms = timing[cls][0]
            return flops * steps / (ms * 1e-3) / 1e12 if ms > 0 else None

        gemm_ms, gemm_launches = timing["gemm_pp"]
        achieved = rate(w["gemm_pp"], "gemm_pp")
        conv0_ms = timing["conv0"][0] / steps
        ln_ms = timing["gemm_ln"][0] / steps
        tail_ms = timing["conv_tail"][0] / steps
        conv0_gbs = w["conv0_bytes"] / (conv0_ms * 1e-3) / 1e9 if conv0_ms > 0 else None
        ln_tf = rate(w["gemm_ln"], "gemm_ln")
        stage_ms = conv0_ms + ln_ms + tail_ms
        stage_bytes = w["conv0_bytes"] + w["gemm_ln_bytes"] + w["conv_tail_bytes"]
        stage_flops = w["conv0"] + w["gemm_ln"] + w["conv_tail"]
        roofline = {
            "kernel": "gemm_pp_kernel<T16, planes, 8>: persistent ping-pong GEMM on 256x256 tiles (128x256 when a product cannot "
                      f"fill the chip): feature projection, QKV / out-proj / FFN of the {spec['layers']} encoder layers, phoneme head",
            "bound": "mfma",
            "achieved": achieved,
            "peak": MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / MFMA_PEAK_TFLOPS if achieved else None,
            "traffic": traffic.get("hbm_bytes_per_launch"),
            "traffic_source": traffic_source,
            "flops_per_launch": w["gemm_pp"] / max(1, w["gemm_pp_launches"]),
            "algorithmic_bytes_per_launch": w["gemm_pp_bytes"] / max(1, w["gemm_pp_launches"]),
            "avg_launch_ms": gemm_ms / gemm_launches if gemm_launches else None,
            "launches_per_step": gemm_launches // max(1, steps),
            "mfma_issue_factor": issue,
            "issued_frac": issue * achieved / MFMA_PEAK_TFLOPS if achieved else None,
            # what a loop of nothing but independent MFMAs sustains on an MI355X (tools/mfma_clock_probe.hip,
            # profiles/r02_mfma_only_ceiling.log: the clock falls to ~2.0 GHz under matrix load), for scale beside `peak`
            "measured_mfma_only_ceiling": MEASURED_MFMA_CEILING_TFLOPS,
            "issued_frac_of_measured_ceiling": issue * achieved / MEASURED_MFMA_CEILING_TFLOPS if achieved else None,
            # north-star "transformer block": every kernel between the feature projection and the heads -- the ping-pong GEMMs
            # (QKV / out-proj / FFN of the 24 layers; the class also holds the feature projection and the phoneme head, 2 of
            # its 98 launches), attention, the LayerNorm rows and the tile-kernel class (positional convolution + the narrow
            # heads) -- algorithmic FLOPs over their summed HIP-event time
            "whole_block": (lambda fl, ms: {
                "flops": fl, "ms": ms, "achieved": fl / (ms * 1e-3) / 1e12 if ms > 0 else None, "peak": MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if ms > 0 else None,
                "issued_frac": issue * fl / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if ms > 0 else None,
                "classes": ["gemm_pp", "attention", "rownorm", "gemm_tile"]})(
                    w["gemm_pp"] + w["attention"] + w["gemm_tile"],
                    (timing["gemm_pp"][0] + timing["attention"][0] + timing["rownorm"][0] + timing["gemm_tile"][0]) / steps),
            # the conv feature extractor (north-star: HBM fraction of the conv stage with rocprof evidence)
            "conv_stage": {
                "conv0": {
                    "kernel": "conv0_kernel: input norm + conv k=10 s=5 (C_in = 1) + LayerNorm + GELU -> planes",
                    "bound": "hbm", "algorithmic_bytes": w["conv0_bytes"], "ms": conv0_ms, "achieved": conv0_gbs,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": conv0_gbs / HBM_PEAK_GBS if conv0_gbs else None,
                    "traffic": traffic.get("conv0_hbm_bytes_per_launch"),
                    # 99 % of this kernel's bytes are stores.  For scale only: the rate of its own store pattern with the
                    # arithmetic removed (tools/store_bw_probe.hip: 5.64-5.68 TB/s; pure store streams reach 5.3-6.6 TB/s by
                    # shape) -- the kernel is VALU-bound (0.45 ms without its stores); `frac` above,
                    # against the 8 TB/s peak, is the roofline figure
                    "own_store_probe_gbs": CONV0_PATTERN_STORE_GBS,
                    "frac_of_own_probe": conv0_gbs / CONV0_PATTERN_STORE_GBS if conv0_gbs else None,
                },
                "conv1_5": {
                    "kernel": "gemm_ln_il_kernel<T16, planes>: 128x512 row-complete implicit GEMM + LayerNorm + GELU",
                    "bound": "mfma", "flops": w["gemm_ln"], "algorithmic_bytes": w["gemm_ln_bytes"], "ms": ln_ms,
                    "achieved": ln_tf, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": ln_tf / MFMA_PEAK_TFLOPS if ln_tf else None,
                    "issued_frac": issue * ln_tf / MFMA_PEAK_TFLOPS if ln_tf else None,
                    "avg_launch_ms": timing["gemm_ln"][0] / timing["gemm_ln"][1] if timing["gemm_ln"][1] else None,
                    "traffic": traffic.get("gemm_ln_hbm_bytes_per_step"),
                },
                "conv6": {"kernel": "gemm_pp_kernel<T16, planes, 4> + rownorm (LayerNorm + GELU + feature-projection LayerNorm)",
                          "flops": w["conv_tail"], "algorithmic_bytes": w["conv_tail_bytes"], "ms": tail_ms},
                "whole_stage": {
                    "algorithmic_bytes": stage_bytes, "flops": stage_flops, "ms": stage_ms,
                    "achieved_gbs": stage_bytes / (stage_ms * 1e-3) / 1e9 if stage_ms > 0 else None,
                    "frac_hbm": stage_bytes / (stage_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if stage_ms > 0 else None,
                    "achieved_tflops": stage_flops / (stage_ms * 1e-3) / 1e12 if stage_ms > 0 else None,
                    "frac_mfma": stage_flops / (stage_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if stage_ms > 0 else None,
                    "note": "layers 1-6 are contractions at ~340-510 FLOP per byte: MFMA-bound, so the stage's HBM fraction is "
                            "low by construction; conv0 is the bandwidth-shaped kernel",
                },
            },
            "timing": "HIP events around every launch in a second pass of the same K steps (recording them costs 0.3-1.2 ms per "
                      "step, so the timed region that yields `value` runs without them)",
        }
        return w, roofline

    def kernel_table(timing, steps):
        return {k: {"ms_per_step": round(v[0] / steps, 4), "launches_per_step": v[1] // max(1, steps)} for k, v in timing.items()}

    # ---- the batch of this rank ----
    n_global = args.utterances
    audio, lengths = synthetic.make_audio(n_global, length, seed=1234)  # same on every rank
    global_batch = Batch(audio, lengths, torch.zeros(n_global, dtype=torch.long))
    if world > 1:
        # (group-norm / unmasked variants: the shards keep the global padded length -- parallel.padding_sensitive)
        shard = parallel.shard_batch(global_batch, rank, world, spec=spec)
        if shard is None:
            raise SystemExit("more ranks than utterances")
        bounds = parallel.shard_bounds(n_global, world)
        if len({hi - lo for lo, hi in bounds}) != 1:
            raise SystemExit("--utterances must be a multiple of --gpus (equal shards, one flat gather per step)")
        n_local = len(shard)
        local = Batch(shard.audio_features.to(device), shard.lengths, shard.language_ids)
        if getattr(shard, "_padded", False):
            local._padded = True
    else:
        n_local = n_global
        local = Batch(audio.to(device), lengths, global_batch.language_ids)

    elapsed, timing = measure(args.precision, args.steps, args.warmup, local)
    w, roofline = summarize(args.precision, args.steps, timing, n_local)
    frames_global = w["frames_per_utt"] * n_global
    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        config_name = f"BASELINE config {args.config}" if world == 1 else f"BASELINE config 3 (config 2 sharded {n_local} utterances per GPU x {world}, RCCL gather of log-probs to rank 0)"
        graph_name = ("baseline checkpoint schema (one phoneme classifier Linear(hidden -> P + 1), no attribute heads, no composition"
                      if preset[3] == "baseline" else
                      "hierarchical checkpoint schema (36 attribute heads; the composed phoneme head reads cat(OUTPUT, softmax of every "
                      "attribute head)" if preset[3] else "multitask checkpoint schema (36 attribute heads + composed phoneme head")
        result = {
            "metric": "encoder frames/sec (whole node), 10s x 32 utterances @16kHz" if args.config == 2 and args.encoder == "xlsr" else
                      f"encoder frames/sec (whole node), {args.seconds:g}s x {n_global} utterances @16kHz (BASELINE config {args.config}"
                      + ("" if args.encoder == "xlsr" else f", {args.encoder} encoder") + ")",
            "value": frames_global * args.steps / elapsed,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong" if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic",
            "config": {
                "workload": f"{config_name}: {graph_name}, allophone "
                            f"pass-through), global batch {n_global} x {args.seconds:.0f} s synthetic 16 kHz utterances, "
                            f"{args.phones}-phone synthetic inventory, procedural weights seed 0",
                "global_batch": n_global,
                "utterances_per_gpu": n_local,
                "frames_per_step": frames_global,
                "parallelism": f"dp{world} (contiguous utterance shards, no data-path collective; one RCCL gather of log-probs + frame "
                               f"lengths to rank 0 per step, overlapped with the next step)" if world > 1 else "single GPU",
                "precision_mode": args.precision,
                "encoder": args.encoder,
            },
            "roofline": roofline,
            "kernels": kernel_table(timing, args.steps),
            "whole_step_tflops": w["total"] * args.steps * world / elapsed / 1e12,
            # launch collapse (ABI 5): the timed steps replay ONE HIP graph of the ~185 launches of a pass
            "launch_collapse": {"graphs_recorded": graph_stats[args.precision][0], "passes_replayed": graph_stats[args.precision][1],
                                "passes_issued": args.steps + args.warmup},
            # amx_pass_info of the last timed pass: ln_fold 1 = the pre-LN layers ran without LayerNorm passes (folded into the
            # products around them: `kernels.rownorm` is then the fold's row statistics + the first / final norm), packed rows, graph
            "pass": pass_stats.get(args.precision),
        }
    # N > 1: the weak-scaling leg (config 2 on every GPU), reported beside the strong-scaling headline
    if world > 1 and not args.no_weak:
        w_audio, w_lengths = synthetic.make_audio(n_global, length, seed=1234 + rank)
        weak_batch = Batch(w_audio.to(device), w_lengths, torch.zeros(n_global, dtype=torch.long))
        e_weak, _ = measure(args.precision, args.steps, args.warmup, weak_batch, timing_pass=False)
        if rank == 0:
            result["weak_scaling"] = {
                "value": frames_global * world * args.steps / e_weak, "unit": "frames/s", "ms_per_step": e_weak / args.steps * 1e3,
                "global_batch": n_global * world, "utterances_per_gpu": n_global,
                "note": "BASELINE config 2 on every GPU (per-GPU work fixed), log-probs gathered to rank 0 the same way",
            }
    # N > 1, on request: SURVEY 8 f1 -- decode on every GPU and gather only the phoneme alignments (token ids, timesteps,
    # scores) to rank 0: synchronous per step (the padded alignment length is agreed on with an all-reduce first)
    if use_dist and args.decoded_gather:
        est = Estimator(spec, state, device, args.precision)

        def decoded_step():
            # capacity = frames of the padded length: no rank can hold a longer alignment, so no per-step agreement (and no
            # host read-back) is needed
            return parallel.gather_decoded(est.greedy_decode_device(est.predict(local, tfi, True)), ["phoneme"], n_global,
                                           device, dst=0, capacity=w["frames_per_utt"])

        for _ in range(args.warmup):
            decoded_step()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            hyps = decoded_step()
        torch.cuda.synchronize()
        dist.barrier()
        e_dec = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        dist.all_reduce(e_dec, op=dist.ReduceOp.MAX)
        est.close()
        if rank == 0:
            result["decoded_gather"] = {
                "value": frames_global * args.steps / float(e_dec.item()), "unit": "frames/s",
                "ms_per_step": float(e_dec.item()) / args.steps * 1e3, "outputs": ["phoneme"],
                "hypotheses_on_rank0": len(hyps["phoneme"]),
                "note": "greedy CTC on every GPU, one packed int32 gather of the alignments per step (no log-probs cross xGMI)",
            }
    # N = 1: the same step when the boundary hands over HOST buffers (pinned audio in, log-probabilities fetched into pinned
    # memory, synchronised every step): the PCIe-inclusive rate.  Reported beside `value`, never as `value`.
    if world == 1 and rank == 0:
        try:
            est = Estimator(spec, state, device, args.precision)
            host_audio = audio.pin_memory()
            probe = est.predict(local, tfi, True, _no_graph=args.no_graph)
            host_out = torch.empty(probe._flat.numel(), dtype=torch.float32).pin_memory()
            del probe

            def host_step():
                dev = Batch(host_audio.to(device, non_blocking=True), lengths, global_batch.language_ids)
                p = est.predict(dev, tfi, True, _no_graph=args.no_graph)
                host_out.copy_(p._flat, non_blocking=True)
                torch.cuda.synchronize()

            for _ in range(args.warmup):
                host_step()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                host_step()
            e_host = time.perf_counter() - t0
            est.close()
            result["pcie_inclusive"] = {
                "value": frames_global * args.steps / e_host, "unit": "frames/s", "ms_per_step": e_host / args.steps * 1e3,
                "h2d_bytes": host_audio.numel() * 4, "d2h_bytes": host_out.numel() * 4,
                "note": "pinned host audio -> HBM, forward pass, every log-probability -> pinned host memory, host synchronised "
                        "each step (no overlap between steps): what a caller that owns host buffers sees; `value` above is the "
                        "HBM-resident rate",
            }
        except Exception as exc:  # informational leg
            result["pcie_inclusive"] = {"error": repr(exc)}
    # the single-plane 16-bit throughput mode of the same workload (error measured and bounded in tests/, not a parity
    # mode): reported beside the parity-mode headline, never as `value`
    if args.also and args.also != args.precision and world == 1:
        e2, t2 = measure(args.also, args.steps, args.warmup, local)
        w2, roof2 = summarize(args.also, args.steps, t2, n_local)
        if rank == 0:
            result["throughput_mode"] = {
                "dtype": args.also, "value": frames_global * args.steps / e2, "unit": "frames/s", "ms_per_step": e2 / args.steps * 1e3,
                "roofline": roof2,
                "kernels": {k: {"ms_per_step": round(v[0] / args.steps, 4)} for k, v in t2.items()},
                "note": "single 16-bit plane per operand (1 MFMA per product); max-abs log-prob error vs the reference "
                        "2.3e-1 (bf16) / 3.0e-2 (f16) at XLS-R shape, see DESIGN.md section 3 -- NOT a parity mode",
            }
    # N = 1: what a ragged batch gets (informational; `value` is the equal-length config above): the encoder layers run on the
    # valid frames only and the conv stack skips tiles that lie in padding, against the padded layout of the same batch
    if world == 1 and not args.no_ragged and n_global > 1:
        try:
            g = torch.Generator().manual_seed(9)
            r_lengths = torch.randint(int(0.2 * length), length + 1, (n_global,), generator=g)
            r_lengths[0] = length
            r_audio = torch.randn(n_global, length, generator=g) * 0.1
            for i in range(n_global):
                r_audio[i, int(r_lengths[i]):] = 0.0
            r_batch = Batch(r_audio.to(device), r_lengths, torch.zeros(n_global, dtype=torch.long))
            est = Estimator(spec, state, device, args.precision)
            timings = {}
            for label, no_pack in (("packed", False), ("padded", True)):
                for _ in range(args.warmup):
                    pred = est.predict(r_batch, tfi, True, _no_pack=no_pack)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    pred = est.predict(r_batch, tfi, True, _no_pack=no_pack)
                torch.cuda.synchronize()
                timings[label] = (time.perf_counter() - t0) / args.steps
            valid = int(pred.lengths.sum())
            est.close()
            result["ragged_batch"] = {
                "workload": f"{n_global} utterances of U[{0.2 * args.seconds:g}, {args.seconds:g}] s padded to {args.seconds:g} s, same model",
                "padding_efficiency": valid / (n_global * int(pred.lengths.max())),
                "value": valid / timings["packed"], "unit": "valid frames/s", "ms_per_step": timings["packed"] * 1e3,
                "padded_layout_ms_per_step": timings["padded"] * 1e3,
                "note": "encoder layers on packed rows + conv stack skipping padding tiles (default for ragged batches) against "
                        "AMX_FLAG_NO_PACK, which computes every padded frame like the reference",
            }
        except Exception as exc:  # informational leg: never costs the headline
            result["ragged_batch"] = {"error": repr(exc)}
    if rank == 0:
        # CPU baseline on the first `cpu_sample` utterances of the benchmark batch (N = 1 only), and -- AFTER every timed
        # region -- the parity spot check of the path that was timed: the outputs of one more `predict` of the benchmark batch
        # against the CPU oracle on those utterances (each result is independent of the batch it sits in: SURVEY.md Appendix
        # A), log-probs on valid frames.  Without the baseline leg the check runs the oracle on one utterance alone.
        oracle_out = oracle_len = None
        # default: the whole batch (configs 2 / 4: ~25 s per oracle run); config 5's 60 s utterances: the first two
        sample = (n_global if args.seconds <= 10.0 else min(2, n_global)) if args.cpu_sample <= 0 else min(args.cpu_sample, n_global)
        n_check = sample if world == 1 else 1
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"], oracle_out, oracle_len = cpu_baseline(spec, state, tfi, audio[:n_check].contiguous(),
                                                                          lengths[:n_check].contiguous())
        else:
            result["cpu_baseline"] = None
        if not args.no_spot_check:
            try:
                if oracle_out is None:
                    from oracle import allophant_oracle as O

                    n_check = 1
                    oracle_out, oracle_len = O.predict(audio[:1].contiguous(), lengths[:1].contiguous(), state, spec, tfi,
                                                       synthetic.category_offsets(spec) if spec.get("composition_categories") else None, True)
                est = Estimator(spec, state, device, args.precision)
                pred = est.predict(local, tfi, True, _no_graph=args.no_graph)  # rank 0's shard starts at utterance 0 of the global batch
                range_error = None
                try:
                    est.check_finite()  # AMX_ERANGE: an activation left the fp16 planes (non-finite logits on a valid frame)
                except FloatingPointError as exc:
                    range_error = str(exc)
                n_check = min(n_check, n_local)
                worst, where, finite = 0.0, None, True
                for name, expected in oracle_out.items():
                    got = pred.outputs[name][:, :n_check].cpu()
                    for i in range(n_check):
                        t_i = int(oracle_len[i])
                        err = (got[:t_i, i] - expected[:t_i, i]).abs().max().item()
                        if not math.isfinite(err):  # a NaN compares false with everything: it must fail, not pass
                            finite = False
                            worst, where = float("inf"), (i, name)
                        elif err > worst:
                            worst, where = err, (i, name)
                est.close()
                split = args.precision.endswith("x3")
                result["parity_spot_check"] = {
                    "max_abs": worst if finite else None, "finite": finite, "range_check": range_error or "ok",
                    "utterance": where[0] if where else None, "output": where[1] if where else None,
                    "utterances_checked": n_check, "outputs_checked": len(oracle_out), "gate": 1e-3 if split else None,
                    # the single-plane modes have no gate, but non-finite outputs fail in every mode
                    "passed": bool(finite and range_error is None and (worst < 1e-3 or not split)),
                    "against": "CPU oracle (oracle/allophant_oracle.py) on the same utterances, log-probabilities of valid frames; "
                               "run after the timed region on the same handle configuration and batch",
                }
            except Exception as exc:
                result["parity_spot_check"] = {"error": repr(exc), "passed": False}
            result["ok"] = bool(result["parity_spot_check"].get("passed"))
        else:
            result["ok"] = None  # not checked
    if use_dist:
        dist.barrier()  # the other ranks stay until rank 0 has finished its spot check: every rank leaves the group together
    sys.stdout.flush()
    os.dup2(saved_stdout, 1)
    os.close(saved_stdout)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        os.dup2(2, 1)  # teardown chatter, if any, stays off stdout too
        dist.destroy_process_group()
    if rank == 0 and result.get("ok") is False:
        # the line above is still printed (it says what failed); the exit status says that the number must not be used
        print("bench.py: the parity spot check of the timed path FAILED (see parity_spot_check)", file=sys.stderr)
        raise SystemExit(3)


if __name__ == "__main__":
    main()
