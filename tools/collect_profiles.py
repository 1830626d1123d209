"""Copies the summaries of tools/profile_bench.sh runs (gpurun_out/prof_<tag>/) into profiles/ and rebuilds
profiles/<round>_traffic.json: the per-launch HBM traffic of the ping-pong GEMM that bench.py reports as
`roofline.traffic`, and the traffic of the conv-stage kernels (`roofline.conv_stage.*.traffic`).

    python tools/collect_profiles.py r02 f16x3:prof_r02_f16x3 bf16:prof_r02_bf16

Correction applied (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts half of the bytes of 16-byte-per-lane streaming
reads (global_load and buffer_load ... lds alike), so it is doubled; WRITE_SIZE is exact; both are in KiB; hits in the
Infinity Cache are counted.  FETCH_SIZE and WRITE_SIZE come from separate --pmc passes.
"""
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
round_tag = sys.argv[1]
traffic = {}
hashes = set()
for arg in sys.argv[2:]:
    precision, directory = arg.split(":")
    src = os.path.join(ROOT, "gpurun_out", directory)
    with open(os.path.join(src, "kernel_source_hash.txt")) as f:  # written on the box by tools/profile_bench.sh
        hashes.add(f.read().strip())
    dst = os.path.join(ROOT, "profiles", f"{round_tag}_{precision}")
    shutil.copy(os.path.join(src, "kernel_stats.csv"), dst + "_kernel_stats.csv")
    shutil.copy(os.path.join(src, "bench_traced.json"), dst + "_bench_under_rocprof.json")
    merged = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        summary = json.load(open(os.path.join(src, f"pmc_{counter}_summary.json")))
        for kernel, counters in summary.items():
            merged.setdefault(kernel, {}).update(counters)
    json.dump(merged, open(dst + "_pmc_hbm.json", "w"), indent=1)
    mfma_path = os.path.join(src, "pmc_MFMA_summary.json")
    if os.path.exists(mfma_path):
        # SQ_VALU_MFMA_BUSY_CYCLES is summed over SIMDs; GRBM_GUI_ACTIVE over the 8 XCDs: SIMD-cycles available to a
        # dispatch = GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs
        mfma = {}
        for kernel, c in json.load(open(mfma_path)).items():
            if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c or c["GRBM_GUI_ACTIVE"]["sum"] <= 0:
                continue
            row = {name: v["per_dispatch"] for name, v in c.items()}
            row["dispatches"] = c["GRBM_GUI_ACTIVE"]["dispatches"]
            row["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"] / (c["GRBM_GUI_ACTIVE"]["sum"] * 128.0)
            mfma[kernel] = row
        json.dump({"formula": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs); counters per "
                              "dispatch, rocprofv3 --pmc pass of bench.py --steps 2", "kernels": mfma},
                  open(dst + "_pmc_mfma.json", "w"), indent=1)
        for kernel, row in sorted(mfma.items(), key=lambda kv: -kv[1]["mfma_busy_frac"])[:4]:
            print(f"  mfma busy {row['mfma_busy_frac']:.3f}  {kernel[-70:]}")
    # the dominant kernel: the 256-row instance of the ping-pong GEMM (the 128-row instance is the last conv layer)
    # (round 6: the instances carry the tile width and the LayerNorm-fold role behind the tile height: <T, planes, 8, NI, FOLD>)
    gemm = [v for k, v in merged.items() if re.search(r"gemm_pp_kernelI\w+?Li[12]ELi8E", k) or ",8>" in k.replace(" ", "") or ", 8, " in k] or \
           [v for k, v in merged.items() if "gemm_pp_kernel" in k]
    if not gemm:
        raise SystemExit(f"no gemm_pp_kernel counters in {src}")

    def per_dispatch(rows):
        fetch = sum(v["FETCH_SIZE"]["sum"] for v in rows) / sum(v["FETCH_SIZE"]["dispatches"] for v in rows)
        write = sum(v["WRITE_SIZE"]["sum"] for v in rows) / sum(v["WRITE_SIZE"]["dispatches"] for v in rows)
        return fetch, write

    fetch_kb, write_kb = per_dispatch(gemm)
    traffic[precision] = {
        "hbm_bytes_per_launch": (2.0 * fetch_kb + write_kb) * 1024.0,
        "fetch_size_kb_raw": fetch_kb,
        "write_size_kb": write_kb,
        "correction": "MI355X_MICROARCH.md 'HBM': FETCH_SIZE counts 1/2 of the bytes of 16-B-per-lane streaming reads "
                      "(global_load and buffer_load ... lds alike) -> doubled; WRITE_SIZE exact; KiB units; Infinity-Cache hits "
                      "are counted",
        "source": f"profiles/{round_tag}_{precision}_pmc_hbm.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, "
                  "bench.py --steps 2)",
    }
    conv0 = [v for k, v in merged.items() if "conv0_kernel" in k or "conv0_mfma_kernel" in k]
    if conv0:
        f, w = per_dispatch(conv0)
        # conv0 reads fp32 audio with 4-byte-per-lane loads (FETCH_SIZE calibration for that width is not in the guide: the
        # raw and the doubled figure are both kept) and writes 16-byte-per-lane plane stores (WRITE_SIZE exact)
        traffic[precision]["conv0_hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
        traffic[precision]["conv0_fetch_size_kb_raw"] = f
        traffic[precision]["conv0_write_size_kb"] = w
    ln = [v for k, v in merged.items() if "gemm_ln_kernel" in k or "gemm_ln_il_kernel" in k]
    if ln:
        f = sum(v["FETCH_SIZE"]["sum"] for v in ln)
        w = sum(v["WRITE_SIZE"]["sum"] for v in ln)
        steps = max(1, ln[0]["FETCH_SIZE"]["dispatches"] // 5)  # 5 launches (conv layers 1-5) per step
        traffic[precision]["gemm_ln_hbm_bytes_per_step"] = (2.0 * f + w) * 1024.0 / steps
    print(precision, {k: round(v, 1) if isinstance(v, float) else v for k, v in traffic[precision].items() if k != "correction"})
if len(hashes) != 1:
    raise SystemExit(f"the profile directories were measured on different kernel sources: {sorted(hashes)}")
traffic["kernel_source_hash"] = hashes.pop()
json.dump(traffic, open(os.path.join(ROOT, "profiles", f"{round_tag}_traffic.json"), "w"), indent=1)
