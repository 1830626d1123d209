"""Developer tool: per-product durations from a rocprofv3 --kernel-trace csv of tools/step_trace.py (32 x 10 s): the ping-pong GEMM launches of
a step are told apart by instance (the LayerNorm fold's consumers / producers are instances of their own) and by duration (QKV ~ 220 us
against FFN1 ~ 300, out-projection ~ 75 against FFN2 ~ 300)."""
import csv
import glob
import re
import sys

path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
groups = {}
for r in rows:
    name = r["Kernel_Name"]
    m = re.search(r"gemm_pp_kernelIDF16_Li2ELi8ELi4E(?:Li(\d)E)?", name)
    if not m:
        if "ln_finalize" in name: key = "ln_finalize"
        elif "rownorm_kernel" in name: key = "rownorm"
        elif "attn_kernel" in name: key = "attention"
        else: continue
    else:
        fold = int(m.group(1) or 0)
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        grid = int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)
        wgs = grid // 512 if grid else 0
        if wgs and wgs < 256: key = f"fold{fold} {'out-proj' if dur < 150 else 'FFN2'}"
        else: key = f"fold{fold} {'QKV' if dur < 262 else 'FFN1'}"
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    groups.setdefault(key, []).append(dur)
for k in sorted(groups):
    v = sorted(groups[k])
    print(f"{k:22s} n {len(v):5d}  median {v[len(v) // 2]:8.1f} us  mean {sum(v) / len(v):8.1f}  p10 {v[len(v) // 10]:8.1f}  p90 {v[9 * len(v) // 10]:8.1f}")
