"""Aggregates rocprofv3 --pmc counter_collection CSVs per kernel name (sum over dispatches, and per-dispatch mean)."""
import csv, glob, sys, collections, json
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for path in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        short = name.replace("(anonymous namespace)::", "").split("(")[0][-90:]
        agg[short][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[short][r["Counter_Name"]] += 1
out = {}
for k in agg:
    out[k] = {c: {"sum": v, "dispatches": cnt[k][c], "per_dispatch": v / cnt[k][c]} for c, v in agg[k].items()}
keys = sorted(out, key=lambda k: -max(v["sum"] for v in out[k].values()))
for k in keys[:8]:
    print(k)
    for c, v in sorted(out[k].items()):
        print(f"    {c:32s} per-dispatch {v['per_dispatch']:.4g}  (n={v['dispatches']})")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
