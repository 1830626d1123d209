#!/bin/bash
# round 6, run 15: first bench lines of the XLS-R 1B / 2B encoders (48 layers, hidden 1280 / 1920) on 32 x 10 s
mkdir -p gpurun_out
O=gpurun_out
(timeout 900 python bench.py --encoder xlsr-1b --also "" --cpu-sample 2 --no-ragged > $O/r06_probe_xlsr1b.json 2> $O/r06_probe_xlsr1b.err; echo rc=$? >> $O/r06_probe_xlsr1b.err)
(timeout 900 python bench.py --encoder xlsr-2b --also "" --cpu-sample 2 --no-ragged > $O/r06_probe_xlsr2b.json 2> $O/r06_probe_xlsr2b.err; echo rc=$? >> $O/r06_probe_xlsr2b.err)
python - <<'PY'
import json
for n in ("xlsr1b", "xlsr2b"):
    try:
        d = json.load(open(f"gpurun_out/r06_probe_{n}.json"))
        print(n, d["ms_per_step"], d["value"], d["ok"], d["roofline"]["frac"], d["roofline"]["whole_block"]["frac"], d["kernels"], d["parity_spot_check"], d["pass"])
    except Exception as e:
        print(n, "ERROR", e)
PY
tail -5 $O/r06_probe_xlsr1b.err $O/r06_probe_xlsr2b.err
