#!/bin/bash
# (gpurun call of round 4) the default bench line in the driver's form, nothing else
set -u
O=gpurun_out/${1:-r04z}; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench.err
echo "bench rc=$?"
python3 - "$O" <<'P'
import json, sys
d=json.loads(open(sys.argv[1]+"/bench_default.json").read().strip().splitlines()[-1]); e=d["extra"]
print(d["value"], d["roofline"]["frac"], "awq", e.get("awq_wallclock_s"), (e.get("awq") or {}).get("stages_s"))
h=e.get("awq_hf_random_init") or {}
print("hf", h.get("quantize_s"), (h.get("stats") or {}).get("stages_s"))
P
