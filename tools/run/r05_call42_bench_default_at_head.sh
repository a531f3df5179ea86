#!/bin/bash
# (gpurun call 42 of round 5) the default bench line at HEAD, first command of a fresh lease (what the driver runs at round end)
set -u
O=gpurun_out/${1:-r05o}; mkdir -p $O
( time timeout 280 python3 bench.py > $O/bench_default.json 2> $O/bench.err ) 2> $O/bench_time.txt
echo "bench rc=$? $(grep real $O/bench_time.txt)"
python3 - "$O" <<'P'
import json, sys
d=json.loads(open(sys.argv[1]+"/bench_default.json").read().strip().splitlines()[-1]); e=d["extra"]
print(d["value"], d["unit"], d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "awq", e.get("awq_wallclock_s"), (e.get("awq") or {}).get("passes"))
h=e.get("awq_hf_random_init") or {}
print("hf", h.get("quantize_s"), (h.get("stats") or {}).get("stages_s"))
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
P
tail -3 $O/bench.err | cut -c1-300
