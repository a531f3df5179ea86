#!/bin/bash
# the default N = 1 bench path after the bench.py edits (short AWQ extra, no HF flow)
set -u
O=gpurun_out/r03zt; mkdir -p $O
timeout 110 python bench.py --awq-layers 2 --awq-batches 8 --no-hf > $O/bench_line.json 2> $O/bench.err
echo "rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03zt/bench_line.json")); e=d.get("extra",{})
print(d["value"], d["scaling"], d["roofline"]["frac"], d["cpu_baseline"]["value"], sorted(e))
PY
tail -2 $O/bench.err
