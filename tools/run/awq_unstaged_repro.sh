#!/bin/bash
# (gpurun call of round 4) where do the driver's 17 unstaged seconds of the INT4-AWQ extra sit?
#  A = round 3's order (CPU baseline inside the process, before the AWQ extras), B = the new default (subprocess, last)
set -u
O=gpurun_out/awq_unstaged; mkdir -p $O
python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-baseline-inproc-first > $O/A_inproc_first.json 2> $O/A.err
echo "A rc=$?"
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/B_default.json 2> $O/B.err
echo "B rc=$?"
MOQ_BENCH_DEBUG_ONE_GPU=1 timeout 600 python3 bench.py --gpus 2 --steps 3 --warmup 1 --layers 2 --awq-layers 1 --awq-batches 2 > $O/C_n2_debug.json 2> $O/C.err
echo "C rc=$?"
python3 - <<'P'
import json
for f in ("A_inproc_first","B_default","C_n2_debug"):
    try:
        d=json.loads(open(f"gpurun_out/awq_unstaged/{f}.json").read().strip().splitlines()[-1])
        e=d.get("extra",{})
        print(f, d["value"], d["roofline"]["frac"], e.get("awq_wallclock_s"), (e.get("awq") or {}).get("stages_s"), (e.get("awq") or {}).get("quantize_stages_s"), (e.get("awq") or {}).get("unstaged_s"), (e.get("awq_hf_random_init") or {}).get("quantize_s"), d.get("cpu_baseline",{}) and d["cpu_baseline"].get("value"))
    except Exception as ex:
        print(f, "ERR", ex)
P
