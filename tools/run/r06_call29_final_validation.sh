#!/bin/bash
# (gpurun call 29 of round 6) HEAD after the device-fuzz fixes: default bench line first, the whole GPU suite as the driver runs
# it (serial, -x), smoke, the per-kernel table
set -u
O=gpurun_out/${1:-r06c29}; mkdir -p $O
export TMPDIR=/tmp
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench.err ) 2> $O/bench_time.txt
echo "bench rc=$? $(grep real $O/bench_time.txt)"
python3 - "$O" <<'P'
import json, sys
d=json.loads(open(sys.argv[1]+"/bench_default.json").read().strip().splitlines()[-1]); e=d["extra"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "awq", e.get("awq_wallclock_s"), "cold", ((e.get("awq") or {}).get("cold_process") or {}).get("quantize_s"), "hf", (e.get("awq_hf_random_init") or {}).get("quantize_s"))
for k in ("per_tensor_amax","mask_2to4","mxfp4_g32_qdq","int4g128_fused_amax_qdq","llama3_70b_int4g128_inplace","llama3_70b_mxfp4_sq"):
    print(k, json.dumps(e.get(k))[:200])
P
( time timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1 ) 2> $O/suite_time.txt
echo "suite rc=$? $(grep real $O/suite_time.txt)"; grep "passed\|failed\|^E  \|^FAILED" $O/gpu_suite.log | tail -8 | cut -c1-300
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 600 python3 tools/kbench.py > $O/kbench.md 2> $O/kbench.err; echo "kbench rc=$?"; wc -l $O/kbench.md
