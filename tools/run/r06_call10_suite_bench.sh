#!/bin/bash
# (gpurun call 10 of round 6) default bench line FIRST on the fresh lease (all extras: cold AWQ process, llama3_70b_mxfp4_sq),
# the whole GPU suite, smoke, the drop-in timing (4 layers / 64k tokens), the N = 1, 2 control flow of the scaling tool
set -u
O=gpurun_out/${1:-r06c10}; mkdir -p $O
export TMPDIR=/tmp
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench.err ) 2> $O/bench_time.txt
echo "bench rc=$? $(grep real $O/bench_time.txt)"
python3 - "$O" <<'P'
import json, sys
d=json.loads(open(sys.argv[1]+"/bench_default.json").read().strip().splitlines()[-1]); e=d["extra"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], "awq", e.get("awq_wallclock_s"), (e.get("awq") or {}).get("stages_s"), (e.get("awq") or {}).get("passes"))
print("cold", json.dumps((e.get("awq") or {}).get("cold_process"))[:900])
print("tie", (e.get("awq") or {}).get("tie_check"))
h=e.get("awq_hf_random_init") or {}
print("hf", h.get("quantize_s"), (h.get("stats") or {}).get("stages_s"), (h.get("stats") or {}).get("tie_check"))
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
for k in ("per_tensor_amax","mask_2to4","mxfp4_g32_qdq","fp8_mask24_step","int4g128_fused_amax_qdq","llama3_70b_int4g128_inplace","llama3_70b_mxfp4_sq","scale_base_n1"):
    print(k, json.dumps(e.get(k))[:260])
P
timeout 2400 python3 -m pytest tests -m gpu -q -n 2 --tb=short > $O/gpu_suite.log 2>&1
echo "suite rc=$?"; grep "passed\|failed\|^E  \|^FAILED" $O/gpu_suite.log | tail -14 | cut -c1-300
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
