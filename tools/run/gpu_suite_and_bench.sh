#!/bin/bash
# (gpurun call of round 4) the whole GPU suite at HEAD + smoke + the full default bench line
set -u
O=gpurun_out/${1:-r04s}; mkdir -p $O; export BENCH_O=$O
timeout 1500 python3 -m pytest tests -m gpu -x -q -n 2 > $O/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -8 $O/gpu_suite.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench.err
echo "bench rc=$?"
python3 - <<'P'
import json
d=json.loads(open(""+__import__("os").environ.get("BENCH_O","gpurun_out/r04s")+"/bench_default.json").read().strip().splitlines()[-1])
e=d.get("extra",{})
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic_source"), d["roofline"].get("node_copy_GBs"), d["roofline"].get("node_read_GBs"))
for k in ("per_tensor_amax","qdq_out_of_place","mask_2to4","mxfp4_g32_qdq","fp8_mask24_step","int4g128_fused_amax_qdq","llama3_70b_int4g128_inplace","scale_base_n1"):
    print(k, e.get(k))
print("awq", e.get("awq_wallclock_s"), (e.get("awq") or {}).get("stages_s"), (e.get("awq") or {}).get("quantize_stages_s"), (e.get("awq") or {}).get("tie_check"))
h=e.get("awq_hf_random_init") or {}
print("hf", h.get("quantize_s"), h.get("quantize_stages_s"), (h.get("stats") or {}).get("stages_s"), (h.get("stats") or {}).get("passes"), (h.get("stats") or {}).get("replayed_passes"), (h.get("stats") or {}).get("tie_check"))
print("cpu", d.get("cpu_baseline"))
P
