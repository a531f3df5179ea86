#!/bin/bash
# (gpurun call 9 of round 5) FIRST command of a fresh lease = the default bench line in the driver's form; the whole GPU
# suite + smoke at HEAD; rocprofv3 kernel trace + PMC of the headline workload with a bench line of the same box; FP8 W + A + KV
# overhead with the host-side validate / lean quantizer construction; the per-kernel table
set -u
O=gpurun_out/r05c9; mkdir -p $O
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench.err ) 2> $O/bench_time.txt
echo "bench rc=$? $(grep real $O/bench_time.txt)"
python3 - "$O" <<'P'
import json, sys
d=json.loads(open(sys.argv[1]+"/bench_default.json").read().strip().splitlines()[-1]); e=d["extra"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic_source"), "awq", e.get("awq_wallclock_s"), (e.get("awq") or {}).get("stages_s"), (e.get("awq") or {}).get("passes"))
h=e.get("awq_hf_random_init") or {}
print("hf", h.get("quantize_s"), (h.get("stats") or {}).get("stages_s"), (h.get("stats") or {}).get("tie_check"))
print("cpu", json.dumps(d.get("cpu_baseline"))[:600])
P
timeout 1500 python3 -m pytest tests -m gpu -q -n 2 --tb=short > $O/gpu_suite.log 2>&1
echo "suite rc=$?"; grep -v "Warning\|warnings.warn\|^  " $O/gpu_suite.log | tail -30 | cut -c1-250
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
bash tools/profile_bench.sh r05h_fp8 --workload fp8
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/prof/r05h_fp8_bench_line.json 2> gpurun_out/prof/r05h_fp8_bench.err
cut -c1-700 gpurun_out/prof/r05h_fp8_bench_line.json; echo
head -14 gpurun_out/prof/r05h_fp8_summary.md
find gpurun_out/prof -name '*.csv' -size +2M -delete 2>/dev/null
for mode in auto off auto; do
  python3 tools/hf_flow_check.py --layers 32 --batches 64 --qformat fp8 --defer-stats $mode > $O/flow_fp8_$mode.json 2> $O/flow_fp8_$mode.err
  python3 - $O/flow_fp8_$mode.json $mode <<'P' | tee -a $O/fp8_overhead.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "plain", d["plain_forward_loop_s"], "quantize", d["quantize_s"], "overhead %.2f %%" % (100 * (d["quantize_s"] / d["plain_forward_loop_s"] - 1)),
      d.get("quantize_stages_s"), d.get("max_calibrate_s"))
P
done
python3 tools/kbench.py 2>&1 | grep -v Warning > $O/kernel_table.md; tail -12 $O/kernel_table.md | cut -c1-200
