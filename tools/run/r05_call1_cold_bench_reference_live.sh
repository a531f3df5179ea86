#!/bin/bash
# (gpurun call 1 of round 5)  FIRST command of a fresh lease = the default bench line in the driver's form (VERDICT r4 next #3:
# a builder line from a cold lease); then the reference on the device (next #1), then round 4's un-run second half (next #2).
set -u
O=gpurun_out/r05c1; mkdir -p $O
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench.err ) 2> $O/bench_time.txt
echo "bench rc=$? $(grep real $O/bench_time.txt)"
python3 - "$O" <<'P'
import json, sys
d=json.loads(open(sys.argv[1]+"/bench_default.json").read().strip().splitlines()[-1]); e=d["extra"]
print(d["value"], d["roofline"]["frac"], "awq", e.get("awq_wallclock_s"), (e.get("awq") or {}).get("stages_s"), (e.get("awq") or {}).get("passes"),
      (e.get("awq") or {}).get("forward_loop_calls"), (e.get("awq") or {}).get("warm_forward_s"), (e.get("awq") or {}).get("stored_input_bytes"))
h=e.get("awq_hf_random_init") or {}
print("hf", h.get("quantize_s"), (h.get("stats") or {}).get("stages_s"))
print("cpu", json.dumps(d.get("cpu_baseline"))[:900])
P
timeout 900 python3 -m pytest tests/test_gpu_reference_live.py -q -m gpu -rA 2>&1 | grep -v "^PASSED\|Warning\|warnings.warn\|^  " | tail -80 > $O/reference_live_tail.txt
tail -60 $O/reference_live_tail.txt
cp gpurun_out/reference_own_gpu_tests_seams.txt $O/ 2>/dev/null
bash tools/run/host_side_second_half_check.sh 2>&1 | tail -12
