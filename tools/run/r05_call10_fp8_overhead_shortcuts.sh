#!/bin/bash
# (gpurun call 10 of round 5) FP8 W + A + KV overhead with the quantizer shortcuts (QuantLinear / attention skip calls that hand
# their input back), host-side clocks of both loops; the suites that walk those paths
set -u
O=gpurun_out/r05c10; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_host.py tests/test_gpu_kv_cache.py tests/test_gpu_moe.py tests/test_gpu_export.py tests/test_gpu_layerwise.py tests/test_gpu_awq_search.py -m gpu -q -n 2 --tb=short 2>&1 | grep -v "Warning\|warnings.warn\|^  " | tail -5 | tee $O/tests_tail.txt
for mode in auto auto off auto; do
  python3 tools/hf_flow_check.py --layers 32 --batches 64 --qformat fp8 --defer-stats $mode > $O/flow_fp8_$mode.json 2> $O/flow_fp8_$mode.err
  python3 - $O/flow_fp8_$mode.json $mode <<'P' | tee -a $O/fp8_overhead.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
m = d.get("max_calibrate_s") or {}
print(sys.argv[2], "plain", d["plain_forward_loop_s"], "quantize", d["quantize_s"], "overhead %.2f %%" % (100 * (d["quantize_s"] / d["plain_forward_loop_s"] - 1)),
      "loop", m.get("forward_loop_s"), "host", d.get("host_side"), "stages", d.get("quantize_stages_s"), {k: v for k, v in m.items() if k.endswith("_s")})
P
done
