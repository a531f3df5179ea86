#!/bin/bash
# (gpurun call 4 of round 5) torch's small ops host vs device; section B again; the deferred per-layer statistics launch:
# its GPU test and the FP8 W + A + KV overhead over the warm plain loop with and without it (<= 3 %)
set -u
O=gpurun_out/r05c4; mkdir -p $O
python3 tools/torch_cpu_vs_gpu_ops.py 2>&1 | grep -v Warning | tee $O/torch_cpu_vs_gpu_ops.jsonl
timeout 600 python3 -m pytest tests/test_gpu_reference_live.py -q -m gpu --tb=short -k "device_equals or int4_awq" > $O/b.txt 2>&1
grep -n "^E   AssertionError\|passed\|failed" $O/b.txt | cut -c1-200 | tail -12
grep "^\[note\] INT4-AWQ" $O/b.txt
timeout 900 python3 -m pytest tests/test_gpu_host.py tests/test_gpu_kv_cache.py tests/test_gpu_moe.py -q -m gpu --tb=short 2>&1 | tail -8
for mode in off auto off auto; do
  python3 tools/hf_flow_check.py --layers 32 --batches 64 --qformat fp8 --defer-stats $mode > $O/flow_fp8_$mode.json 2> $O/flow_fp8_$mode.err
  python3 - $O/flow_fp8_$mode.json $mode <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "plain", d["plain_forward_loop_s"], "quantize", d["quantize_s"], "overhead %.2f %%" % (100 * (d["quantize_s"] / d["plain_forward_loop_s"] - 1)),
      d.get("max_calibrate_s"))
P
done
