#!/bin/bash
# (gpurun call 5 of round 5; calls 3 / 4 were lost with the container) tests/test_gpu_reference_live.py in full at HEAD,
# the deferred per-layer statistics launch (its GPU tests + FP8 W + A + KV overhead with / without it), torch's small ops host vs device
set -u
O=gpurun_out/r05c5; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_reference_live.py -q -m gpu --tb=short -rA > $O/reference_live_full.txt 2>&1
grep -v "^PASSED\|Warning\|warnings.warn\|^  \|Searching for sparse\|Inserted \|Captured stdout\|^_____" $O/reference_live_full.txt | tail -60
cp gpurun_out/reference_own_gpu_tests_seams.txt $O/ 2>/dev/null
grep "^\[seams\]" $O/reference_own_gpu_tests_seams.txt | head -30
python3 tools/torch_cpu_vs_gpu_ops.py 2>&1 | grep -v Warning | tee $O/torch_cpu_vs_gpu_ops.jsonl | cut -c1-300
timeout 900 python3 -m pytest tests/test_gpu_host.py tests/test_gpu_kv_cache.py tests/test_gpu_moe.py -q -m gpu --tb=short 2>&1 | tail -8 | tee $O/host_kv_moe_tail.txt
for mode in off auto off auto; do
  python3 tools/hf_flow_check.py --layers 32 --batches 64 --qformat fp8 --defer-stats $mode > $O/flow_fp8_$mode.json 2> $O/flow_fp8_$mode.err
  python3 - $O/flow_fp8_$mode.json $mode <<'P'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "plain", d["plain_forward_loop_s"], "quantize", d["quantize_s"], "overhead %.2f %%" % (100 * (d["quantize_s"] / d["plain_forward_loop_s"] - 1)),
      d.get("max_calibrate_s"))
P
done
