#!/bin/bash
# (gpurun call 8 of round 5) the 1024-thread running fold: FP8 W + A + KV overhead over the warm plain loop with / without the
# per-layer statistics launch (<= 3 %), kernel totals of the deferred run, the ordered finalize with 128
# loads in flight (awq_weight_scale), reduce_amax over apart kept dims
set -u
O=gpurun_out/r05c8; mkdir -p $O
ROOT=$(pwd); export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_host.py tests/test_gpu_awq_search.py -m gpu -q -n 2 --tb=short 2>&1 | grep -v "Warning\|warnings.warn\|^  " | tail -6 | tee $O/tests_tail.txt
for mode in off auto off auto; do
  python3 tools/hf_flow_check.py --layers 32 --batches 64 --qformat fp8 --defer-stats $mode > $O/flow_fp8_$mode.json 2> $O/flow_fp8_$mode.err
  python3 - $O/flow_fp8_$mode.json $mode <<'P' | tee -a $O/fp8_overhead.txt
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "plain", d["plain_forward_loop_s"], "quantize", d["quantize_s"], "overhead %.2f %%" % (100 * (d["quantize_s"] / d["plain_forward_loop_s"] - 1)),
      d.get("max_calibrate_s"))
P
done
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$O/prof_flow_auto -o flow -- python3 $ROOT/tools/hf_flow_check.py --layers 32 --batches 64 --qformat fp8 --defer-stats auto > $ROOT/$O/flow_prof_auto.json 2> $ROOT/$O/flow_prof_auto.err
rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$O/prof_small -o small -- python3 $ROOT/tools/kbench.py "awq_weight_scale,col_abs" > $ROOT/$O/prof_small.log 2>&1
cd $ROOT
python3 tools/kstats_all_md.py $O/prof_flow_auto 30 > $O/flow_kernels_auto.md; grep "moq::\|Total" $O/flow_kernels_auto.md | cut -c1-170
python3 tools/kstats_md.py $O/prof_small | tee $O/small_kernels_kernel_only.md
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
