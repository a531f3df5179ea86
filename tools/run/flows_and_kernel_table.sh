#!/bin/bash
# (gpurun call of round 4) device-side entropy / percentile searches (tests + flow timing) and the per-kernel table on the new grids
set -u
O=gpurun_out/r04j; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_host.py tests/test_gpu_calibrate_weights.py tests/test_gpu_reference_style.py tests/test_gpu_dist_nccl.py -x -q > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log
for q in int8_entropy int8_percentile; do
  python3 tools/hf_flow_check.py --layers 4 --batches 16 --qformat $q > $O/flow_$q.json 2> $O/flow_$q.err; echo "$q rc=$?"; cat $O/flow_$q.json
done
python3 tools/kbench.py row_hist > $O/kernel_rowhist.md 2>&1; cat $O/kernel_rowhist.md | tail -2
