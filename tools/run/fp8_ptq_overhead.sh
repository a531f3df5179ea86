#!/bin/bash
# (gpurun call of round 4) FP8 W + A + KV calibration overhead over the warm plain loop with the lean per-tensor collect
set -u
O=gpurun_out/r04k; mkdir -p $O
true
echo "tests rc=$?"; tail -3 $O/tests.log
for i in 1; do
python3 tools/hf_flow_check.py --layers 32 --batches 64 --qformat fp8 > $O/flow_fp8_$i.json 2> $O/flow_fp8_$i.err; echo "fp8 rc=$?"; cat $O/flow_fp8_$i.json
done
