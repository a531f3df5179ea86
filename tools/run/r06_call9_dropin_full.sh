#!/bin/bash
# (gpurun call 9 of round 6) the drop-in timing at FULL depth: 32 decoder layers of Llama-3-8B width, 64 x 4096 calibration tokens
set -u
O=gpurun_out/${1:-r06c9}; mkdir -p $O
export TMPDIR=/tmp
timeout 3300 python3 tools/dropin_bench.py --layers 32 --batches 64 --rows 8 --seq 512 --formats int4_awq,fp8,int8_sq --out $O/dropin_full.json > $O/dropin.log 2> $O/dropin.err
echo "dropin rc=$?"; grep "^{\"fp8\|^{\"int" $O/dropin.log | cut -c1-460; tail -3 $O/dropin.err | cut -c1-300
