#!/bin/bash
# round 3, GPU call 4: suite at HEAD, SparseGPT table (pipelined trailing kernel), FP8 PTQ overhead after the
# weights-once change, histogram sweep after the hoisted loads
set -u
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -14 ) > $O/gpu_tests.txt
timeout 300 python tools/sgpt_bench.py > $O/sgpt_table.md 2> $O/sgpt.err
timeout 300 python tools/hf_flow_check.py --layers 32 --batches 64 --qformat fp8 --note "r03d weights collected once" >> $O/hf.jsonl 2> $O/hf1.err
timeout 300 python tools/hf_flow_check.py --layers 32 --batches 64 --qformat int8_sq --note "r03d" >> $O/hf.jsonl 2> $O/hf2.err
timeout 200 python tools/exp/hist_sweep.py > $O/hist_sweep.txt 2>&1
ls -la $O
