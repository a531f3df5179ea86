#!/bin/bash
# (gpurun call 2 of round 5) the failing sections of tests/test_gpu_reference_live.py with their tracebacks; the reference's own
# GPU tests with the full log; FP8 W + A + KV PTQ overhead at HEAD
set -u
O=gpurun_out/r05c2; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_reference_live.py -q -m gpu --tb=short -x -k "device_equals" > $O/b_first_failure.txt 2>&1
grep -v "Warning\|warnings.warn" $O/b_first_failure.txt | tail -60
timeout 600 python3 -m pytest tests/test_gpu_reference_live.py -q -m gpu --tb=line -k "device_equals or int4_awq" 2>&1 | grep -v "Warning\|warnings.warn\|^  " | tail -40 | tee $O/b_all_lines.txt
timeout 600 python3 -m pytest tests/test_gpu_reference_live.py -q -m gpu --tb=short -k "own_gpu_tests" 2>&1 | tail -15
cp gpurun_out/reference_own_gpu_tests_seams.txt $O/
grep -n "test_overflow_fp16" -B2 -A40 gpurun_out/reference_own_gpu_tests_seams.txt | grep -m1 -A45 "^[0-9]*[-:]___.*test_overflow_fp16" | head -70
python3 tools/hf_flow_check.py --layers 32 --batches 64 --qformat fp8 > $O/flow_fp8.json 2> $O/flow_fp8.err; echo "fp8 rc=$?"; cat $O/flow_fp8.json
