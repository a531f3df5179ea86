#!/bin/bash
# (gpurun call 3 of round 5) tests/test_gpu_reference_live.py in full at HEAD (section B in numerics mode "device")
set -u
O=gpurun_out/r05c3; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_reference_live.py -q -m gpu --tb=short > $O/reference_live_full.txt 2>&1
grep -v "Warning\|warnings.warn\|^  \|Searching for sparse" $O/reference_live_full.txt | tail -70
cp gpurun_out/reference_own_gpu_tests_seams.txt $O/ 2>/dev/null
grep "^\[seams\]" $O/reference_own_gpu_tests_seams.txt | head -30
