#!/bin/bash
set -u
O=gpurun_out/r03g; mkdir -p $O
timeout 300 python tools/exp/gemm_pitch_probe.py > $O/gemm_pitch.md 2>&1
( timeout 300 python -m pytest tests/test_gpu_gemm.py -m gpu -q 2>&1 | tail -4 ) > $O/gpu_tests.txt
cat $O/gemm_pitch.md
