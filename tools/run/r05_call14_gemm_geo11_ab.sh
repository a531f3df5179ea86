#!/bin/bash
# (gpurun call 14 of round 5) GEO 11 = GEO 10 with the loss epilogue's out_actual tile in two halves, the first carried by
# the last K iteration's (otherwise empty) LDS-DMA pieces: correctness under the GEMM / AWQ-search suites, then A/B on one box
set -u
O=gpurun_out/r05c14; mkdir -p $O
MOQ_TUNE_GEMM_GEO=11 timeout 900 python3 -m pytest tests/test_gpu_gemm.py tests/test_gpu_awq_search.py -m gpu -q --tb=short 2>&1 | grep -v "Warning\|warnings.warn\|^  " | tail -12 | cut -c1-300 | tee $O/geo11_tests_tail.txt
for g in 10 11 10 11; do
  MOQ_TUNE_GEMM_GEO=$g python3 tools/gemm_bench.py 2>/dev/null | grep "^| 8b\|^| 70b\|^| square" | cut -d'|' -f2,3,4,6,7,10 | sed "s/^/GEO $g /" | tee -a $O/gemm_ab.txt
done
