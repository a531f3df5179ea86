#!/bin/bash
# round 3, GPU call 3: GEO 11 / 12 (register-prefetch loops) against GEO 10 / 4, same box; nccl world-1 test; HF flow after the host-math fix
set -u
O=gpurun_out/r03c; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_dist_nccl.py tests/test_gpu_awq_search.py -m gpu -q 2>&1 | tail -25 ) > $O/gpu_tests.txt
for G in 10 11 12 4; do MOQ_TUNE_GEMM_GEO=$G timeout 200 python tools/gemm_bench.py > $O/gemm_geo$G.md 2>&1; done
timeout 300 python tools/hf_flow_check.py --layers 32 --batches 64 --qformat int4_awq --note "r03c single-thread host math" >> $O/hf.jsonl 2> $O/hf.err
ls -la $O
