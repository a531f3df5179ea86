#!/bin/bash
# round 3, GPU call 6: fp32 matrix-core GEMM tests, SparseGPT trailing kernel with prefetched RMW, rocprof of the FP8 PTQ flow
set -u
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_sparsegpt.py tests/test_gpu_awq_search.py tests/test_gpu_host.py tests/test_gpu_layerwise.py -m gpu -q 2>&1 | tail -12 ) > $O/gpu_tests.txt
timeout 300 python tools/sgpt_bench.py > $O/sgpt_table.md 2> $O/sgpt.err
R=$(pwd); cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof_fp8 -o fp8 -- python $R/tools/hf_flow_check.py --layers 8 --batches 16 --qformat fp8 --note "8 layers x 16 batches under rocprof" >> $R/$O/hf.jsonl 2> $R/$O/hf_prof.err
cd $R
find $O/prof_fp8 -type f ! -name '*stats*.csv' -delete 2>/dev/null
ls -la $O
