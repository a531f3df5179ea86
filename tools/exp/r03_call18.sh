#!/bin/bash
# epilogue rewrite (branch-free interior tiles): parity first, then the GEMM table and both AWQ flows
set -u
O=gpurun_out/r03r; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_awq_search.py tests/test_gpu_sparsegpt.py tests/test_gpu_host.py -m gpu -q -x 2>&1 | tail -5 ) > $O/gpu_tests.txt
timeout 300 python tools/gemm_bench.py > $O/gemm_table.md 2> $O/gemm_table.err
timeout 300 python tools/awq_bench.py --layers 32 --batches 64 --search auto > $O/awq_auto.json 2> $O/awq_auto.err
timeout 400 python tools/hf_flow_check.py --layers 32 --batches 64 --qformat int4_awq > $O/hf_awq.json 2> $O/hf_awq.err
cat $O/gpu_tests.txt; cat $O/gemm_table.md; tail -c 1500 $O/awq_auto.json; tail -c 1500 $O/hf_awq.json
