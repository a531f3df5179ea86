#!/bin/bash
set -u
O=gpurun_out/r03n; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 300 python -m pytest tests/test_gpu_fold_weight.py -m gpu -q 2>&1 | tail -3 ) > $O/gpu_tests.txt
for G in 8 4 16 2 32; do MOQ_TUNE_GEMM_GROUP=$G timeout 120 python tools/gemm_bench.py 2>&1 | grep "^| 8b\|^| sq" | cut -d'|' -f2,4,7 > $O/group$G.txt; done
cat $O/gpu_tests.txt; for G in 8 4 16 2 32; do echo "== group $G"; cat $O/group$G.txt; done
