#!/bin/bash
set -u
O=gpurun_out/r03o; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 600 python -m pytest tests/test_gpu_sparsegpt.py tests/test_gpu_plugin_adapters.py -m gpu -q 2>&1 | tail -8 ) > $O/gpu_tests.txt
timeout 300 python tools/sgpt_bench.py > $O/sgpt_table.md 2> $O/sgpt.err
cat $O/gpu_tests.txt; tail -5 $O/sgpt_table.md
