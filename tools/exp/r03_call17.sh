#!/bin/bash
set -u
O=gpurun_out/r03q; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 600 python -m pytest tests/test_gpu_sparsegpt.py -m gpu -q 2>&1 | tail -3 ) > $O/gpu_tests.txt
timeout 200 python tools/exp/sgpt_trailing_probe.py > $O/probe.md 2>&1
timeout 300 python tools/sgpt_bench.py 2>/dev/null | tail -4 > $O/sgpt_table.md
cat $O/gpu_tests.txt $O/probe.md $O/sgpt_table.md
