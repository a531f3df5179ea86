#!/bin/bash
# full GPU tier + smoke at the round's last code state
set -u
O=gpurun_out/r03zs; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
timeout 150 python -m pytest tests -m gpu -x -q > $O/gpu_suite.txt 2>&1
tail -5 $O/gpu_suite.txt > $O/gpu_suite_tail.txt; cat $O/gpu_suite_tail.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
