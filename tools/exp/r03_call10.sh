#!/bin/bash
set -u
O=gpurun_out/r03j; mkdir -p $O
( hostname; rocm-smi --showuniqueid --showbus 2>&1 | head -20 ) > $O/box.txt 2>&1
timeout 60 python -c "
import torch
x=torch.randn(1<<28,device='cuda'); print('torch only ok', (x*2).sum().item())
a=torch.randn(4096,4096,device='cuda',dtype=torch.bfloat16); print('matmul ok', (a@a).float().abs().mean().item())
" > $O/torch_only.txt 2>&1; echo "torch_only rc=$?" >> $O/log.txt
timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 > $O/parity.txt; echo "parity rc=$?" >> $O/log.txt
timeout 100 python tools/exp/gemm_pitch_probe.py > $O/probe.txt 2>&1; echo "probe rc=$?" >> $O/log.txt
cat $O/box.txt $O/torch_only.txt $O/parity.txt $O/log.txt; tail -5 $O/probe.txt
