#!/bin/bash
# round 3: record run at HEAD -- full GPU suite tail, smoke, default bench line
set -u
O=gpurun_out/r03l; mkdir -p $O
export TMPDIR=/tmp
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $O/gpu_tests.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/gpu_tests.txt; cat $O/smoke.txt | tail -1; head -c 400 $O/bench.json
