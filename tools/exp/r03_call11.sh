#!/bin/bash
set -u
O=gpurun_out/r03k; mkdir -p $O
export TMPDIR=/tmp
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
MOQ_FORCE_DIST=1 timeout 300 python bench.py --steps 3 --warmup 1 --no-hf --awq-layers 2 --awq-batches 4 > $O/bench_force_dist.out 2> $O/bench_force_dist.err
echo "stdout lines: $(wc -l < $O/bench_force_dist.out)" > $O/log.txt; tail -c 200 $O/bench_force_dist.out >> $O/log.txt; echo >> $O/log.txt
timeout 200 python tools/gemm_bench.py > $O/gemm_bench.md 2>&1; echo "gemm_bench rc=$?" >> $O/log.txt
timeout 300 bash tools/exp/gemm_pmc.sh r03 10 > $O/gemm_pmc.log 2>&1; echo "pmc rc=$?" >> $O/log.txt
cp gpurun_out/prof/r03_geo10_pmc.md $O/ 2>/dev/null
rm -rf gpurun_out/prof
cat $O/log.txt
