#!/bin/bash
set -u
O=gpurun_out/r03m; mkdir -p $O
export TMPDIR=/tmp
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 600 python -m pytest tests/test_gpu_sparsegpt.py tests/test_gpu_gemm.py tests/test_gpu_fold_weight.py -m gpu -q 2>&1 | tail -8 ) > $O/gpu_tests.txt
MOQ_BENCH_DEBUG_ONE_GPU=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --awq-layers 2 --awq-batches 4 > $O/bench_n2.out 2> $O/bench_n2.err
echo "N2 stdout lines: $(wc -l < $O/bench_n2.out)" > $O/log.txt
cat $O/gpu_tests.txt $O/log.txt
