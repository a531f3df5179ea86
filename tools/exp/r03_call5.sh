#!/bin/bash
# round 3, GPU call 5: SparseGPT trailing kernel with batched loads; N = 2 control flow of bench.py on one GPU (gloo debug mode)
set -u
O=gpurun_out/r03e; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_sparsegpt.py tests/test_gpu_input_quant.py tests/test_gpu_host.py -m gpu -q 2>&1 | tail -8 ) > $O/gpu_tests.txt
timeout 300 python tools/sgpt_bench.py > $O/sgpt_table.md 2> $O/sgpt.err
MOQ_BENCH_DEBUG_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 1 --awq-layers 2 --awq-batches 4 > $O/bench_n2_debug.json 2> $O/bench_n2_debug.err
timeout 300 python tools/hf_flow_check.py --layers 32 --batches 64 --qformat fp8 --note "r03e warm plain loop" >> $O/hf.jsonl 2> $O/hf1.err
ls -la $O
