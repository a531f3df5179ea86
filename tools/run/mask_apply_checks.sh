#!/bin/bash
# (gpurun call of round 4) fused 2:4 mask + apply (+ abs-max): parity, the sparsify flows, the fp8-mask24 step
set -u
O=gpurun_out/r04m; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_chunk_bodies.py tests/test_gpu_parity.py tests/test_gpu_moe.py tests/test_gpu_sparsegpt.py tests/test_gpu_dist_nccl.py -x -q -k "mask or spars or dist or nccl" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log
python3 tools/hf_flow_check.py --layers 32 --batches 2 --qformat sparse_magnitude > $O/flow_sparse.json 2> $O/flow_sparse.err; echo "sparse rc=$?"; cat $O/flow_sparse.json
python3 bench.py --gpus 1 --steps 10 --warmup 3 --workload fp8-mask24 --model mixtral-8x7b --no-extra --no-cpu-baseline > $O/bench_fp8_mask24_mixtral.json 2> $O/b.err; echo "bench rc=$?"; cat $O/bench_fp8_mask24_mixtral.json | cut -c1-900
python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --awq-layers 0 > $O/bench_default_short.json 2> $O/b2.err; echo "bench2 rc=$?"; python3 -c "
import json; d=json.loads(open('$O/bench_default_short.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline']['frac'], d['extra'].get('fp8_mask24_step'), d['extra'].get('mask_2to4'))"
