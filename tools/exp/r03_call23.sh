#!/bin/bash
# new chunk-body tests; bench N > 1 control flow with the asynchronous amax exchange (gloo two ranks on one GPU, RCCL one rank)
set -u
O=gpurun_out/r03w; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 900 python -m pytest tests/test_gpu_chunk_bodies.py -m gpu -q 2>&1 | tail -8 ) > $O/gpu_tests.txt
MOQ_BENCH_DEBUG_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 5 --warmup 1 --awq-layers 2 --awq-batches 4 --no-hf > $O/bench_n2_debug.json 2> $O/bench_n2_debug.err
MOQ_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_force_dist.json 2> $O/bench_force_dist.err
timeout 300 python bench.py --steps 10 --warmup 2 --no-extra --no-cpu-baseline > $O/bench_plain.json 2> $O/bench_plain.err
cat $O/gpu_tests.txt
python - <<'PY'
import json
for f in ('bench_n2_debug','bench_force_dist','bench_plain'):
    try:
        txt=open(f'gpurun_out/r03w/{f}.json').read().strip().splitlines()
        d=json.loads(txt[-1]); print(f, 'lines', len(txt), d['n_gpus'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['config'].get('parallelism','')[:120])
    except Exception as e:
        print(f, 'ERR', e); print(open(f'gpurun_out/r03w/{f}.err').read()[-1500:])
PY
