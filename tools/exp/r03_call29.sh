#!/bin/bash
# bench default (in place) vs --out-of-place, twice each in fresh processes; N = 2 control flow
set -u
O=gpurun_out/r03zc; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
for i in 1 2; do
timeout 300 python bench.py --no-hf --awq-layers 0 --no-cpu-baseline > $O/inplace_$i.json 2> $O/inplace_$i.err
timeout 300 python bench.py --out-of-place --no-hf --awq-layers 0 --no-cpu-baseline > $O/oop_$i.json 2> $O/oop_$i.err
done
MOQ_BENCH_DEBUG_ONE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 1 --awq-layers 2 --awq-batches 4 --no-hf > $O/bench_n2_debug.json 2> $O/bench_n2_debug.err
python - <<'PY'
import json
for f in ('inplace_1','oop_1','inplace_2','oop_2','bench_n2_debug'):
    try:
        d=json.loads(open(f'gpurun_out/r03zc/{f}.json').read().strip().splitlines()[-1])
        e=d.get('extra',{})
        print(f, d['n_gpus'], d['value'], d['ms_per_step'], 'frac', d['roofline']['frac'], 'copy', d['roofline'].get('node_copy_GBs'), '|', d['config']['workload'][-60:], '| oop', e.get('qdq_out_of_place'), '| 70b', (e.get('llama3_70b_int4g128_inplace') or {}).get('frac_of_8TBs'))
    except Exception as ex:
        print(f, 'ERR', ex); print(open(f'gpurun_out/r03zc/{f}.err').read()[-800:])
PY
