#!/bin/bash
# (gpurun call of round 4) EIGHT ranks on one GPU through the gloo debug mode (never a measurement): the 8-way deal, the amax
# bucket, the weak leg, the 8-rank AWQ flow (Gram matrices dealt over 8 owners) and rank 0's reporting, on reduced layers
set -u
O=gpurun_out/r04r; mkdir -p $O
export MOQ_BENCH_DEBUG_ONE_GPU=1
for wl in int4g128 mxfp4-sq; do
  timeout 1200 python3 bench.py --gpus 8 --steps 2 --warmup 1 --workload $wl --layers 2 --awq-layers 1 --awq-batches 8 > $O/n8_$wl.json 2> $O/n8_$wl.err
  echo "$wl rc=$?"; python3 -c "
import json
d=json.loads(open('$O/n8_$wl.json').read().strip().splitlines()[-1])
e=d.get('extra',{})
print(d['metric'], d['value'], d['scaling'], d['n_gpus'], d['config']['parallelism'][:110], '| per_rank', len(d['roofline'].get('per_rank',[])), '| weak:', (e.get('weak_scaling') or {}).get('value'), '| awq:', e.get('awq_wallclock_s'), (e.get('awq') or {}).get('best_alpha_hist'), '| cpu:', (d.get('cpu_baseline') or {}).get('value'))
"
done
