#!/bin/bash
# (gpurun call of round 4) two-rank control flow of the N > 1 defaults on ONE GPU through the gloo debug mode (never a measurement):
# fp8 -> Mixtral-8x7B, int4g128 -> Llama-3-70B, mxfp4-sq -> Llama-3-70B; bare command (self-launch); reduced layers
set -u
O=gpurun_out/r04l; mkdir -p $O
export MOQ_BENCH_DEBUG_ONE_GPU=1
for wl in fp8 int4g128 mxfp4-sq mask24; do
  timeout 900 python3 bench.py --gpus 2 --steps 3 --warmup 1 --workload $wl --layers 2 --awq-layers 1 --awq-batches 2 > $O/n2_$wl.json 2> $O/n2_$wl.err
  echo "$wl rc=$?"; python3 -c "
import json,sys
d=json.loads(open('$O/n2_$wl.json').read().strip().splitlines()[-1])
e=d.get('extra',{})
print(d['metric'], d['value'], d['scaling'], d['n_gpus'], d['config']['model'], d['config']['baseline_config'], '| weak:', (e.get('weak_scaling') or {}).get('value'), '| awq:', e.get('awq_wallclock_s'), '| cpu:', (d.get('cpu_baseline') or {}).get('value'))
"
done
unset MOQ_BENCH_DEBUG_ONE_GPU
MOQ_FORCE_DIST=1 timeout 900 python3 bench.py --gpus 1 --steps 3 --warmup 1 --layers 4 --awq-layers 1 --awq-batches 2 --no-hf > $O/force_dist_n1.json 2> $O/force_dist_n1.err; echo "force-dist rc=$?"; tail -c 600 $O/force_dist_n1.json
