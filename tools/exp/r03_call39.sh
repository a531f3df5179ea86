#!/bin/bash
# N > 1 control flow of bench.py after the weak-scaling switch: two ranks on ONE GPU through the gloo debug mode (never a
# measurement), the strong option, and the N > 1 code under RCCL with a world of one rank.
set -u
O=gpurun_out/r03zp; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
MOQ_BENCH_DEBUG_ONE_GPU=1 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  bench.py --gpus 2 --steps 5 --warmup 2 --awq-layers 0 --no-cpu-baseline > $O/n2_weak_debug_line.json 2> $O/n2_weak.err
echo "weak rc=$?"; cut -c1-1500 $O/n2_weak_debug_line.json; tail -3 $O/n2_weak.err
MOQ_BENCH_DEBUG_ONE_GPU=1 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 \
  bench.py --gpus 2 --steps 5 --warmup 2 --awq-layers 0 --no-cpu-baseline --scaling strong > $O/n2_strong_debug_line.json 2> $O/n2_strong.err
echo "strong rc=$?"; cut -c1-700 $O/n2_strong_debug_line.json; tail -3 $O/n2_strong.err
MOQ_FORCE_DIST=1 timeout 200 python bench.py --steps 5 --warmup 2 --awq-layers 0 --no-hf --no-cpu-baseline --workload int4g128 --layers 8 > $O/force_dist_int4_line.json 2> $O/force_dist.err
echo "force rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03zp/force_dist_int4_line.json")); print(d["value"], d["scaling"], d.get("extra",{}).get("strong_scaling"))
PY
tail -2 $O/force_dist.err
