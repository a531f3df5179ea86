#!/bin/bash
# bench.py N = 2 control flow (two ranks on one GPU, gloo debug mode) with the AWQ extra: once normally, once with a 1 s
# watchdog that must fire and still deliver the line.
set -u
O=gpurun_out/r03zq; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
MOQ_BENCH_DEBUG_ONE_GPU=1 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29513 \
  bench.py --gpus 2 --steps 5 --warmup 2 --layers 4 --awq-layers 2 --awq-batches 8 > $O/n2_awq_line.json 2> $O/n2_awq.err
echo "awq rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03zq/n2_awq_line.json")); e=d["extra"]; print(d["value"], d["scaling"], d["cpu_baseline"]["value"], e["awq_wallclock_s"], list(e["awq"])[:4], e["strong_scaling"]["value"])
PY
tail -2 $O/n2_awq.err
MOQ_BENCH_DEBUG_ONE_GPU=1 timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29514 \
  bench.py --gpus 2 --steps 5 --warmup 2 --layers 4 --awq-layers 8 --awq-batches 16 --awq-watchdog-s 1 --no-cpu-baseline > $O/n2_watchdog_line.json 2> $O/n2_watchdog.err
echo "watchdog rc=$?"; python - <<'PY'
import json
d=json.load(open("gpurun_out/r03zq/n2_watchdog_line.json")); print(d["value"], d["extra"]["awq"])
PY
tail -2 $O/n2_watchdog.err
