#!/bin/bash
# The driver's N = 1, 2, 4, 8 scaling run of bench.py, command line for command line (tools/scale_n.py has the details).
# On a multi-GPU node:   bash tools/run/scale_n.sh                      -> gpurun_out/SCALE_local.json
# On ONE GPU (control flow only, never a measurement):  MOQ_BENCH_DEBUG_ONE_GPU=1 bash tools/run/scale_n.sh 1,2 "--layers 2 --no-extra --no-cpu-baseline"
set -u
cd "$(dirname "$0")/../.."
python3 tools/scale_n.py --gpus "${1:-1,2,4,8}" --steps "${STEPS:-20}" --warmup "${WARMUP:-5}" --extra-args "${2:-}" --out gpurun_out/SCALE_local.json
