#!/bin/bash
# (gpurun call, end of round 4) rocprofv3 kernel-trace + PMC of the default workload at HEAD and a bench line of the same box right after
set -u
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
bash tools/profile_bench.sh r05_fp8 --workload fp8
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/prof/r05_fp8_bench_line.json 2> gpurun_out/prof/r05_fp8_bench.err
cut -c1-900 gpurun_out/prof/r05_fp8_bench_line.json
head -12 gpurun_out/prof/r05_fp8_summary.md
find gpurun_out/prof -name '*.csv' -size +2M -delete 2>/dev/null
du -sh gpurun_out/prof
