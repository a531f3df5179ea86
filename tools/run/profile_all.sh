#!/bin/bash
# (gpurun call of round 4) rocprofv3 kernel-trace + PMC (FETCH_SIZE / WRITE_SIZE in separate passes) of the bench command per workload,
# and a bench line of the same box right after (same session)
set -u
for wl in fp8 int4g128 mask24 mxfp4; do
  bash tools/profile_bench.sh r04_$wl --workload $wl
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/prof/r04_fp8_bench_line.json 2> gpurun_out/prof/r04_fp8_bench.err
cat gpurun_out/prof/r04_fp8_bench_line.json | cut -c1-1200
ls gpurun_out/prof | head -40; du -sh gpurun_out/prof
cat gpurun_out/prof/r04_fp8_summary.md | head -30
