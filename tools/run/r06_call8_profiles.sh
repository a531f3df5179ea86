#!/bin/bash
# (gpurun call 8 of round 6) rocprofv3 kernel-trace + PMC (separate passes) of the default FP8 workload and of configs[4]'s fused
# step on Llama-3-70B, a bench line of the same box right after each; the updated host test file
set -u
rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
export TMPDIR=/tmp
bash tools/profile_bench.sh r06_fp8 --workload fp8
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-extra --no-cpu-baseline > gpurun_out/prof/r06_fp8_bench_line.json 2> gpurun_out/prof/r06_fp8_bench.err
cut -c1-700 gpurun_out/prof/r06_fp8_bench_line.json; echo
head -14 gpurun_out/prof/r06_fp8_summary.md
bash tools/profile_bench.sh r06_mxfp4-sq_llama3-70b --workload mxfp4-sq --model llama3-70b
python3 bench.py --gpus 1 --steps 10 --warmup 3 --workload mxfp4-sq --model llama3-70b --no-extra --no-cpu-baseline > gpurun_out/prof/r06_mxfp4-sq_llama3-70b_bench_line.json 2> gpurun_out/prof/r06_mxfp4-sq_bench.err
cut -c1-700 gpurun_out/prof/r06_mxfp4-sq_llama3-70b_bench_line.json; echo
head -14 gpurun_out/prof/r06_mxfp4-sq_llama3-70b_summary.md
find gpurun_out/prof -name '*.csv' -size +2M -delete 2>/dev/null
du -sh gpurun_out/prof
timeout 600 python3 -m pytest tests/test_gpu_host.py -m gpu -q --tb=short -k "statistics_launch" 2>&1 | tail -3
