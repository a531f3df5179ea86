#!/bin/bash
# round 3, GPU call 8: bench.py's N > 1 path under RCCL with one rank (MOQ_FORCE_DIST=1); PMC of the release error GEMM; GEMM table
set -u
O=gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
MOQ_FORCE_DIST=1 timeout 600 python bench.py --steps 5 --warmup 1 --awq-layers 4 --awq-batches 8 --no-hf > $O/bench_force_dist.json 2> $O/bench_force_dist.err
timeout 200 python tools/gemm_bench.py > $O/gemm_bench.md 2>&1
timeout 400 bash tools/exp/gemm_pmc.sh r03 10 > $O/gemm_pmc.log 2>&1
cp gpurun_out/prof/r03_geo10_pmc.md $O/ 2>/dev/null
rm -rf gpurun_out/prof
tail -c 600 $O/bench_force_dist.err; ls -la $O
