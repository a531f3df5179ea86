#!/bin/bash
# (gpurun call of round 4) chunks per workgroup x occupancy cap for the four whole-model read+write kernels and the read-only abs-max
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/r04h; mkdir -p $O
MOQ_LIB_PATH=$ROOT/model-optimizer_amd/csrc/libmoquant_exp.so python3 tools/pool_placement.py --sweep --sets 3 --out $O/sweep.json > $O/sweep.log 2> $O/sweep.err
echo "sweep rc=$?"; cat $O/sweep.log; tail -3 $O/sweep.err
# the release library on the same layout (no knobs): timing of every set
python3 tools/pool_placement.py --sets 3 --out $O/release_timing.json > $O/release_timing.log 2>&1
cat $O/release_timing.log | tail -12
