#!/bin/bash
# round 4, call 4: dense single-window variants (adjacent chunks per workgroup, read phase / write phase) on the same sets
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/r04d; mkdir -p $O
MOQ_LIB_PATH=$ROOT/model-optimizer_amd/csrc/libmoquant_exp.so python3 tools/pool_placement.py --sweep --quick --sets 4 --out $O/sweep.json > $O/sweep.log 2> $O/sweep.err
echo "sweep rc=$?"; cat $O/sweep.log; tail -3 $O/sweep.err
