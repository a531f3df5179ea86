#!/bin/bash
# (gpurun call 26 of round 6) the seeded differential fuzzers on the device against the staged reference, fresh seed (6), after
# the round's kernel changes (pattern-counter histogram, MXFP4 hardware converters, device numerics)
# HISTORY: the first run of this script lost the GPU box (twice): calib_fuzz seed 6 case 27 grew a histogram to 8e5 bins and the
# then-unbounded host pass of the histogram MSE search asked for 2.7 TB (fixed in calib._compute_amax_mse; the fuzzer now caps
# the growth).  The other four fuzzers were then run one call each (r06c27 / r06c28 / r06c35).
set -u
O=gpurun_out/${1:-r06c26}; mkdir -p $O
export TMPDIR=/tmp
run() { name=$1; shift; ( time timeout 900 python3 "$@" > $O/$name.log 2>&1 ) 2> $O/$name.time; echo "== $name rc=$? $(grep real $O/$name.time)"; tail -3 $O/$name.log | cut -c1-400; }
run calib_fuzz tools/calib_fuzz.py 40 6
run ops_fuzz tools/ops_fuzz.py 200 6
run quantizer_fuzz tools/quantizer_fuzz.py 400 6
run sparsity_fuzz tools/sparsity_fuzz.py 150 6
run flow_fuzz tools/flow_fuzz.py 120 6
