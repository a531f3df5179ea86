#!/bin/bash
# (gpurun call 2 of round 6) the fused fold + MX kernels (parity, then configs[4]'s step on Llama-3-70B), section A' again,
# the INT4-AWQ device test that now compares every searched linear's alpha (OPT fp16: VERDICT r5 weak #1), the drop-in timing at
# four layers / 64k calibration tokens
set -u
O=gpurun_out/${1:-r06c2}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "fold_composed or multi_tensor_mx" > $O/fold_parity.log 2>&1
echo "fold parity rc=$?"; tail -3 $O/fold_parity.log | cut -c1-300
timeout 1200 python3 -m pytest tests/test_gpu_reference_live.py -m gpu -q --tb=short -k "algorithm_seam or eager_search" > $O/live.log 2>&1
echo "live rc=$?"; grep "passed\|failed\|^E  \|^FAILED\|\[note\] INT4-AWQ on the device\|MXFP4" $O/live.log | tail -20 | cut -c1-1800
timeout 900 python3 bench.py --gpus 1 --steps 10 --warmup 3 --workload mxfp4-sq --model llama3-70b --no-extra > $O/bench_mxfp4sq_70b.json 2> $O/bench_mxfp4sq_70b.err
echo "mxfp4-sq 70b rc=$?"; tail -1 $O/bench_mxfp4sq_70b.json | cut -c1-900; tail -3 $O/bench_mxfp4sq_70b.err | cut -c1-300
timeout 900 python3 bench.py --gpus 1 --steps 10 --warmup 3 --workload mxfp4 --model llama3-70b --no-extra > $O/bench_mxfp4_70b.json 2> $O/bench_mxfp4_70b.err
echo "mxfp4 70b rc=$?"; tail -1 $O/bench_mxfp4_70b.json | cut -c1-600
timeout 3000 python3 tools/dropin_bench.py --layers 4 --batches 16 --rows 8 --seq 512 --out $O/dropin.json > $O/dropin.log 2> $O/dropin.err
echo "dropin rc=$?"; grep "^{\"fp8\|^{\"int\|^{\"mx" $O/dropin.log | cut -c1-700; tail -3 $O/dropin.err | cut -c1-300
