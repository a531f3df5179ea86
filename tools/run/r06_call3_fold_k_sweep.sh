#!/bin/bash
# (gpurun call 3 of round 6) fused fold + MX: parity again, then chunks-per-workgroup sweep (scale registers reused across
# consecutive chunks) on all Llama-3-70B weights with the experiment library; the attribution test of the OPT fp16 difference
set -u
O=gpurun_out/${1:-r06c3}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "fold_composed or multi_tensor_mx" > $O/fold_parity.log 2>&1
echo "fold parity rc=$?"; tail -2 $O/fold_parity.log | cut -c1-300
timeout 900 python3 -m pytest tests/test_gpu_reference_live.py -m gpu -q --tb=short -k "torchs_order or eager_search" > $O/live.log 2>&1
echo "live rc=$?"; grep "passed\|failed\|^E  \|^FAILED\|\[note\] INT4-AWQ on the device" $O/live.log | tail -12 | cut -c1-1500
for K in 1 2 4 8 16; do
  MOQ_LIB_PATH=$(pwd)/model-optimizer_amd/csrc/libmoquant_exp.so MOQ_TUNE_FOLD_K=$K timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 3 --workload mxfp4-sq --model llama3-70b --no-extra --no-cpu-baseline > $O/k$K.json 2> $O/k$K.err
  echo "K=$K $(python3 -c "import json,sys; d=json.loads(open('$O/k$K.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['min_launch_ms'])")"
done
timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 3 --workload mxfp4 --model llama3-70b --no-extra --no-cpu-baseline > $O/mx_plain.json 2> $O/mx_plain.err
echo "plain mx $(python3 -c "import json,sys; d=json.loads(open('$O/mx_plain.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])")"
