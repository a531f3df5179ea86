#!/bin/bash
# (gpurun call 4 of round 6) tiled fold kernels: parity, configs[4]'s step on all Llama-3-70B weights, the OPT fp16 diagnostic
# and tests after the resmooth mean follows the numerics mode
set -u
O=gpurun_out/${1:-r06c4}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -k "fold_composed or multi_tensor_mx" > $O/fold_parity.log 2>&1
echo "fold parity rc=$?"; tail -2 $O/fold_parity.log | cut -c1-300; grep "^E  " $O/fold_parity.log | head -5 | cut -c1-300
for i in 1 2; do
timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 3 --workload mxfp4-sq --model llama3-70b --no-extra --no-cpu-baseline > $O/sq$i.json 2> $O/sq$i.err
echo "mxfp4-sq $(python3 -c "import json,sys; d=json.loads(open('$O/sq$i.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'], d['roofline']['min_launch_ms'])")"
done
timeout 600 python3 bench.py --gpus 1 --steps 10 --warmup 3 --workload mxfp4 --model llama3-70b --no-extra --no-cpu-baseline > $O/mx_plain.json 2> $O/mx_plain.err
echo "plain mx $(python3 -c "import json,sys; d=json.loads(open('$O/mx_plain.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['frac'])")"
python3 tools/diag/opt_awq_device_diff.py opt 2>&1 | grep "===\|PRE\|CKPT\|logits" | cut -c1-200
timeout 900 python3 -m pytest tests/test_gpu_reference_live.py -m gpu -q --tb=short -k "torchs_order or eager_search" > $O/live.log 2>&1
echo "live rc=$?"; grep "passed\|failed\|^E  \|^FAILED\|\[note\] INT4-AWQ on the device" $O/live.log | tail -12 | cut -c1-900
