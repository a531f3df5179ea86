#!/bin/bash
# (gpurun call 1 of round 6) S7, the algorithm seam, on the device: section A' of the live-reference tests, then the four-row
# drop-in timing (reference eager / + kernel seams / + algorithm seam / this package's own quantize) on two Llama-3-8B-wide layers
set -u
O=gpurun_out/${1:-r06c1}; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_reference_live.py -m gpu -q --tb=short -k "algorithm_seam or through_installed_seams" > $O/aprime.log 2>&1
echo "A' rc=$?"; grep "passed\|failed\|^E  \|^FAILED\|\[note\]" $O/aprime.log | tail -30 | cut -c1-600
timeout 2400 python3 tools/dropin_bench.py --layers 2 --out $O/dropin.json > $O/dropin.log 2> $O/dropin.err
echo "dropin rc=$?"; grep -v "^{\"what\"" $O/dropin.log | cut -c1-700; tail -5 $O/dropin.err | cut -c1-300
