#!/bin/bash
# (gpurun call 11 of round 6) the reference's own test_quantize_cuda.py with install(algorithms=True) on top of the kernel seams
# (nothing that passes with the kernel seams may fail), then the 4-layer drop-in table again at HEAD (the committed one predates
# the device-numerics weight scale: it still said 33 / 56)
set -u
O=gpurun_out/${1:-r06c11}; mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python3 -m pytest tests/test_gpu_reference_live.py -m gpu -q --tb=short -k "algorithm_seam_installed" > $O/s7_ref_tests.log 2>&1
echo "s7 ref tests rc=$?"; grep "passed\|failed\|^E  \|^FAILED\|\[note\]" $O/s7_ref_tests.log | tail -14 | cut -c1-1600
timeout 3000 python3 tools/dropin_bench.py --layers 4 --batches 16 --rows 8 --seq 512 --out $O/dropin.json > $O/dropin.log 2> $O/dropin.err
echo "dropin rc=$?"; grep "^{\"fp8\|^{\"int\|^{\"mx" $O/dropin.log | cut -c1-700; tail -3 $O/dropin.err | cut -c1-300
