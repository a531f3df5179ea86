#!/bin/bash
# (gpurun call 12 of round 6) the reference's own algorithm-level GPU tests (quantize, calib, real quantize, layerwise, export)
# with install(algorithms=True) on top of the kernel seams
set -u
O=gpurun_out/${1:-r06c12}; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python3 -m pytest tests/test_gpu_reference_live.py -m gpu -q --tb=short -k "algorithm_seam_installed" > $O/s7_ref_tests.log 2>&1
echo "s7 ref tests rc=$?"; grep "passed\|failed\|^E  \|^FAILED\|\[note\]" $O/s7_ref_tests.log | tail -14 | cut -c1-2400
cp gpurun_out/reference_own_gpu_tests_algorithm_seam*.txt $O/ 2>/dev/null
grep "^FAILED\|^ERROR" $O/reference_own_gpu_tests_algorithm_seam.txt | cut -c1-260 | head -40
