#!/bin/bash
# (gpurun call 18 of round 6) section C of the live tests after its rewrite: every reference GPU test file plain / + kernel seams /
# + algorithm seam, in one session
set -u
O=gpurun_out/${1:-r06c18}; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python3 -m pytest tests/test_gpu_reference_live.py -m gpu -q --tb=short -k "references_own" > $O/section_c.log 2>&1
echo "section C rc=$?"; grep "passed\|failed\|^E  \|^FAILED\|\[note\]" $O/section_c.log | tail -14 | cut -c1-3000
cp gpurun_out/reference_own_gpu_tests_*.txt $O/ 2>/dev/null
