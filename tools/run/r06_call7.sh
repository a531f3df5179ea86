#!/bin/bash
# (gpurun call 7 of round 6) device numerics now take BOTH rounded means of AWQ (activation, weight scale) by torch's expressions:
# the live-reference file (8B-width alphas / vectors), the host file (probation), drop-in INT4-AWQ rows
set -u
O=gpurun_out/${1:-r06c7}; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests/test_gpu_reference_live.py tests/test_gpu_host.py tests/test_gpu_awq_search.py -m gpu -q --tb=short -n 2 > $O/tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|^E  \|^FAILED\|\[note\] INT4-AWQ\|\[note\] reference INT4\|\[note\] reference W4A8" $O/tests.log | tail -24 | cut -c1-900
timeout 2400 python3 tools/dropin_bench.py --layers 4 --batches 16 --rows 8 --seq 512 --formats int4_awq --out $O/dropin_awq.json > $O/dropin.log 2> $O/dropin.err
echo "dropin rc=$?"; grep "^{\"int" $O/dropin.log | cut -c1-420
