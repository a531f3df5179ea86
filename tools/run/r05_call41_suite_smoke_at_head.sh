#!/bin/bash
# (gpurun call 41 of round 5, the last one: 7.7 GPU-minutes left) HEAD after the CPU-side fuzz finds: the whole GPU suite in the
# driver's serial form (the two new reference-live cases of per-layer overrides among them) and smoke()
set -u
O=gpurun_out/${1:-r05n}; mkdir -p $O
( time timeout 330 python3 -m pytest tests -m gpu -q --tb=short > $O/gpu_suite.log 2>&1 ) 2> $O/suite_time.txt
echo "suite rc=$? $(grep real $O/suite_time.txt)"; grep "passed\|failed\|^E  \|^FAILED" $O/gpu_suite.log | tail -12 | cut -c1-250
grep -h "INT8 per-channel MLP\|FP8 attention\|skipped" $O/gpu_suite.log | tail -5 | cut -c1-300
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
