#!/bin/bash
# (gpurun call of round 4) the GPU suite + the kernel table at the release library after the cold-kernel batch
set -u
O=gpurun_out/r04u; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q -n 2 > $O/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -6 $O/gpu_suite.log
timeout 600 python3 tools/kbench.py > $O/kbench_release.md 2> $O/kbench_release.err
echo "kbench release rc=$?"
grep -E "pack|unpack|scale_cols|awq|mask_2to4|INT4|MXFP|two-level" $O/kbench_release.md
