#!/bin/bash
# (gpurun call of round 4) the GPU suite + the kernel table at the release library after a cold-kernel batch; the transpose
# with its 64 x 64 and 128 x 64 tiles side by side (experiment library)
set -u
O=gpurun_out/${1:-r04u}; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q -n 2 > $O/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -6 $O/gpu_suite.log
timeout 600 python3 tools/kbench.py > $O/kbench_release.md 2> $O/kbench_release.err
echo "kbench release rc=$?"
grep -E "pack|unpack|scale_cols|awq|mask_2to4|INT4|MXFP|two-level|transpose|row_hist" $O/kbench_release.md
for T in 0 1; do
  MOQ_LIB_PATH=$(pwd)/model-optimizer_amd/csrc/libmoquant_exp.so MOQ_TUNE_TRANSPOSE_TALL=$T timeout 600 python3 tools/kbench.py 2>/dev/null | grep -E "transpose" | sed "s/^/tall=$T /"
done
