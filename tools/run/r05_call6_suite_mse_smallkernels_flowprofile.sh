#!/bin/bash
# (gpurun call 6 of round 5) whole GPU suite at HEAD (packed MSE sweep, FP8 export in numerics mode "device", ordered finalize
# sums with 32 loads in flight); the MSE sweep against its VALU roofline; kernel-ONLY durations of the flow-sized calibration
# kernels (rocprofv3); where the FP8 W + A + KV pass spends its 5 % (kernel totals with / without deferral)
set -u
O=gpurun_out/r05c6; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 1500 python3 -m pytest tests -m gpu -q -n 2 --tb=short > $O/gpu_suite.log 2>&1
echo "suite rc=$?"; grep -v "Warning\|warnings.warn\|^  " $O/gpu_suite.log | tail -40
python3 tools/kbench.py "mse_sweep,67 MB,columns,col_abs,awq_weight_scale,row_hist" 2>&1 | grep -v Warning | tee $O/kbench_small_and_mse.md
rocprofv3 --kernel-trace --stats -d $O/prof_small -- python3 tools/kbench.py "67 MB,columns,col_abs,awq_weight_scale,row_hist,moq_hist_abs 2048 bins" > $O/prof_small.log 2>&1
python3 tools/kstats_md.py $O/prof_small | tee $O/small_kernels_kernel_only.md
for mode in off auto; do
  rocprofv3 --kernel-trace --stats -d $O/prof_flow_$mode -- python3 tools/hf_flow_check.py --layers 32 --batches 64 --qformat fp8 --defer-stats $mode > $O/flow_prof_$mode.json 2> $O/flow_prof_$mode.err
  tail -1 $O/flow_prof_$mode.json | cut -c1-400
  python3 tools/kstats_all_md.py $O/prof_flow_$mode 40 > $O/flow_kernels_$mode.md; head -50 $O/flow_kernels_$mode.md | cut -c1-200
done
find $O -name "*.csv" -size +2M -delete; find $O -name "*.db" -delete; du -sh $O
