#!/bin/bash
# (gpurun call 7 of round 5) the MSE block sweep with four packets per lane; kernel-ONLY durations of the flow-sized calibration
# kernels (rocprofv3 -f csv; call 6 wrote the default rocpd database and no CSV); kernel totals of the FP8 W + A + KV pass
# with / without the per-layer statistics launch
set -u
O=gpurun_out/r05c7; mkdir -p $O
ROOT=$(pwd); export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_mse.py tests/test_gpu_host.py -m gpu -q -n 2 --tb=short 2>&1 | grep -v "Warning\|warnings.warn\|^  " | tail -12 | tee $O/mse_host_tests_tail.txt
python3 tools/kbench.py "mse_sweep" 2>&1 | grep -v Warning | tee $O/kbench_mse.md
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$O/prof_small -o small -- python3 $ROOT/tools/kbench.py "67 MB,columns,col_abs,awq_weight_scale,row_hist,moq_hist_abs 2048 bins" > $ROOT/$O/prof_small.log 2>&1
cd $ROOT
python3 tools/kstats_md.py $O/prof_small | tee $O/small_kernels_kernel_only.md
for mode in off auto; do
  cd /tmp
  rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$O/prof_flow_$mode -o flow -- python3 $ROOT/tools/hf_flow_check.py --layers 32 --batches 64 --qformat fp8 --defer-stats $mode > $ROOT/$O/flow_prof_$mode.json 2> $ROOT/$O/flow_prof_$mode.err
  cd $ROOT
  tail -1 $O/flow_prof_$mode.json | cut -c1-200
  python3 tools/kstats_all_md.py $O/prof_flow_$mode 45 > $O/flow_kernels_$mode.md; head -52 $O/flow_kernels_$mode.md | cut -c1-190
done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete; du -sh $O
