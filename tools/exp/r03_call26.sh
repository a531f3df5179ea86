#!/bin/bash
# running abs-max at flow sizes: kernel-only durations (rocprofv3) of the previous library and the new one
set -u
O=$PWD/gpurun_out/r03z; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_input_quant.py tests/test_gpu_host.py tests/test_gpu_streams.py -m gpu -q 2>&1 | grep "FAILED\|passed\|failed" | cut -c1-200 ) > $O/gpu_tests.txt
ROOT=$PWD; export TMPDIR=/tmp; cd /tmp
for rows in 512 2048 4096 7168 16384; do
for lib in prev new; do
  if [ $lib = prev ]; then export MOQ_LIB_PATH=$ROOT/tools/exp/bin/libmoquant_prev.so; else unset MOQ_LIB_PATH; fi
  timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $O/t_${lib}_$rows -o t -- python $ROOT/tools/exp/amax_sweep.py $rows > $O/t_${lib}_$rows.log 2>&1
  f=$(find $O/t_${lib}_$rows -name '*kernel_stats.csv' | head -1)
  echo "$lib rows=$rows $(grep amax_kernel $f | head -1 | cut -c1-200)" >> $O/summary.txt
done; done
unset MOQ_LIB_PATH; cd $ROOT
find $O -type f ! -name '*.txt' ! -name '*.log' -delete 2>/dev/null
cat $O/gpu_tests.txt $O/summary.txt
