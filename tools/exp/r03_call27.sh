#!/bin/bash
# histogram / abs-max + histogram at flow sizes: kernel-only durations (rocprofv3)
set -u
O=$PWD/gpurun_out/r03za; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
ROOT=$PWD; export TMPDIR=/tmp; cd /tmp
for rows in 2048 4096 7168 16384; do
  timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $O/h_$rows -o t -- python $ROOT/tools/exp/hist_one.py $rows > $O/h_$rows.log 2>&1
  f=$(find $O/h_$rows -name '*kernel_stats.csv' | head -1)
  grep "input_quant_kernel" $f | cut -c1-220 | sed "s/^/rows=$rows /" >> $O/summary.txt
done
cd $ROOT
find $O -type f ! -name '*.txt' ! -name '*.log' -delete 2>/dev/null
cat $O/summary.txt; tail -3 $O/h_2048.log
