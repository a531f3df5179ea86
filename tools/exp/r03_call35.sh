#!/bin/bash
# histogram at flow sizes: the table kernel (default for 16-bit inputs) against the arithmetic kernel, kernel-only durations
set -u
O=$PWD/gpurun_out/r03zk; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
ROOT=$PWD; export TMPDIR=/tmp; cd /tmp
export MOQ_LIB_PATH=$ROOT/model-optimizer_amd/csrc/libmoquant_exp.so
for rows in 512 2048 4096 7168 16384; do for h in 1 0; do
  MOQ_TUNE_HIST=$h timeout 120 rocprofv3 --kernel-trace --stats -f csv -d $O/h${h}_$rows -o t -- python $ROOT/tools/exp/hist_one.py $rows > $O/h${h}_$rows.log 2>&1
  f=$(find $O/h${h}_$rows -name '*kernel_stats.csv' | head -1)
  grep "hist_kernel\|input_quant_kernel<2, 0, false, false, true" $f | cut -c1-170 | sed "s/^/rows=$rows table=$h /" >> $O/summary.txt
done; done
cd $ROOT
find $O -type f ! -name '*.txt' ! -name '*.log' -delete 2>/dev/null
cat $O/summary.txt
