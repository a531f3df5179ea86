#!/bin/bash
# rocprofv3 trace + PMC of the int8 and int4g128 workloads in place (completes the committed PMC traffic of the five configurations)
set -u
O=gpurun_out/r03zo; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
for wl in int8 int4g128; do
  timeout 400 bash tools/profile_bench.sh r03_$wl --workload $wl > $O/prof_$wl.log 2>&1
  cp gpurun_out/prof/r03_${wl}_summary.md gpurun_out/prof/r03_${wl}_pmc.json $O/ 2>/dev/null
  sed -n '/## PMC/,$p' $O/r03_${wl}_summary.md | grep "mt_" | cut -c1-200
done
