#!/bin/bash
# the default command (FP8, in place): bench line, then rocprofv3 trace + PMC passes in the same session
set -u
O=gpurun_out/r03zj; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
timeout 300 python bench.py --no-hf --awq-layers 0 > $O/bench_line.json 2> $O/bench.err
timeout 900 bash tools/profile_bench.sh r03ip_fp8 --workload fp8 > $O/prof.log 2>&1
cp gpurun_out/prof/r03ip_fp8_summary.md gpurun_out/prof/r03ip_fp8_pmc.json $O/ 2>/dev/null
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03zj/bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'])
PY
head -12 $O/r03ip_fp8_summary.md; sed -n '/## PMC/,$p' $O/r03ip_fp8_summary.md | head -12
