#!/bin/bash
# chunk-body tests; bench lines + rocprofv3 (trace + PMC) of the mask24 and mxfp4 workloads in one session
set -u
O=gpurun_out/r03y; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 600 python -m pytest tests/test_gpu_chunk_bodies.py -m gpu -q 2>&1 | grep "FAILED\|passed\|failed" | cut -c1-200 ) > $O/gpu_tests.txt
for wl in mask24 mxfp4; do
timeout 200 python bench.py --workload $wl --no-extra --no-hf > $O/bench_$wl.json 2> $O/bench_$wl.err
timeout 600 bash tools/profile_bench.sh r03_$wl --workload $wl > $O/prof_$wl.log 2>&1
done
mkdir -p $O/prof; cp gpurun_out/prof/*_summary.md $O/prof/ 2>/dev/null
cat $O/gpu_tests.txt
python - <<'PY'
import json
for wl in ('mask24','mxfp4'):
    d=json.loads(open(f'gpurun_out/r03y/bench_{wl}.json').read().strip().splitlines()[-1])
    print(wl, d['value'], d['unit'], d['ms_per_step'], d['roofline'])
PY
head -20 $O/prof/r03_mask24_summary.md; head -20 $O/prof/r03_mxfp4_summary.md
