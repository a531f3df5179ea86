#!/bin/bash
# bench lines of the other BASELINE configurations with the in-place default (8B: int8, int4g128, mxfp4, mask24; 70B: int4g128, mxfp4-sq)
set -u
O=gpurun_out/r03zn; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
for wl in int8 int4g128 mxfp4 mask24; do
  timeout 200 python bench.py --workload $wl --no-extra --no-cpu-baseline > $O/line_8b_$wl.json 2> $O/err_8b_$wl.txt
done
timeout 300 python bench.py --model llama3-70b --workload int4g128 --no-extra --no-cpu-baseline --steps 5 --warmup 1 > $O/line_70b_int4g128.json 2> $O/err_70b_int4g128.txt
timeout 300 python bench.py --model llama3-70b --workload mxfp4-sq --no-extra --no-cpu-baseline --steps 5 --warmup 1 > $O/line_70b_mxfp4-sq.json 2> $O/err_70b_mxfp4sq.txt
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r03zn/line_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], d['value'], d['unit'], 'ms/step', d['ms_per_step'], '|', r['kernel'], 'frac', r['frac'], 'traffic', r.get('traffic'))
    except Exception as e: print(f, 'ERR', e)
PY
