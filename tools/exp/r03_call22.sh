#!/bin/bash
# record run: whole GPU suite, INT8 A/B, default bench line (writes gpurun_out/r03v)
set -u
O=gpurun_out/r03v; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -12 ) > $O/gpu_suite_tail.txt
for wl in int8; do
MOQ_LIB_PATH=$PWD/tools/exp/bin/libmoquant_prev.so timeout 200 python bench.py --workload $wl --no-extra --no-cpu-baseline --no-hf 2>/dev/null | tail -1 | sed 's/^/prev /' >> $O/bench_ab.txt
timeout 200 python bench.py --workload $wl --no-extra --no-cpu-baseline --no-hf 2>/dev/null | tail -1 | sed 's/^/new  /' >> $O/bench_ab.txt
done
timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
cat $O/gpu_suite_tail.txt; tail -2 $O/smoke.txt
python - <<'PY'
import json
for l in open('gpurun_out/r03v/bench_ab.txt'):
    tag, js = l[:5], l[5:]
    try:
        d=json.loads(js); print(tag, d['config']['workload'][:40], d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'])
    except Exception as e: print(tag, 'ERR', js[:200])
d=json.loads(open('gpurun_out/r03v/bench_default_line.json').read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], d['roofline'])
e=d['extra']
for k in e:
    if isinstance(e[k], dict): print(k, {kk: vv for kk, vv in e[k].items() if not isinstance(vv, dict)})
    else: print(k, e[k])
print(d['cpu_baseline'])
PY
