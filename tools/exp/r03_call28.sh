#!/bin/bash
# record run: whole GPU suite, smoke, default bench line, every HF-topology flow at full size
set -u
O=gpurun_out/r03zm; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $O/gpu_suite_tail.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1
timeout 900 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err
for q in fp8 int8_sq mxfp4 mxfp4_sq w4a8_awq; do
  timeout 400 python tools/hf_flow_check.py --layers 32 --batches 64 --qformat $q --note r03zm >> $O/hf_flow_check.jsonl 2>> $O/hf_flow.err
done
cat $O/gpu_suite_tail.txt; tail -1 $O/smoke.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03zm/bench_default_line.json').read().strip().splitlines()[-1])
print(d['value'], d['unit'], d['ms_per_step'], d['roofline'])
e=d['extra']
for k in e:
    if isinstance(e[k], dict): print(k, {kk: vv for kk, vv in e[k].items() if not isinstance(vv, dict)})
    else: print(k, e[k])
for l in open('gpurun_out/r03zm/hf_flow_check.jsonl'):
    try: x=json.loads(l)
    except Exception: continue
    print(x.get('qformat'), 'plain', x.get('plain_forward_loop_s'), 'quantize', x.get('quantize_s'), 'fq fwd', x.get('fake_quant_forward_s'), 'export', x.get('export_state_dict_s'), 'tensors', x.get('exported_tensors'))
PY
