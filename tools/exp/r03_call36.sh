#!/bin/bash
# the other calibration algorithms and the sparsity modes at real layer shapes (4 decoder layers, 16 x 4096 tokens): wall-clock
set -u
O=gpurun_out/r03zl; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
for q in int8_mse fp8_mse int4_mse int4_awq_clip int4_awq_full sparse_magnitude sparsegpt; do
  timeout 300 python tools/hf_flow_check.py --layers 4 --batches 16 --qformat $q --note r03zl >> $O/flows.jsonl 2>> $O/flows.err || echo "{\"qformat\": \"$q\", \"failed\": true}" >> $O/flows.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r03zl/flows.jsonl'):
    try: d=json.loads(l)
    except Exception: print('?', l[:200]); continue
    print({k: d[k] for k in d if k in ('qformat','failed','plain_forward_loop_s','quantize_s','sparsify_s','export_state_dict_s','masked_linears','kept_fraction','enabled_quantizers')})
PY
tail -5 $O/flows.err
