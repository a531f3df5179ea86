#!/bin/bash
# MODE 0 epilogue with the out_actual tile staged through the LDS: parity, GEMM table prev vs new, both AWQ flows
set -u
O=gpurun_out/r03zg; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_awq_search.py tests/test_gpu_host.py tests/test_gpu_export.py -m gpu -q 2>&1 | grep "FAILED\|passed\|failed" | cut -c1-200 ) > $O/gpu_tests.txt
MOQ_LIB_PATH=$PWD/tools/exp/bin/libmoquant_prev.so timeout 300 python tools/gemm_bench.py > $O/gemm_prev.md 2> $O/gemm_prev.err
timeout 300 python tools/gemm_bench.py > $O/gemm_new.md 2> $O/gemm_new.err
timeout 300 python tools/awq_bench.py --layers 32 --batches 64 --search auto > $O/awq_auto.json 2> $O/awq_auto.err
timeout 400 python tools/hf_flow_check.py --layers 32 --batches 64 --qformat int4_awq --note "r03zg staged out_actual tile" > $O/hf_awq.json 2> $O/hf_awq.err
cat $O/gpu_tests.txt; for g in prev new; do echo $g; grep "^| [0-9a-z]" $O/gemm_$g.md | cut -d'|' -f2,3,4,6,7,10; done
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03zg/awq_auto.json').read().strip().splitlines()[-1]); print('awq synthetic', d['value'], d['stages_s'], d.get('best_alpha_hist'))
d=json.loads(open('gpurun_out/r03zg/hf_awq.json').read().strip().splitlines()[-1]); print('awq hf', d['quantize_s'], d['awq_stats']['stages_s'], d['awq_best_alpha_hist'])
PY
