#!/bin/bash
# branch-free MX rounding, two-level MX on the chunk skeleton, 2:4 mask without table loads: parity, kernel table A/B, bench A/B
set -u
O=gpurun_out/r03t; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_mxfp8.py tests/test_gpu_qtensor.py tests/test_gpu_export.py tests/test_gpu_fuzz.py tests/test_gpu_host.py tests/test_gpu_sparsegpt.py tests/test_gpu_reference_style.py tests/test_gpu_fold_weight.py -m gpu -q -x 2>&1 | tail -4 ) > $O/gpu_tests.txt
for k in mask_2to4 mx_fused; do
MOQ_LIB_PATH=$PWD/tools/exp/bin/libmoquant_prev.so timeout 200 python tools/kbench.py $k 2>/dev/null | grep "^| moq" | sed 's/^/prev /' >> $O/ktable.txt
timeout 200 python tools/kbench.py $k 2>/dev/null | grep "^| moq" | sed 's/^/new  /' >> $O/ktable.txt
done
for wl in mask24 mxfp4; do
MOQ_LIB_PATH=$PWD/tools/exp/bin/libmoquant_prev.so timeout 200 python bench.py --workload $wl --no-extra --no-cpu-baseline --no-hf 2>/dev/null | tail -1 | sed 's/^/prev /' >> $O/bench.txt
timeout 200 python bench.py --workload $wl --no-extra --no-cpu-baseline --no-hf 2>/dev/null | tail -1 | sed 's/^/new  /' >> $O/bench.txt
done
cat $O/gpu_tests.txt; cut -c1-150 $O/ktable.txt; python - <<'PY'
import json
for l in open('gpurun_out/r03t/bench.txt'):
    tag, js = l[:5], l[5:]
    try:
        d=json.loads(js); print(tag, d['config']['workload'][:40], d['value'], d['unit'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'])
    except Exception as e: print(tag, 'ERR', js[:200])
PY
