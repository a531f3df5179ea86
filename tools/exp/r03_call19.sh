#!/bin/bash
# streaming packers / fold kernels with branch-free full chunks: parity, then the kernel table old library vs new
set -u
O=gpurun_out/r03s; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 900 python -m pytest tests/test_gpu_qtensor.py tests/test_gpu_export.py tests/test_gpu_fp8_2d.py tests/test_gpu_input_quant.py tests/test_gpu_fold_weight.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_host.py tests/test_gpu_calibrate_weights.py -m gpu -q -x 2>&1 | tail -5 ) > $O/gpu_tests.txt
MOQ_LIB_PATH=$PWD/tools/exp/bin/libmoquant_prev.so timeout 300 python tools/kbench.py > $O/ktable_prev.md 2> $O/ktable_prev.err
timeout 300 python tools/kbench.py > $O/ktable_new.md 2> $O/ktable_new.err
MOQ_LIB_PATH=$PWD/tools/exp/bin/libmoquant_prev.so timeout 200 python tools/exp/hist_sweep.py > $O/hist_prev.txt 2>&1
timeout 200 python tools/exp/hist_sweep.py > $O/hist_new.txt 2>&1
cat $O/gpu_tests.txt
python - <<'PY'
import re
def rd(p):
    d={}
    for l in open(p):
        c=[x.strip() for x in l.split('|')]
        if len(c)>4 and re.match(r'^[0-9.]+$', c[2] or 'x'): d[c[1][:70]]=(float(c[2]), float(c[4]))
    return d
a,b=rd('gpurun_out/r03s/ktable_prev.md'),rd('gpurun_out/r03s/ktable_new.md')
for k in a:
    if k in b: print(f"{k:72s} {a[k][0]:.3f} ms {a[k][1]:.3f} -> {b[k][0]:.3f} ms {b[k][1]:.3f}")
PY
paste $O/hist_prev.txt $O/hist_new.txt | tail -9
