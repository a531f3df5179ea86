#!/bin/bash
# fused input-quantizer kernel: pre_quant_scale vectors requested up front, no per-packet modulo -- parity + A/B
set -u
O=gpurun_out/r03u; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
( timeout 900 python -m pytest tests/test_gpu_input_quant.py tests/test_gpu_host.py tests/test_gpu_kv_cache.py tests/test_gpu_reference_style.py -m gpu -q -x 2>&1 | tail -4 ) > $O/gpu_tests.txt
for k in moq_input_quant; do
MOQ_LIB_PATH=$PWD/tools/exp/bin/libmoquant_prev.so timeout 200 python tools/kbench.py $k 2>/dev/null | grep "^| moq" | sed 's/^/prev /' >> $O/ktable.txt
timeout 200 python tools/kbench.py $k 2>/dev/null | grep "^| moq" | sed 's/^/new  /' >> $O/ktable.txt
done
cat $O/gpu_tests.txt; cut -c1-150 $O/ktable.txt
