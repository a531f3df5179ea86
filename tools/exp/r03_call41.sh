#!/bin/bash
# GPU tier of the calibrator host changes (entropy totals, vectorised mse threshold search, calibrate_weights mse)
set -u
O=gpurun_out/r03zr; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
timeout 200 python -m pytest tests/test_gpu_host.py tests/test_gpu_calibrate_weights.py tests/test_gpu_reference_style.py -m gpu -x -q > $O/tests.txt 2>&1
tail -4 $O/tests.txt
timeout 60 python - > $O/mse_time.txt 2>&1 <<'PY'
import time, torch, sys
sys.path.insert(0, '.'); import _moa_import; moa = _moa_import.load()
from model_optimizer_amd import calib
x = torch.randn(8, 512, 4096, device='cuda', dtype=torch.bfloat16)
c = calib.HistogramCalibrator(8, None, False); c.collect(x); torch.cuda.synchronize()
for m in ("mse", "percentile", "entropy"):
    t = time.perf_counter(); a = c.compute_amax(m); torch.cuda.synchronize(); print(m, round((time.perf_counter() - t) * 1e3, 2), "ms", float(a))
w = torch.randn(4096, 4096, device='cuda', dtype=torch.bfloat16)
class L(torch.nn.Module):
    def __init__(s):
        super().__init__(); s.weight = torch.nn.Parameter(w); s.weight_quantizer = moa.TensorQuantizer(moa.QuantizerAttributeConfig(num_bits=8, axis=0))
l = L()
for m in ("mse", "percentile"):
    t = time.perf_counter(); calib.calibrate_weights(l, method=m); torch.cuda.synchronize(); print("calibrate_weights 4096x4096", m, round(time.perf_counter() - t, 3), "s")
PY
cat $O/mse_time.txt | tail -6
