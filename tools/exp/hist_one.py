"""One size of tools/exp/hist_sweep.py, for `rocprofv3 --kernel-trace --stats` (kernel-only durations).  argv: rows"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load()
ops = moa.ops
dev = "cuda:0"
rows, cols = int(sys.argv[1]), 8192
g = torch.Generator(device=dev).manual_seed(0)
ch = torch.exp(torch.randn(cols, generator=g, device=dev))
ch[:8] *= 30
x = (torch.randn(rows, cols, generator=g, device=dev) * ch).to(torch.bfloat16)
amax = float(ops.reduce_amax(x))
counts = torch.zeros(2048, dtype=torch.int64, device=dev)
run = torch.zeros(1, dtype=torch.float32, device=dev)
for _ in range(100):
    ops.hist_abs(x, 2048, amax, counts=counts)
for _ in range(100):  # abs-max + histogram in one read (HistogramCalibrator with a known range)
    ops.input_quant(x, None, amax_running=run, hist_counts=counts, hist_max_edge=amax)
torch.cuda.synchronize()
print("rows", rows, int(counts.sum()))
