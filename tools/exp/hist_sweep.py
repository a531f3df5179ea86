"""Histogram collection (moq_hist_abs, 2048 bins, bf16 activations with outlier channels) against the input size."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load()
ops = moa.ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
print("| activation | ms | GB/s | frac of 8 TB/s |\n|---|---|---|---|")
for rows in (2048, 4096, 8192, 16384, 32768, 65536, 131072):
    cols = 8192
    ch = torch.exp(torch.randn(cols, generator=g, device=dev))
    ch[:8] *= 30
    x = (torch.randn(rows, cols, generator=g, device=dev) * ch).to(torch.bfloat16)
    amax = float(ops.reduce_amax(x))
    counts = torch.zeros(2048, dtype=torch.int64, device=dev)
    fn = lambda: ops.hist_abs(x, 2048, amax, counts=counts)
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    gb = x.numel() * 2 / 1e9
    print(f"| {rows} x {cols} ({gb * 1e3:.0f} MB) | {ms:.3f} | {gb / ms * 1e3:.0f} | {gb / ms * 1e3 / 8000:.3f} |")
    del x
