"""Fixed cost of the histogram kernel at 67 MB against the number of bins (flush = bins x workgroups global atomics)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load()
ops = moa.ops
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
ch = torch.exp(torch.randn(8192, generator=g, device=dev)); ch[:8] *= 30
for rows in (512, 4096):
    x = (torch.randn(rows, 8192, generator=g, device=dev) * ch).to(torch.bfloat16)
    amax = float(ops.reduce_amax(x))
    for bins in (64, 256, 1024, 2048, 8192):
        counts = torch.zeros(bins, dtype=torch.int64, device=dev)
        fn = lambda: ops.hist_abs(x, bins, amax, counts=counts)
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            fn()
        b.record(); torch.cuda.synchronize()
        print(f"{rows} x 8192 ({x.numel() * 2 / 1e6:.0f} MB) bins {bins:5d}: {a.elapsed_time(b) / 50 * 1e3:6.1f} us")
