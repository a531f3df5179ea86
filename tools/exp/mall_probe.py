"""Does the second pass over a weight hit the Infinity Cache?  Times the FP8 QDQ of one tensor cold (after a 2 GB
sweep of other memory) and warm (right after the abs-max pass over the same tensor), for sizes around the 256 MB MALL."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load()
ops = moa.ops
dev = "cuda:0"
big = torch.empty(1 << 30, dtype=torch.bfloat16, device=dev)  # 2 GB flusher


def ev():
    return torch.cuda.Event(enable_timing=True)


for mb in (16, 33, 67, 117, 200, 400):
    n = mb * (1 << 20) // 2
    x = torch.randn(n, device=dev).to(torch.bfloat16).view(-1, 4096)
    y = torch.empty_like(x)
    cold, warm, amax_t = [], [], []
    for _ in range(5):
        big.zero_()
        torch.cuda.synchronize()
        a, b = ev(), ev()
        amax = ops.reduce_amax(x)
        torch.cuda.synchronize()
        big.zero_()
        torch.cuda.synchronize()
        a.record(); y = ops.scaled_e4m3(x, amax); b.record()
        torch.cuda.synchronize()
        cold.append(a.elapsed_time(b))
        big.zero_()
        torch.cuda.synchronize()
        a, b, c = ev(), ev(), ev()
        a.record(); amax = ops.reduce_amax(x); b.record()
        y = ops.scaled_e4m3(x, amax); c.record()
        torch.cuda.synchronize()
        amax_t.append(a.elapsed_time(b)); warm.append(b.elapsed_time(c))
    f = lambda v: sorted(v)[len(v) // 2]
    gb = n * 2 / 1e9
    print(f"{mb:4d} MB: amax {f(amax_t)*1e3:7.1f} us ({gb/f(amax_t)*1e3:6.0f} GB/s)  QDQ cold {f(cold)*1e3:7.1f} us ({2*gb/f(cold)*1e3:6.0f} GB/s r+w)"
          f"  QDQ warm {f(warm)*1e3:7.1f} us ({2*gb/f(warm)*1e3:6.0f} GB/s r+w)")
