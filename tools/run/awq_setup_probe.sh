#!/bin/bash
# (gpurun call of round 4) where do 5 s of awq_lite's "setup" stage come from on some boxes?  weight scales vs Gram buffers,
# in a fresh process and after a 137 GB allocation was returned to the driver
set -u
python3 - <<'P'
import time, torch, sys, os
sys.path.insert(0, os.getcwd())
import _moa_import
moa = _moa_import.load()
ops = moa.ops
dev = torch.device("cuda", 0)
def sync():
    torch.cuda.synchronize(); return time.perf_counter()
shapes = [(4096, 4096), (1024, 4096), (1024, 4096), (4096, 4096), (14336, 4096), (14336, 4096), (4096, 14336)] * 32
ws = [(torch.randn(s, device=dev) * 0.02).to(torch.bfloat16) for s in shapes]
for label in ("fresh process", "second time", "after 137 GB alloc + empty_cache"):
    if label.startswith("after"):
        big = torch.empty(int(137e9) // 2, dtype=torch.bfloat16, device=dev); big.fill_(1); torch.cuda.synchronize(); del big
        torch.cuda.empty_cache()
    t0 = sync()
    sc = [ops.awq_weight_scale(w, 128) for w in ws]
    t1 = sync()
    grams = [torch.zeros(w.shape[1], w.shape[1], dtype=torch.float32, device=dev) for w in ws]
    t2 = sync()
    print(f"{label}: weight scales {t1 - t0:.3f} s, Gram buffers ({sum(g.numel() for g in grams) * 4 / 1e9:.1f} GB) {t2 - t1:.3f} s", flush=True)
    del sc, grams
    if label == "second time":
        torch.cuda.empty_cache()
P
