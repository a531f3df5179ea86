"""Separates the two candidates of skew_probe.py: (a) the byte distance between input and output, (b) WHICH allocation the outputs
live in.  Three output pools are allocated and kept; inside each, the same set of skews is timed (whole-model FP8 QDQ)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load()
from model_optimizer_amd.multi_tensor import SegmentTable
dev = "cuda:0"
shapes = ([(4096, 4096), (1024, 4096), (1024, 4096), (4096, 4096), (14336, 4096), (14336, 4096), (4096, 14336)]) * 32
g = torch.Generator(device=dev).manual_seed(0)
ws = [(torch.randn(s, generator=g, device=dev) * 0.02).to(torch.bfloat16) for s in shapes]
n_tot = sum(w.numel() for w in ws)

def timed(fn, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

t0 = SegmentTable(ws, outputs=ws); t0.calibrate_amax()
for _ in range(100): t0.fake_quant_e4m3()
torch.cuda.synchronize()
pools = [torch.empty(n_tot + len(ws) * (1 << 21), dtype=torch.bfloat16, device=dev) for _ in range(3)]
skews = (0, 16384, 32768, 65536, 131072, 1 << 20)
print("| pool | " + " | ".join(f"skew {s}" for s in skews) + " | read-only sweep of the pool | fill of the pool |")
print("|---|" + "---|" * (len(skews) + 2))
for pi, pool in enumerate(pools):
    row = []
    for skew in skews:
        outs, off, base = [], 0, pool.data_ptr()
        for w in ws:
            cur = base + off * 2
            adj = ((w.data_ptr() + skew) % (1 << 21) - cur) % (1 << 21)
            off += adj // 2
            outs.append(pool[off:off + w.numel()].view(w.shape))
            off += w.numel()
        t = SegmentTable(ws, outputs=outs); t.calibrate_amax()
        row.append(timed(lambda: t.fake_quant_e4m3()))
        del t, outs
    flat = pool[:n_tot]
    rd = timed(lambda: moa.ops.reduce_amax(flat))
    wr = timed(lambda: flat.zero_())
    print(f"| {pi} (0x{pool.data_ptr():x}) | " + " | ".join(f"{m:.3f}" for m in row) + f" | {n_tot * 2 / rd / 1e9:.2f} TB/s | {n_tot * 2 / wr / 1e9:.2f} TB/s |")
ms = timed(lambda: t0.fake_quant_e4m3())
wflat_rd = sum(timed(lambda w=w: moa.ops.reduce_amax(w), reps=2) for w in ws[:7])
print(f"| in place | {ms:.3f} |")
