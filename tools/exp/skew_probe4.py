"""Where inside a slow output pool is the time lost?  Five pools: whole-model FP8 QDQ, then the same in seven slices of 32
tensors (2 GB each).  Also: outputs as separate torch.empty_like tensors (the current default), twice."""
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
print(f"in place: {timed(lambda: t0.fake_quant_e4m3()):.3f} ms; free memory {torch.cuda.mem_get_info()[0] / 1e9:.0f} GB")
keep = []
for rep in range(2):
    outs = [torch.empty_like(w) for w in ws]
    t = SegmentTable(ws, outputs=outs); t.calibrate_amax()
    print(f"separate torch.empty_like outputs #{rep}: {timed(lambda: t.fake_quant_e4m3()):.3f} ms")
    keep.append(outs); del t
pools = [torch.empty(n_tot, dtype=torch.bfloat16, device=dev) for _ in range(5)]
print("| pool | whole model ms | " + " | ".join(f"tensors {i}-{i + 31}" for i in range(0, 224, 32)) + " | sum of slices |")
print("|---|---|" + "---|" * 8)
for pi, pool in enumerate(pools):
    outs, off = [], 0
    for w in ws:
        outs.append(pool[off:off + w.numel()].view(w.shape)); off += w.numel()
    t = SegmentTable(ws, outputs=outs); t.calibrate_amax()
    full = timed(lambda: t.fake_quant_e4m3())
    parts = []
    for i in range(0, 224, 32):
        tp = SegmentTable(ws[i:i + 32], outputs=outs[i:i + 32]); tp.calibrate_amax()
        parts.append(timed(lambda: tp.fake_quant_e4m3()))
        del tp
    print(f"| {pi} (0x{pool.data_ptr():x}) | {full:.3f} | " + " | ".join(f"{p:.3f}" for p in parts) + f" | {sum(parts):.3f} |")
    del t, outs
