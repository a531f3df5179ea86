"""In place: does it matter whether the 224 weights are 224 separate allocations or views of one arena?  Per-tensor abs-max and
in-place FP8 QDQ over the whole model, for the separately allocated set and for three arenas."""
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
print("| weights | abs-max ms | in-place FP8 QDQ ms | TB/s |\n|---|---|---|---|")
def row(name, t):
    am = timed(lambda: t.calibrate_amax()); q = timed(lambda: t.fake_quant_e4m3())
    print(f"| {name} | {am:.3f} | {q:.3f} | {n_tot * 4 / q / 1e9:.2f} |")
row("224 separate allocations", t0)
for k in range(3):
    arena = torch.empty(n_tot, dtype=torch.bfloat16, device=dev)
    vs, off = [], 0
    for w in ws:
        v = arena[off:off + w.numel()].view(w.shape); v.copy_(w); vs.append(v); off += w.numel()
    t = SegmentTable(vs, outputs=vs)
    row(f"arena {k} (0x{arena.data_ptr():x})", t)
    globals()[f"keep{k}"] = arena
row("224 separate allocations (again)", t0)
