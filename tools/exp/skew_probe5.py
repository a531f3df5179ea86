"""Does the traversal of the copy-shaped multi-tensor kernels decide which output pools are slow?  Experiment library
(MOQ_TUNE_CHUNKS_PER_WG is read on every call there): four pools x chunks-per-workgroup settings, whole-model FP8 QDQ.
grid = n_chunks / div clamped to [2048, 131072]; a workgroup walks chunks blockIdx, blockIdx + grid, ..."""
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
divs = (1, 2, 4, 8, 16, 32, 128, 416)
pools = [torch.empty(n_tot, dtype=torch.bfloat16, device=dev) for _ in range(4)]
print("| outputs | " + " | ".join(f"div {d}" for d in divs) + " |")
print("|---|" + "---|" * len(divs))
def row(name, t):
    r = []
    for d in divs:
        os.environ["MOQ_TUNE_CHUNKS_PER_WG"] = str(d)
        r.append(timed(lambda: t.fake_quant_e4m3()))
    print(f"| {name} | " + " | ".join(f"{m:.3f}" for m in r) + " |")
row("in place", t0)
for pi, pool in enumerate(pools):
    outs, off = [], 0
    for w in ws:
        outs.append(pool[off:off + w.numel()].view(w.shape)); off += w.numel()
    t = SegmentTable(ws, outputs=outs); t.calibrate_amax()
    row(f"pool {pi}", t)
    del t, outs
