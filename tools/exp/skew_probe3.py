"""Which cheap probe predicts how fast a whole-model out-of-place QDQ runs into a given output pool?  Four pools; for each: the
QDQ over everything, the QDQ over the first ~1 GB of tensors only, and a torch copy of the same first tensors."""
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
print(f"in place: {timed(lambda: t0.fake_quant_e4m3()):.3f} ms")
pools = [torch.empty(n_tot, dtype=torch.bfloat16, device=dev) for _ in range(4)]
print("| pool | whole-model QDQ ms | QDQ of the first 14 tensors (0.87 GB) ms | QDQ of the LAST 14 tensors ms | torch copy of the first 14 ms |")
print("|---|---|---|---|---|")
for pi, pool in enumerate(pools):
    outs, off = [], 0
    for w in ws:
        outs.append(pool[off:off + w.numel()].view(w.shape)); off += w.numel()
    t = SegmentTable(ws, outputs=outs); t.calibrate_amax()
    full = timed(lambda: t.fake_quant_e4m3())
    th = SegmentTable(ws[:14], outputs=outs[:14]); th.calibrate_amax()
    head = timed(lambda: th.fake_quant_e4m3(), reps=10)
    tt = SegmentTable(ws[-14:], outputs=outs[-14:]); tt.calibrate_amax()
    tail = timed(lambda: tt.fake_quant_e4m3(), reps=10)
    def cp():
        for w, o in zip(ws[:14], outs[:14]): o.copy_(w)
    c = timed(cp, reps=10)
    print(f"| {pi} (0x{pool.data_ptr():x}) | {full:.3f} | {head:.4f} | {tail:.4f} | {c:.4f} |")
    del t, th, tt, outs
