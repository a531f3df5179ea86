"""Does the distance between the input and the output of an out-of-place QDQ matter?  The whole-model multi-tensor FP8 QDQ (the
bench's dominant kernel) over Llama-3-8B-sized weights, outputs carved from one pool with a chosen byte skew after each tensor
(torch's allocator hands out 2 MiB-aligned blocks: input and output of a tensor are a multiple of 2 MiB apart), and in place."""
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

# warm the clocks
t0 = SegmentTable(ws, outputs=None); t0.calibrate_amax()
for _ in range(100): t0.fake_quant_e4m3()
torch.cuda.synchronize()
print(f"| outputs | FP8 QDQ ms | TB/s (4 B / element) |\n|---|---|---|")
ms = timed(lambda: t0.fake_quant_e4m3())
print(f"| torch.empty_like per tensor (2 MiB-aligned blocks) | {ms:.3f} | {n_tot * 4 / ms / 1e9:.2f} |")
base_delta = [(o.data_ptr() - w.data_ptr()) % (1 << 21) for w, o in zip(ws, t0.outputs)][:4]
print("  (output - input) mod 2 MiB of the first tensors:", base_delta, file=sys.stderr)
del t0
for skew in (0, 16384, 24576, 32768, 49152, 65536, 131072, 1 << 20, (1 << 21) - 65536, (1 << 21) - 16384):
    pool = torch.empty(n_tot + len(ws) * ((skew + (1 << 21)) // 2 + 64), dtype=torch.bfloat16, device=dev)
    outs, off = [], 0
    base = pool.data_ptr()
    for w in ws:
        # place the output so that (out - in) mod 2 MiB == skew
        cur = base + off * 2
        want = (w.data_ptr() + skew) % (1 << 21)
        adj = (want - cur) % (1 << 21)
        off += adj // 2
        outs.append(pool[off:off + w.numel()].view(w.shape))
        off += w.numel()
    t = SegmentTable(ws, outputs=outs); t.calibrate_amax()
    ms = timed(lambda: t.fake_quant_e4m3())
    ms_mx = timed(lambda: t.mx_fused_amax_convert(32, "E2M1"))
    tg = SegmentTable(ws, outputs=outs, group_size=128)
    ms_g = timed(lambda: tg.amax_qdq_int_group(4, False, False))
    print(f"| pool, (out - in) mod 2 MiB = {skew} B | {ms:.3f} | {n_tot * 4 / ms / 1e9:.2f} | MXFP4 {ms_mx:.3f} ms {n_tot * 4 / ms_mx / 1e9:.2f} | INT4 g128 {ms_g:.3f} ms {n_tot * 4 / ms_g / 1e9:.2f} |")
    del t, tg, outs, pool
t = SegmentTable(ws, outputs=ws); t.calibrate_amax()
ms = timed(lambda: t.fake_quant_e4m3())
print(f"| in place | {ms:.3f} | {n_tot * 4 / ms / 1e9:.2f} |")
x = torch.empty(1 << 29, dtype=torch.bfloat16, device=dev); y = torch.empty_like(x)
ms = timed(lambda: y.copy_(x))
print(f"| (torch copy of 1 GiB, for the node class) | {ms:.3f} | {x.numel() * 4 / ms / 1e9:.2f} |")
