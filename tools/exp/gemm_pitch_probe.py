"""Does the operand row pitch matter?  Same error GEMM at Cin = 4096 (8 KiB rows: every row of a tile starts at the same
offset modulo 8 KiB) against Cin = 4160 / 4224 (rows shifted by 128 / 256 B each) -- TFLOP/s per shape, single launch and
batched.  Usage (GPU box): python tools/exp/gemm_pitch_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load(); ops = moa.ops
DEV = "cuda:0"
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print("| T x Cout x Cin | single ms | TFLOP/s | batched(11) ms/cand | TFLOP/s | F.linear TFLOP/s |\n|---|---|---|---|---|---|")
for t, n, k in ((4096, 4096, 4096), (4096, 4096, 4160), (4096, 4096, 4224), (4096, 4096, 4352), (4096, 14336, 4096), (4096, 14336, 4160),
                (4096, 4096, 14336), (4096, 4096, 14400), (4096, 4096, 16384), (4096, 4096, 16448)):
    x = torch.randn(t, k, device=DEV).to(torch.bfloat16); w = (torch.randn(n, k, device=DEV) * 0.02).to(torch.bfloat16)
    ref = torch.nn.functional.linear(x, w)
    acc = torch.zeros(1, dtype=torch.float32, device=DEV)
    ms = timed(lambda: ops.awq_err_gemm(x, w, ref, None, acc))
    xs = x.unsqueeze(0).expand(11, -1, -1).contiguous(); ws = w.unsqueeze(0).expand(11, -1, -1).contiguous()
    accs = torch.zeros(11, dtype=torch.float32, device=DEV)
    msb = timed(lambda: ops.awq_err_gemm_multi(xs, ws, ref, None, accs), reps=3) / 11
    msl = timed(lambda: torch.nn.functional.linear(x, w))
    fl = 2.0 * t * n * k
    print(f"| {t} x {n} x {k} | {ms:.3f} | {fl / ms / 1e9:.0f} | {msb:.3f} | {fl / msb / 1e9:.0f} | {fl / msl / 1e9:.0f} |")
    del x, w, ref, xs, ws
