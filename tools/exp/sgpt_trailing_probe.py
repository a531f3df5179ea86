"""Single trailing updates of a 4096 x 14336 working matrix at several block positions: ms, TFLOP/s, GB/s of the w read-modify-write."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load(); ops = moa.ops
DEV = "cuda:0"
rows, ld = 4096, 14336
w = torch.randn(rows, ld, device=DEV) * 0.02
delta = torch.randn(rows, 128, device=DEV) * 0.01
hinv = torch.randn(ld, ld, device=DEV) * 0.05
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
print("| i1 | cols updated | ms | TFLOP/s | w RMW GB/s | library matmul+sub ms |\n|---|---|---|---|---|---|")
for i1 in (0, 4096, 8192, 12288, 14080):
    ncols = ld - i1 - 128
    ms = timed(lambda: ops.sgpt_trailing_update(w, i1, delta, hinv))
    def lib():
        w[:, i1 + 128:] -= delta.matmul(hinv[i1:i1 + 128, i1 + 128:])
    msl = timed(lib)
    print(f"| {i1} | {ncols} | {ms:.3f} | {2.0 * rows * 128 * ncols / ms / 1e9:.0f} | {rows * ncols * 8 / ms / 1e6:.0f} | {msl:.3f} |")
