"""Gram accumulation (error GEMM MODE 2: G = G * decay + scale * X^T X, upper tiles) at the AWQ shapes: one launch over
T staged tokens.  A/B: MOQ_TUNE_GEMM_GROUP (workgroup -> tile order)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load()
ops = moa.ops
dev = "cuda:0"
for cin, tokens in ((4096, 16384), (14336, 16384), (8192, 16384)):
    x = torch.randn(tokens, cin, device=dev).to(torch.bfloat16)
    g = torch.zeros(cin, cin, dtype=torch.float32, device=dev)
    fn = lambda: ops.hessian_accum(g, x, 1.0, 1.0 / tokens, upper_only=True)
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    tiles = (cin // 256) * (cin // 256 + 1) // 2
    flop = 2.0 * tiles * 256 * 256 * tokens
    print(f"Cin {cin:6d} T {tokens}: {ms:7.3f} ms  {flop / ms / 1e9:7.0f} TFLOP/s on the {tiles} upper tiles (incl. the transpose of x)")
