"""Three error-GEMM shapes, single launch, TFLOP/s -- for A/B runs of differently built libraries (MOQ_LIB_PATH)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load()
ops = moa.ops
out = []
for t, n, k in [(4096, 4096, 4096), (4096, 4096, 14336), (4096, 28672, 8192), (8192, 8192, 8192)]:
    x = torch.randn(t, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
    ref = torch.nn.functional.linear(x, w)
    acc = torch.zeros(1, device="cuda")
    for _ in range(3):
        ops.awq_err_gemm(x, w, ref, None, acc)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.awq_err_gemm(x, w, ref, None, acc)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    out.append(f"{t}x{n}x{k}: {2.0 * t * n * k / ms / 1e9:.0f}")
print(os.environ.get("MOQ_LIB_PATH", "default").split("/")[-1], " | ".join(out))
