"""The library GEMM (F.linear -> hipBLASLt) on the error GEMM's shape, a few launches -- target of the same PMC pass as
tools/exp/gemm_pmc.py, for comparison (L2 hit rate, matrix-pipe busy, LDS activity of the MT256x256x64 kernel)."""
import sys, torch
t, n, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (4096, 14336, 4096)
x = torch.randn(t, k, device="cuda").to(torch.bfloat16)
w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
for _ in range(8):
    y = torch.nn.functional.linear(x, w)
torch.cuda.synchronize()
