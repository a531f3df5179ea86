"""Which hipBLASLt kernel serves F.linear on the AWQ shapes (its name encodes the tiling / staging recipe)."""
import torch
for t, n, k in [(4096, 14336, 4096), (4096, 4096, 4096), (4096, 4096, 14336), (8192, 8192, 8192)]:
    x = torch.randn(t, k, device="cuda").to(torch.bfloat16)
    w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
    for _ in range(3):
        torch.nn.functional.linear(x, w)
torch.cuda.synchronize()
