import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load(); ops = moa.ops
from oracle import oracle
for dtype in (torch.bfloat16, torch.float16):
    torch.manual_seed(7)
    t, n, k = 192, 320, 4096
    x = torch.randn(t, k).to(dtype); w = (torch.randn(n, k) * 0.02).to(dtype)
    _, want = oracle.awq_err_gemm(x, w, None, None, return_out=True)
    got = ops.gemm_nt(x.cuda(), w.cuda()).cpu()
    lib = torch.nn.functional.linear(x.cuda(), w.cuda()).cpu()
    exact = (x.double() @ w.double().T)
    for name, g in (("ours", got), ("torch", lib)):
        same = (g.view(torch.int16) == want.view(torch.int16)).float().mean().item()
        d = (g.double() - want.double()).abs()
        sp = torch.finfo(dtype).eps * 2.0 ** torch.floor(torch.log2(want.double().abs().clamp_min(1e-30)))
        u = d / sp
        i = u.argmax()
        print(dtype, name, "identical", same, "max ulps", u.max().item(), "n>1ulp", int((u > 1.001).sum()),
              "worst: got", g.reshape(-1)[i].item(), "want", want.reshape(-1)[i].item(), "exact", exact.reshape(-1)[i].item())
        print("   max |fp32-level err| vs exact before rounding n/a; max abs diff", d.max().item())
