"""One shape of the error GEMM, a few launches -- the target of `rocprofv3 --pmc ...` experiments."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load()
ops = moa.ops
t, n, k = 4096, 14336, 4096
x = torch.randn(t, k, device="cuda").to(torch.bfloat16)
w = (torch.randn(n, k, device="cuda") * 0.02).to(torch.bfloat16)
ref = torch.nn.functional.linear(x, w)
acc = torch.zeros(1, device="cuda")
for _ in range(6):
    ops.awq_err_gemm(x, w, ref, None, acc)
torch.cuda.synchronize()
