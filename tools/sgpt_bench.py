"""SparseGPT kernels on Llama shapes: Hessian accumulation (transpose + MFMA contraction into the fp32 Hessian) next
to the reference's arithmetic (fp32 copy + fp32 library GEMM), and the create_sgpt_mask column sweep + trailing GEMM.
Usage (GPU box): python tools/sgpt_bench.py [> profiles/rNN_sgpt_table.md]"""

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _moa_import  # noqa: E402

moa = _moa_import.load()
ops, sparsity = moa.ops, moa.sparsity
DEV = "cuda:0"


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    torch.manual_seed(0)
    print("| Hessian update (upper tiles; one symmetrize pass at the end), 4096 tokens bf16 | MFMA path ms | TFLOP/s (2 T Cin^2) | fp32 library path ms | TFLOP/s |")
    print("|---|---|---|---|---|")
    for cin in (4096, 8192, 14336):
        x = torch.randn(4096, cin, device=DEV).to(torch.bfloat16)
        h = torch.zeros(cin, cin, device=DEV)
        ms = timed(lambda: ops.hessian_accum(h, x, 0.5, 0.001, upper_only=True))
        h2 = torch.zeros(cin, cin, device=DEV)

        def ref():
            xf = x.float()
            h2.mul_(0.5).addmm_(xf.t(), xf, alpha=0.001)

        ms2 = timed(ref)
        fl = 2 * 4096 * cin * cin
        print(f"| Cin={cin} | {ms:.3f} | {fl / ms / 1e9:.0f} | {ms2:.3f} | {fl / ms2 / 1e9:.0f} |")
        del h, h2, x
    print()
    print("| create_sgpt_mask (2:4, col block 128), bf16 weight | total ms | column sweeps ms | trailing updates (fp32 MFMA, fma chain) ms | TFLOP/s | same updates as library fp32 matmul ms |")
    print("|---|---|---|---|---|---|")
    for co, ci in ((4096, 4096), (14336, 4096), (4096, 14336)):
        w = (torch.randn(co, ci, device=DEV) * 0.02).to(torch.bfloat16)
        a = torch.randn(ci, 2 * ci, device=DEV)
        hinv = torch.linalg.cholesky(torch.linalg.inv(a @ a.t() / (2 * ci) + 0.1 * torch.eye(ci, device=DEV)), upper=True).contiguous()
        cfg = {"pattern": "2:4 sparsity", "col_block_size": 128}
        total = timed(lambda: sparsity.create_sgpt_mask(w, None, cfg, hessian_inv=hinv), reps=2)
        wf = w.float().contiguous()
        sweep = timed(lambda: [ops.sgpt_block_sweep(wf, i1, 128, hinv) for i1 in range(0, ci, 128)], reps=2)
        delta = torch.randn(co, 128, device=DEV)
        upd = timed(lambda: [ops.sgpt_trailing_update(wf, i1, delta, hinv) for i1 in range(0, ci - 128, 128)], reps=2)

        def lib_updates():
            for i1 in range(0, ci - 128, 128):
                wf[:, i1 + 128:] -= delta.matmul(hinv[i1:i1 + 128, i1 + 128:])

        lib = timed(lib_updates, reps=2)
        fl = sum(2.0 * co * 128 * (ci - i1 - 128) for i1 in range(0, ci - 128, 128))
        print(f"| {co}x{ci} | {total:.2f} | {sweep:.2f} | {upd:.2f} | {fl / upd / 1e9:.0f} | {lib:.2f} |")


if __name__ == "__main__":
    main()
