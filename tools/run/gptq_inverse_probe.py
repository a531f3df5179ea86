"""(gpurun probe, round 4) where the 15 s of the Llama-3-8B GPTQ update go: the damped inverse factor (cholesky,
cholesky_inverse, cholesky(upper)) under torch's default linalg backend and under hipSOLVER, the block sweep + trailing
updates, and the reported Hessian-weighted error, at the two Hessian sizes of the model."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import  # noqa: E402

moa = _moa_import.load()
gptq = moa.gptq


def clock(fn, n=3):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def main():
    dev = "cuda"
    torch.manual_seed(0)
    for cols, rows in ((4096, 4096), (4096, 14336), (14336, 4096)):
        x = torch.randn(8192, cols, device=dev) * (torch.rand(cols, device=dev) + 0.5)
        h = (x.t() @ x) * (2.0 / x.shape[0])
        w = torch.randn(rows, cols, device=dev) * 0.02
        out = {}
        for lib in ("default", "cusolver", "magma"):
            try:
                torch.backends.cuda.preferred_linalg_library(lib)
                out[lib] = round(clock(lambda: gptq.compute_hessian_inverse(h, w, 0.01), 2), 4)
            except Exception as e:  # noqa: BLE001
                out[lib] = repr(e)[:80]
        torch.backends.cuda.preferred_linalg_library("default")
        hinv = gptq.compute_hessian_inverse(h, w, 0.01)
        model = torch.nn.Sequential(torch.nn.Linear(cols, rows, bias=False)).to(dev)
        model[0].weight.data.copy_(w)
        moa.quantize(model, moa.model_quant.INT4_BLOCKWISE_WEIGHT_ONLY_CFG, None)
        q = model[0].weight_quantizer
        ww = w.clone()
        out["sweep"] = round(clock(lambda: gptq.gptq_blockwise_update(ww.copy_(w), hinv, 128, q), 2), 4)
        out["mse"] = round(clock(lambda: gptq.relative_mse(ww, w, h), 2), 4)
        out["dead_cols"] = round(clock(lambda: bool(gptq.dead_columns(w).any()), 3), 5)
        print(f"Cin={cols} Cout={rows}", out, flush=True)


if __name__ == "__main__":
    main()
