"""AWQ-clip block-search kernel (moq_awq_clip_loss) on Llama shapes: ms per call, weight GB/s, MFMA TFLOP/s.
One call = one linear x one calibration batch (4096 tokens sub-sampled to 64), all 11 clip ratios.
Usage (GPU box): python tools/clip_bench.py [> profiles/rNN_clip_table.md]"""

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _moa_import  # noqa: E402

moa = _moa_import.load()
ops = moa.ops
DEV = "cuda:0"


def main():
    torch.manual_seed(1234)
    shrinks = torch.tensor([round(float(k), 2) for k in torch.arange(0.5, 1.0, 0.05)] + [1.0], device=DEV)
    print("| linear (Cout x Cin), bf16, g=128, 64 of 4096 tokens, 11 clip ratios | ms | weight GB/s | block-dot TFLOP/s (12 x 2*64*Cout*Cin) |")
    print("|---|---|---|---|")
    for name, co, ci in [("8b q/o", 4096, 4096), ("8b k/v", 1024, 4096), ("8b gate/up", 14336, 4096),
                         ("8b down", 4096, 14336), ("70b gate/up", 28672, 8192), ("70b down", 8192, 28672)]:
        w = (torch.randn(co, ci, device=DEV) * 0.02).to(torch.bfloat16)
        x = torch.randn(4096, ci, device=DEV).to(torch.bfloat16)
        amax = ops.reduce_amax(w.view(-1, 128), axis=(1,)).float().reshape(-1)
        loss = torch.zeros(11, ci // 128, co, device=DEV)
        fn = lambda: ops.awq_clip_loss(x, w, amax, shrinks, 128, 4, loss, token_step=64)  # noqa: E731
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(f"| {name} {co}x{ci} | {ms:.3f} | {2 * co * ci / ms / 1e6:.0f} | {12 * 2 * 64 * co * ci / ms / 1e9:.0f} |")


if __name__ == "__main__":
    main()
