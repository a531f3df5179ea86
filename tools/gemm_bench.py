"""MFMA error-GEMM throughput on one MI355X at the Llama-3-8B / 70B linear shapes of one calibration batch
(8 x 512 = 4096 tokens).  2*T*Cout*Cin flop / HIP-event time, against the 2.5 PFLOP/s dense bf16 peak.
A/B knob: MOQ_TUNE_GEMM_GEO=4|10 (read once per process).  Also times torch's library GEMM (F.linear) + the
unfused loss ops the reference would run, for scale.
Usage (GPU box): python tools/gemm_bench.py [> profiles/rNN_gemm_table.md]"""

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _moa_import  # noqa: E402

if "--lib" in sys.argv:  # the experiment library instead of the release one: _lib reads MOQ_LIB_PATH when it is first imported
    os.environ["MOQ_LIB_PATH"] = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
moa = _moa_import.load()
ops = moa.ops
DEV = "cuda:0"
PEAK = 2500.0


def timed(fn, reps=10):
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
    shapes = [("8b q/o", 4096, 4096, 4096), ("8b k/v", 4096, 1024, 4096), ("8b gate/up", 4096, 14336, 4096),
              ("8b down", 4096, 4096, 14336), ("70b gate/up", 4096, 28672, 8192), ("70b down", 4096, 8192, 28672),
              ("square 8192", 8192, 8192, 8192)]
    print(f"variant: MOQ_TUNE_GEMM_GEO={os.environ.get('MOQ_TUNE_GEMM_GEO', '10 (default)')}\n")
    print("| shape (T x Cout x Cin) | fused err-GEMM ms | TFLOP/s | frac of 2.5 PF | batched (11 cand.) ms/cand | TFLOP/s | frac | F.linear ms | F.linear TFLOP/s | F.linear + unfused loss ms |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    for name, t, n, k in shapes:
        x = torch.randn(t, k, device=DEV).to(torch.bfloat16)
        w = (torch.randn(n, k, device=DEV) * 0.02).to(torch.bfloat16)
        ref = torch.nn.functional.linear(x, w)
        acc = torch.zeros(1, dtype=torch.float32, device=DEV)
        ms = timed(lambda: ops.awq_err_gemm(x, w, ref, None, acc))
        ncand = 11 if t * k * 11 * 2 + n * k * 11 * 2 < 60e9 else 4
        xs = x.unsqueeze(0).repeat(ncand, 1, 1)
        wh = w.unsqueeze(0).repeat(ncand, 1, 1)
        accm = torch.zeros(ncand, dtype=torch.float32, device=DEV)
        ms_multi = timed(lambda: ops.awq_err_gemm_multi(xs, wh, ref, None, accm), reps=3) / ncand
        del xs, wh
        ms_lin = timed(lambda: torch.nn.functional.linear(x, w))

        def unfused():
            out = torch.nn.functional.linear(x, w)
            return (out - ref).float().pow(2).mean()

        ms_unf = timed(unfused)
        fl = 2.0 * t * n * k
        print(f"| {name} {t}x{n}x{k} | {ms:.3f} | {fl / ms / 1e9:.0f} | {fl / ms / 1e9 / PEAK:.3f} | {ms_multi:.3f} | "
              f"{fl / ms_multi / 1e9:.0f} | {fl / ms_multi / 1e9 / PEAK:.3f} | {ms_lin:.3f} | "
              f"{fl / ms_lin / 1e9:.0f} | {ms_unf:.3f} |")
        del x, w, ref


if __name__ == "__main__":
    main()
