"""Is the error GEMM power-bound?  Same kernel, same shape, operands of decreasing bit activity: N(0,1) x N(0, 0.02^2) bf16 (the
bench data), small integers, constants, zeros.  TFLOP/s, and the shader clock while the kernel runs (s_memtime against the
100 MHz s_memrealtime, experiment library), for ours and for the library GEMM (F.linear; clock not available there)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load()
from model_optimizer_amd import _lib
lib = _lib.lib()
dev = "cuda:0"
P = lambda t: ctypes.c_void_p(t.data_ptr())
t, n, k = 4096, 14336, 4096
def timed(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); [fn() for _ in range(reps)]; b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
g = torch.Generator(device=dev).manual_seed(0)
cases = {
    "N(0,1) x N(0,0.02^2)": (torch.randn(t, k, device=dev, generator=g), torch.randn(n, k, device=dev, generator=g) * 0.02),
    "integers -3..3": (torch.randint(-3, 4, (t, k), device=dev, generator=g).float(), torch.randint(-3, 4, (n, k), device=dev, generator=g).float()),
    "all ones": (torch.ones(t, k, device=dev), torch.ones(n, k, device=dev)),
    "zeros": (torch.zeros(t, k, device=dev), torch.zeros(n, k, device=dev)),
}
print("| operands | err-GEMM ms | TFLOP/s | shader clock in the K loop | MFMA pipe busy in the K loop | F.linear ms | TFLOP/s |")
print("|---|---|---|---|---|---|---|")
for name, (xf, wf) in cases.items():
    x, w = xf.to(torch.bfloat16), wf.to(torch.bfloat16)
    ref = torch.nn.functional.linear(x, w)
    ws = torch.zeros(int(lib.moq_awq_err_gemm_workspace(t, n)), dtype=torch.float32, device=dev)
    acc = torch.zeros(1, dtype=torch.float32, device=dev)
    def run():
        assert lib.moq_awq_err_gemm(P(x), P(w), P(ref), None, t, n, k, _lib.BF16, P(ws), P(acc), None) == 0
    os.environ["MOQ_TUNE_GEMM_STAT"] = "0"
    ms = timed(run)
    st = {}
    for mode in (1, 4, 5):
        os.environ["MOQ_TUNE_GEMM_STAT"] = str(mode)
        for _ in range(3): run()
        ws.zero_(); run(); torch.cuda.synchronize()
        st[mode] = ws[:896].double().mean().item()
    os.environ["MOQ_TUNE_GEMM_STAT"] = "0"
    ml = timed(lambda: torch.nn.functional.linear(x, w))
    fl = 2.0 * t * n * k
    print(f"| {name} | {ms:.3f} | {fl / ms / 1e9:.0f} | {st[4] / st[5] * 100:.0f} MHz | {2048 / st[1]:.2f} | {ml:.3f} | {fl / ml / 1e9:.0f} |")
