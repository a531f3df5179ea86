"""s_memtime statistics of the error GEMM's K loop (experiment library, GEO 10, MOQ_TUNE_GEMM_STAT): ticks per K-tile, ticks
parked at the tile boundary, lead between a tile's first DMA piece and the wait that needs it, ticks of the whole K loop --
per workgroup, summarised over the grid; and the kernel's wall time for the tick rate."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load()
from model_optimizer_amd import _lib
lib = _lib.lib()
dev = "cuda:0"
P = lambda t: ctypes.c_void_p(t.data_ptr())
for (t, n, k) in [(4096, 14336, 4096), (4096, 4096, 14336), (8192, 8192, 8192)]:
    x = torch.randn(t, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16)
    ref = torch.nn.functional.linear(x, w)
    nws = int(lib.moq_awq_err_gemm_workspace(t, n))
    ws = torch.zeros(nws, dtype=torch.float32, device=dev)
    acc = torch.zeros(1, dtype=torch.float32, device=dev)
    def run():
        rc = lib.moq_awq_err_gemm(P(x), P(w), P(ref), None, t, n, k, _lib.BF16, P(ws), P(acc), None)
        assert rc == 0, rc
    os.environ["MOQ_TUNE_GEMM_STAT"] = "0"
    for _ in range(3): run()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); [run() for _ in range(5)]; b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 5
    ntile = ((t + 255) // 256) * ((n + 255) // 256)
    out = {}
    for mode, name in ((1, "ticks per K-tile"), (2, "parked at the boundary per K-tile"), (3, "lead of the DMA per K-tile"), (4, "K loop, whole"), (5, "K loop, whole, in 100 MHz ticks")):
        os.environ["MOQ_TUNE_GEMM_STAT"] = str(mode)
        ws.zero_(); run(); torch.cuda.synchronize()
        v = ws[:ntile].double()
        out[name] = (v.mean().item(), v.min().item(), v.max().item())
    os.environ["MOQ_TUNE_GEMM_STAT"] = "0"
    rounds = ntile / 256
    print(f"{t} x {n} x {k}: kernel {ms * 1e3:.1f} us, {ntile} tiles = {rounds:.2f} rounds of 256 CUs, {k // 64} K-tiles per tile")
    for name, (m, lo, hi) in out.items():
        print(f"   {name}: mean {m:.0f} (min {lo:.0f}, max {hi:.0f}) ticks")
    full, real = out["K loop, whole"][0], out["K loop, whole, in 100 MHz ticks"][0]
    print(f"   -> K loop {real / 100:.1f} us per tile; s_memtime runs at {full / real * 100:.0f} MHz while the kernel runs; "
          f"64 MFMAs of 32 cycles per SIMD and K-tile = {2048 / out['ticks per K-tile'][0]:.2f} of the K-tile's ticks")
