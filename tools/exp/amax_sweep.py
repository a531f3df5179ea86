"""Running abs-max of one activation (moq_amax with accumulate, the MaxCalibrator collect of the calibration loop) at the
sizes the flow presents; run under `rocprofv3 --kernel-trace --stats` for kernel-only durations (event timing of a
sub-20-us kernel measures the Python call instead).  argv: rows (of 8192 bf16 columns)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load()
ops = moa.ops
dev = "cuda:0"
rows = int(sys.argv[1])
g = torch.Generator(device=dev).manual_seed(0)
xs = [(torch.randn(rows, 8192, generator=g, device=dev) * (1.0 + 0.01 * i)).to(torch.bfloat16) for i in range(4)]
buf = torch.zeros(1, dtype=torch.float32, device=dev)
for i in range(200):  # a calibration loop: the running maximum is broken by a few early batches only
    ops.reduce_amax(xs[i % 4], out=buf, accumulate=True)
torch.cuda.synchronize()
ref = max(float(x.float().abs().max()) for x in xs)
assert float(buf) == ref, (float(buf), ref)
print("rows", rows, "MB", rows * 8192 * 2 / 1e6, "amax", float(buf))
