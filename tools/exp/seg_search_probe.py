"""Does the per-workgroup segment lookup (binary search over the chunk prefix sums) cost the whole-model launches anything?
The same 13.96 GB of bf16 in ONE allocation, described as 1 / 224 / 3584 / 28672 segments: abs-max sweep and FP8 QDQ in place.

Round 6, one MI355X: no.  FP8 QDQ 4.57 / 4.57 / 4.55 / 4.45 ms (6.11-6.13 TB/s), abs-max 2.03 / 2.02 / 1.98 ms for 224 /
3584 / 28672 segments (6.88-6.90 TB/s); ONE segment is the slow case of the abs-max (2.43 ms: its fold stage is one
workgroup's 852 k chunk maxima)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import

moa = _moa_import.load()
from model_optimizer_amd.multi_tensor import SegmentTable

dev = "cuda:0"
n = 6979321856
flat = (torch.randn(n // 64, device=dev) * 0.02).to(torch.bfloat16).repeat(64)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for rnd in range(2):
    for n_seg in (1, 224, 3584, 28672):
        per = n // n_seg // 8192 * 8192
        parts = [flat[i * per:(i + 1) * per] for i in range(n_seg)]
        tab = SegmentTable(parts, outputs=parts)
        tot = per * n_seg
        ta = timed(tab.calibrate_amax)
        tq = timed(tab.fake_quant_e4m3)
        print(f"segments {n_seg:6d}: abs-max {ta:.3f} ms = {tot * 2 / ta / 1e9:.3f} TB/s   fp8 qdq {tq:.3f} ms = {tot * 4 / tq / 1e9:.3f} TB/s", flush=True)
