"""How many host threads does the CPU oracle actually scale to on this box?  (bench.py's cpu_baseline leg)"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from oracle import oracle  # noqa: E402

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    if os.path.exists(p):
        print(p, open(p).read().strip())
gomp = ctypes.CDLL("libgomp.so.1")
for shape in ((4096, 4096), (14336, 4096)):
    w = (torch.randn(*shape, generator=torch.Generator().manual_seed(1)) * 0.02).to(torch.bfloat16)
    for th in (1, 4, 8, 16, 32, 64, 128, 256):
        if th > (os.cpu_count() or 1):
            break
        gomp.omp_set_num_threads(th)
        oracle.fake_quant_e4m3(w, oracle.reduce_amax(w).reshape(1))
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 1.5:
            a = oracle.reduce_amax(w)
            oracle.fake_quant_e4m3(w, a.reshape(1))
            reps += 1
        dt = time.perf_counter() - t0
        print(f"{shape} threads={th:3d}: {reps * w.numel() * 2 / dt / 1e9:7.3f} GB/s of weights")
