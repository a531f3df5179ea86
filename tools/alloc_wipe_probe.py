"""Does an allocation that follows a large free wait for the driver's background wipe of the freed VRAM?  (round 5: one cold
lease in three paid 3.6 s in the first batch of the AWQ cache pass, right after bench.py had handed back 137 GB.)
Prints the time to get and touch 48 GiB (a) on a fresh process, (b) right after freeing 137 GB, (c) the same after a pause."""
import time

import torch

dev = torch.device("cuda:0")


def grab(n_gib, label):
    torch.cuda.synchronize()
    t = time.perf_counter()
    x = torch.empty(n_gib << 30, dtype=torch.uint8, device=dev)
    t1 = time.perf_counter() - t
    x.zero_()
    torch.cuda.synchronize()
    t2 = time.perf_counter() - t
    del x
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    print(f"{label}: malloc {t1 * 1e3:.1f} ms, malloc + touch {t2 * 1e3:.1f} ms, + free {1e3 * (time.perf_counter() - t):.1f} ms", flush=True)


torch.zeros(1, device=dev)
grab(48, "(a) fresh process, 48 GiB")
grab(48, "(a') again")
for pause in (0.0, 0.0, 2.0):
    big = torch.empty(137 << 30, dtype=torch.uint8, device=dev)
    big.zero_()
    torch.cuda.synchronize()
    t = time.perf_counter()
    del big
    torch.cuda.empty_cache()
    torch.cuda.synchronize()
    print(f"    freed 137 GiB in {1e3 * (time.perf_counter() - t):.1f} ms; pause {pause} s", flush=True)
    time.sleep(pause)
    grab(48, f"(b) 48 GiB after freeing 137 GiB, pause {pause} s")
    grab(48, "(b') again")
