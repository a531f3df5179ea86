"""The REFERENCE's own eager CPU path for the bench workloads, timed in the BUILD container (the GPU box has no
/root/reference): SURVEY.md 8(d) "CPU baseline" -- max_calibrate(TensorQuantizer) + one QDQ per weight, bf16 input,
torch.set_num_threads(all cores), median of 3.  Writes profiles/r02_ref_cpu_baseline.json; bench.py quotes it beside
its own timed C port (`cpu_baseline.reference_eager`).

Usage (build container only):  python tools/ref_cpu_baseline.py
"""

import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ref_shim  # noqa: E402

ref_shim.install()

import modelopt.torch.quantization as mtq  # noqa: E402,F401
from modelopt.torch.quantization.config import QuantizerAttributeConfig  # noqa: E402
from modelopt.torch.quantization.model_calib import max_calibrate  # noqa: E402
from modelopt.torch.quantization.nn import TensorQuantizer  # noqa: E402


def main():
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    w = (torch.randn(4096, 4096, generator=torch.Generator().manual_seed(1234)) * 0.02).to(torch.bfloat16)
    n_bytes = w.numel() * 2
    cfgs = {"fp8": QuantizerAttributeConfig(num_bits=(4, 3), axis=None),
            "int8": QuantizerAttributeConfig(num_bits=8, axis=None),
            "int4g128": QuantizerAttributeConfig(num_bits=4, block_sizes={-1: 128, "type": "static"})}
    out = {"host": "build container", "cores": cores, "torch": torch.__version__, "sample": "4096x4096 bf16 weight, "
           "max_calibrate(TensorQuantizer) + one QDQ forward, median of 3 after one warm-up", "workloads": {}}
    for name, cfg in cfgs.items():
        times = []
        for rep in range(4):
            q = TensorQuantizer(cfg)
            t0 = time.perf_counter()
            with torch.no_grad():
                max_calibrate(q, lambda qq: qq(w), distributed_sync=False)
                q(w)
            times.append(time.perf_counter() - t0)
        med = statistics.median(times[1:])
        out["workloads"][name] = {"seconds": round(med, 4), "GBs": round(n_bytes / med / 1e9, 4)}
        print(name, out["workloads"][name], flush=True)
    with open(os.path.join(ROOT, "profiles", "r02_ref_cpu_baseline.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
