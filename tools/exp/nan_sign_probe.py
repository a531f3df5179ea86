"""fp8_pack of a +NaN next to a NEGATIVE finite neighbour in a bf16 packet gave 0xFF where the oracle has 0x7F"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _moa_import
moa = _moa_import.load()
ops = moa.ops
nan = float("nan")
for dt in (torch.bfloat16, torch.float16, torch.float32):
    for name, vals in (("neg, +NaN", [-1.5, nan] * 4), ("+NaN, neg", [nan, -1.5] * 4), ("pos, +NaN", [1.5, nan] * 4),
                       ("neg big, +NaN", [-1e6, nan] * 4), ("-inf, +NaN", [float("-inf"), nan] * 4)):
        x = torch.tensor(vals, dtype=torch.float32).to(dt)
        for s in (0.37, 1.0):
            q = ops.fp8_quantize(x.cuda(), torch.tensor([s], dtype=dt).cuda()).view(torch.uint8).cpu().tolist()
            print(dt, name, "scale", s, "| in", [hex(v & 0xFFFF) for v in (x.view(torch.int16) if dt != torch.float32 else (x.view(torch.int32) >> 16)).tolist()[:2]],
                  "| fp8_pack", [hex(v) for v in q[:4]])
