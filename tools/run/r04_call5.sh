#!/bin/bash
# round 4, call 5: the one-chunk-per-workgroup kernels (cheap prologue) vs the strided order, same sets; bitwise check first
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/r04g; mkdir -p $O
export MOQ_LIB_PATH=$ROOT/model-optimizer_amd/csrc/libmoquant_exp.so
python3 - <<'P' 2>&1 | tee $O/check.log
import os, sys, torch
sys.path.insert(0, os.getcwd())
import _moa_import
moa = _moa_import.load()
from model_optimizer_amd.multi_tensor import SegmentTable
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(3)
shapes = [(257, 384), (1024, 4096), (3, 5), (8192,), (640, 128), (1, 8), (4096, 1024), (77, 128)] * 40  # 320 segments: two lookup rounds
for dt in (torch.bfloat16, torch.float16, torch.float32):
    ws = [(torch.randn(s, generator=g, device=dev) * 0.02).to(dt) for s in shapes]
    ws[5] = ws[5].reshape(-1)[:8]
    outs = {}
    for mode in ("0", "1"):
        os.environ["MOQ_TUNE_MAP_ONE"] = mode
        t = SegmentTable(ws); t.calibrate_amax()
        a = [o.clone() for o in t.fake_quant_e4m3()]
        b = [o.clone() for o in t.fake_quant_int(8, False, True)]
        gs = [w for w in ws if w.numel() % 128 == 0]
        tg = SegmentTable(gs, group_size=128)
        c = [o.clone() for o in tg.amax_qdq_int_group(4, False, False)]
        outs[mode] = (a, b, c, tg.amax_flat.clone())
    ok = all(torch.equal(x.view(torch.uint8), y.view(torch.uint8)) for k in range(3) for x, y in zip(outs["0"][k], outs["1"][k]))
    ok = ok and torch.equal(outs["0"][3], outs["1"][3])
    print(dt, "one-chunk kernels bitwise equal to the strided kernels:", ok)
os.environ.pop("MOQ_TUNE_MAP_ONE")
P
python3 tools/pool_placement.py --sweep --quick --sets 4 --out $O/sweep.json > $O/sweep.log 2> $O/sweep.err
echo "sweep rc=$?"; cat $O/sweep.log; tail -3 $O/sweep.err
