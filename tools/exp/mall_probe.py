"""Does the Infinity Cache (256 MiB) keep a bucket of weights between the amax pass and the QDQ pass?

Per-tensor FP8 calibrate + QDQ reads every weight twice (amax, then QDQ: 6 B/elem algorithmic).  If the two passes
run bucket by bucket (bucket <= the cache) the second read may be served on-die.  Prints ms per whole-model step for
bucket sizes, out of place and in place, launched from Python and replayed from a HIP graph.
"""
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
moa = importlib.import_module("model-optimizer_amd")
mt = moa.multi_tensor

H, I, KV, L = 4096, 14336, 1024, 32
SHAPES = [(H, H), (KV, H), (KV, H), (H, H), (I, H), (I, H), (H, I)]


def timed(fn, iters=5):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def buckets(ws, outs, limit_bytes):
    tabs, cur_w, cur_o, cur_b = [], [], [], 0
    for w, o in zip(ws, outs):
        b = w.numel() * w.element_size()
        if cur_w and cur_b + b > limit_bytes:
            tabs.append(mt.SegmentTable(cur_w, cur_o))
            cur_w, cur_o, cur_b = [], [], 0
        cur_w.append(w); cur_o.append(o); cur_b += b
    if cur_w:
        tabs.append(mt.SegmentTable(cur_w, cur_o))
    return tabs


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    ws = [(torch.randn(s, device=dev) * 0.02).to(torch.bfloat16) for _ in range(L) for s in SHAPES]
    outs = [torch.empty_like(w) for w in ws]
    total = sum(w.numel() * 2 for w in ws)
    print(f"{len(ws)} tensors, {total / 1e9:.2f} GB")
    for inplace in (False, True):
        o = ws if inplace else outs
        whole = mt.SegmentTable(ws, o)
        def two_pass():
            whole.calibrate_amax(); whole.fake_quant_e4m3()
        ms = timed(two_pass)
        print(f"inplace={inplace} whole-model two-pass: {ms:.3f} ms  ({total / ms / 1e6:.0f} GB/s weights)")
        for mb in (24, 48, 96, 128, 160, 240, 480):
            tabs = buckets(ws, o, mb * 1e6)
            def step():
                for t in tabs:
                    t.calibrate_amax(); t.fake_quant_e4m3()
            ms_py = timed(step)
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                step()
                torch.cuda.synchronize()
                with torch.cuda.graph(g, stream=s):
                    step()
            ms_g = timed(g.replay)
            print(f"inplace={inplace} bucket<={mb:4d} MB ({len(tabs):3d} buckets): python {ms_py:.3f} ms, graph {ms_g:.3f} ms"
                  f"  ({total / ms_g / 1e6:.0f} GB/s weights)")
            # the bucketed result must equal the whole-model result
        a = whole.calibrate_amax().clone()
        b = torch.cat([t.calibrate_amax() for t in buckets(ws, o, 96e6)])
        assert torch.equal(a, b)


if __name__ == "__main__":
    main()
