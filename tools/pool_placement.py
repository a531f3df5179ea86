"""Why does the whole-model in-place QDQ of the 14 GB Llama-3-8B pool swing between 0.68 and 0.80 of 8 TB/s from process to
process while the 137 GB Llama-3-70B pool holds 0.77?  (VERDICT round 3, weak #2.)

ONE process builds the same 224 weights several times, each time placed differently, and times three launches over each
set (HIP events, this stream): the read-only abs-max, the in-place FP8 QDQ (the bench's dominant kernel) and the fused
per-group INT4 pass.

    arena-first  : one 14 GB block, the FIRST device allocation of the process, tensors carved at 2 MiB-aligned offsets
    arena-packed : a second 14 GB block, tensors back to back (16-byte aligned only)
    blocks-a..   : 224 caching-allocator blocks per set (what bench.py's make_weights does), several sets one after the other
    after-137GB  : 224 blocks allocated after a 137 GB allocation was made and released to the driver (empty_cache)
    again-first  : the arena-first set once more at the end (drift of the box during the run)

`--pmc` : no timing loops -- every set's FP8 QDQ is launched exactly `--launches` times in set order, so that a rocprofv3
--pmc pass of this script gives per-dispatch counters that map back to the sets (dispatch i of mt_map_kernel belongs to set
i // launches).  tools/run/pool_placement_pmc.sh drives the passes; tools/run/pool_placement_sweep.sh the `--sweep` run
(grid shape x occupancy per kernel; needs the experiment library, MOQ_EXPERIMENTS=1 build.sh); profiles/r04_pool_placement.md
is the write-up.  (The round's other sweeps -- windows at a chosen distance, adjacent chunks per workgroup, split chunks --
used experiment kernels that were not kept; their results are profiles/r04_pool_sweep_box*.json.)
"""

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _moa_import  # noqa: E402

SHAPES_8B = [(4096, 4096), (1024, 4096), (1024, 4096), (4096, 4096), (14336, 4096), (14336, 4096), (4096, 14336)] * 32


def fill(t, seed):
    g = torch.Generator(device=t.device).manual_seed(seed)
    w = torch.randn(t.shape, generator=g, device=t.device, dtype=torch.float32) * 0.02
    t.copy_(w.to(torch.bfloat16))
    del w


def carve(arena, shapes, align_elems):
    out, off = [], 0
    for s in shapes:
        n = s[0] * s[1]
        off = -(-off // align_elems) * align_elems
        out.append(arena[off:off + n].view(s))
        off += n
    return out


def arena_elems(shapes, align_elems):
    off = 0
    for s in shapes:
        off = -(-off // align_elems) * align_elems + s[0] * s[1]
    return off


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=5, help="number of caching-allocator sets")
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--pmc", action="store_true")
    ap.add_argument("--sweep", action="store_true", help="grid-shape sweep of the FP8 QDQ over every set (experiment library)")
    ap.add_argument("--quick", action="store_true", help="--sweep: the short list of grid shapes")
    ap.add_argument("--launches", type=int, default=2)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    moa = _moa_import.load()
    from model_optimizer_amd.multi_tensor import SegmentTable

    sets = []  # (name, tensors)
    # 1. the arena is the first device allocation of the process
    a0 = torch.empty(arena_elems(SHAPES_8B, 1 << 20), dtype=torch.bfloat16, device=dev)
    w0 = carve(a0, SHAPES_8B, 1 << 20)
    for i, w in enumerate(w0):
        fill(w, 1234 + i)
    sets.append(("arena-first", w0))
    a1 = torch.empty(arena_elems(SHAPES_8B, 8), dtype=torch.bfloat16, device=dev)
    w1 = carve(a1, SHAPES_8B, 8)
    for i, w in enumerate(w1):
        fill(w, 1234 + i)
    sets.append(("arena-packed", w1))
    for k in range(args.sets):
        ws = []
        for i, s in enumerate(SHAPES_8B):
            t = torch.empty(s, dtype=torch.bfloat16, device=dev)
            fill(t, 1234 + i)
            ws.append(t)
        sets.append((f"blocks-{chr(97 + k)}", ws))
    # after a 137 GB allocation was made and given back to the driver
    torch.cuda.empty_cache()
    free, _ = torch.cuda.mem_get_info(dev)
    if free > 150e9:
        big = torch.empty(int(137e9) // 2, dtype=torch.bfloat16, device=dev)
        big.fill_(1.0)
        torch.cuda.synchronize()
        del big
        torch.cuda.empty_cache()
        ws = []
        for i, s in enumerate(SHAPES_8B):
            t = torch.empty(s, dtype=torch.bfloat16, device=dev)
            fill(t, 1234 + i)
            ws.append(t)
        sets.append(("after-137GB", ws))
    sets.append(("again-first", w0))

    n_elem = sum(s[0] * s[1] for s in SHAPES_8B)
    tabs = [(name, SegmentTable(ws, outputs=ws), SegmentTable(ws, outputs=ws, group_size=128)) for name, ws in sets]
    for _, t, _ in tabs:
        t.calibrate_amax()
    torch.cuda.synchronize()
    # power-state ramp
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        tabs[0][1].fake_quant_e4m3()
        torch.cuda.synchronize()

    if args.sweep:
        # the experiment library (MOQ_LIB_PATH=.../libmoquant_exp.so) reads the MOQ_TUNE_* knobs on every call: one process
        # compares grid shapes on the SAME allocations.  chunks per workgroup W: a workgroup visits chunks blockIdx + k * grid,
        # i.e. the launch sweeps W windows (n_chunks / W) x 16 KiB apart; LDS: dynamic LDS per workgroup (occupancy cap).
        settings = [("8 chunks / WG (rounds 1-3)", {"MOQ_TUNE_CHUNKS_PER_WG": "8", "MOQ_TUNE_COPY_GRID_CAP": "131072", "MOQ_TUNE_COPY_LDS": "0"}),
                    ("2 chunks / WG", {"MOQ_TUNE_CHUNKS_PER_WG": "2", "MOQ_TUNE_COPY_LDS": "0"}),
                    ("1 chunk / WG", {"MOQ_TUNE_CHUNKS_PER_WG": "1", "MOQ_TUNE_COPY_LDS": "0"}),
                    ("1 chunk / WG, 24 KiB LDS", {"MOQ_TUNE_CHUNKS_PER_WG": "1", "MOQ_TUNE_COPY_LDS": "24576"}),
                    ("1 chunk / WG, 32 KiB LDS (release)", {"MOQ_TUNE_CHUNKS_PER_WG": "1", "MOQ_TUNE_COPY_LDS": "32768"}),
                    ("1 chunk / WG, 40 KiB LDS", {"MOQ_TUNE_CHUNKS_PER_WG": "1", "MOQ_TUNE_COPY_LDS": "40960"})]
        knobs = ("MOQ_TUNE_CHUNKS_PER_WG", "MOQ_TUNE_COPY_GRID_CAP", "MOQ_TUNE_COPY_LDS", "MOQ_TUNE_READ_CHUNKS_PER_WG")

        def t_ms(fn):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / args.reps

        masks = {name: [torch.empty(w.shape, dtype=torch.bool, device=dev) for w in t.inputs] for name, t, _ in tabs[:1]}
        mtab = SegmentTable(tabs[0][1].inputs, outputs=masks[tabs[0][0]])
        rows = []
        for label, env in settings:
            for k in knobs:
                os.environ.pop(k, None)
            os.environ.update(env)
            row = {"order": label}
            for name, t, tg in tabs:
                row["fp8:" + name] = round(n_elem * 4 / t_ms(lambda: t.fake_quant_e4m3()) / 1e9 / 8.0, 4)
            for name, t, tg in tabs[:4]:
                row["int4g128:" + name] = round(n_elem * (4 + 4 / 128) / t_ms(lambda: tg.amax_qdq_int_group(4, False, False)) / 1e9 / 8.0, 4)
                row["mxfp4:" + name] = round(n_elem * 4 / t_ms(lambda: t.mx_fused_amax_convert(32, "E2M1")) / 1e9 / 8.0, 4)
            row["mask24:" + tabs[0][0]] = round(n_elem * 3 / t_ms(lambda: mtab.mask_2to4()) / 1e9 / 8.0, 4)
            rows.append(row)
            print(json.dumps(row), flush=True)
        for label, env in (("read-only abs-max, 8 chunks / WG (release)", {"MOQ_TUNE_READ_CHUNKS_PER_WG": "8"}),
                           ("read-only abs-max, 1 chunk / WG", {"MOQ_TUNE_READ_CHUNKS_PER_WG": "1"})):
            for k in knobs:
                os.environ.pop(k, None)
            os.environ.update(env)
            row = {"order": label}
            for name, t, tg in tabs:
                row["amax:" + name] = round(n_elem * 2 / t_ms(lambda: t.calibrate_amax()) / 1e9 / 8.0, 4)
            rows.append(row)
            print(json.dumps(row), flush=True)
        for k in knobs:
            os.environ.pop(k, None)
        if args.out:
            with open(args.out, "w") as f:
                json.dump(rows, f, indent=1)
        return

    if args.pmc:
        for _, t, _ in tabs:
            for _ in range(args.launches):
                t.fake_quant_e4m3()
            torch.cuda.synchronize()
        print(json.dumps({"pmc_order": [n for n, _, _ in tabs], "launches": args.launches}), flush=True)
        return

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.reps)]
        for a, b in ev:
            a.record()
            fn()
            b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in ev)
        return ms[len(ms) // 2], ms[0], ms[-1]

    rows = []
    for name, t, tg in tabs:
        ptrs = [w.data_ptr() for w in t.inputs] if hasattr(t, "inputs") else []
        am = timed(lambda: t.calibrate_amax())
        fq = timed(lambda: t.fake_quant_e4m3())
        gq = timed(lambda: tg.amax_qdq_int_group(4, False, False))
        row = {"set": name, "amax_ms": round(am[0], 4), "amax_TBs": round(n_elem * 2 / am[0] / 1e9, 3),
               "fp8_ms": round(fq[0], 4), "fp8_min_ms": round(fq[1], 4), "fp8_max_ms": round(fq[2], 4),
               "fp8_TBs": round(n_elem * 4 / fq[0] / 1e9, 3), "fp8_frac": round(n_elem * 4 / fq[0] / 1e9 / 8.0, 4),
               "int4g128_ms": round(gq[0], 4), "int4g128_frac": round(n_elem * (4 + 4 / 128) / gq[0] / 1e9 / 8.0, 4),
               "va_span_GiB": round((max(ptrs) - min(ptrs)) / 2**30, 2) if ptrs else None}
        rows.append(row)
        print(json.dumps(row), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
