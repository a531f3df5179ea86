"""Seeded random TensorQuantizer configurations on the device: this package against the REFERENCE's own TensorQuantizer (eager
path) on the same tensors -- calibrated amax and fake-quantized output, bit for bit.  Test infrastructure (needs the staged
reference, tools/stage_reference.sh); tests/test_gpu_reference_live.py runs a fixed slice of it.

    python tools/quantizer_fuzz.py [cases] [seed]"""
import json
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import _moa_import  # noqa: E402
import ref_shim  # noqa: E402

DEV = os.environ.get("MOQ_FUZZ_DEVICE", "cuda")  # "cpu": the host logic through tests/hostmem_backend.py (oracle-served C-ABI)


def load_package():
    """The package; on MOQ_FUZZ_DEVICE=cpu with its C-ABI served by the oracle (the CPU tier's stand-in), so that the same
    random cases run in the build container against the reference's CPU path."""
    moa = _moa_import.load()
    if DEV == "cpu":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hostmem_backend

        class _Setter:
            @staticmethod
            def setattr(obj, name, value):
                setattr(obj, name, value)

        hostmem_backend.install(_Setter, moa)
    return moa


def draw_extra(rng: random.Random) -> dict:
    """The corners beside draw()'s grid (opt-in: `extras`, so that the fixed-seed slice of the GPU suite keeps its cases):
    top-level dynamic quantizers (per tensor / per token), offsets (`bias`: static or dynamic, mean or max-min, over the token
    and batch axes of a [batch, heads, tokens, dim] tensor), a smoothing scale in front, constant amax, NaN in the input."""
    kind = rng.choice(["dynamic", "dynamic", "bias", "bias", "pre_quant_scale", "constant", "nan"])
    fmt = rng.choice(["int8", "fp8", "fp8", "int4"])
    nb = {"int8": 8, "int4": 4, "fp8": (4, 3)}[fmt]
    dtype = rng.choice(["bfloat16", "float16", "float32"])
    cfg = {"num_bits": nb, "axis": None}
    shape = [rng.randint(1, 40), rng.choice([8, 64, 128, 520])]
    pqs = False
    if kind == "dynamic":
        cfg["type"] = "dynamic"
        cfg["axis"] = rng.choice([None, None, 0, (0,)])  # axis 0 of [tokens, features]: one scale per token
        if rng.random() < 0.4:
            shape = [rng.randint(1, 3)] + shape
            cfg["axis"] = rng.choice([None, (0, 1)])
    elif kind == "bias":
        shape = [rng.randint(1, 3), rng.choice([1, 2, 4]), rng.randint(1, 24), rng.choice([16, 32, 64])]
        cfg["bias"] = {-2: None, -4: None, "type": rng.choice(["static", "dynamic"]), "method": rng.choice(["mean", "max_min"])}
        if rng.random() < 0.3:
            cfg["bias"] = {-2: None, "type": rng.choice(["static", "dynamic"])}
    elif kind == "pre_quant_scale":
        pqs = True
        cfg["axis"] = rng.choice([None, -1])
    elif kind == "constant":
        if rng.random() < 0.5:
            cfg.update(num_bits=(4, 3), use_constant_amax=True)
        else:
            cfg["constant_amax"] = rng.choice([0.5, 3.0, 448.0])
    return {"fmt": fmt, "dtype": dtype, "shape": shape, "gran": "extra_" + kind, "cfg": cfg, "seed": rng.randint(0, 1 << 30),
            "scale": rng.choice([0.02, 1.0, 30.0]), "specials": rng.random() < 0.2, "pqs": pqs, "nan": kind == "nan"}


def draw(rng: random.Random) -> dict:
    """One configuration: format, granularity, tensor shape and dtype, two calibration batches."""
    fmt = rng.choice(["int8", "int4", "int8", "fp8", "fp8", "int6"])
    nb = {"int8": 8, "int4": 4, "int6": 6, "fp8": (4, 3)}[fmt]
    dtype = rng.choice(["bfloat16", "float16", "float32"])
    rank = rng.choice([2, 2, 2, 3])
    cols = rng.choice([8, 16, 64, 128, 256, 520, 1000, 1024, 4096, 4100])
    shape = ([rng.randint(1, 5)] if rank == 3 else []) + [rng.randint(1, 300), cols]
    gran = rng.choice(["tensor", "tensor", "axis0", "axis_last", "block", "block", "block_dynamic", "block2d", "axis_pair",
                       "block_first"])
    cfg = {"num_bits": nb}
    if gran == "axis0":
        cfg["axis"] = 0
    elif gran == "axis_last":
        cfg["axis"] = -1
    elif gran == "block2d":  # tiles over the last two axes (any rank)
        cfg["axis"] = None
        cfg["block_sizes"] = {-1: rng.choice([16, 32, 128]), -2: rng.choice([4, 16, 64])}
    elif gran == "axis_pair":  # keep two axes of a rank-3 tensor, or the row axis of a matrix
        cfg["axis"] = (0, 2) if rank == 3 else (0,)
    elif gran == "block_first":  # blocks along a non-last axis
        cfg["axis"] = None
        cfg["block_sizes"] = {0: rng.choice([2, 4, 8])}
    elif gran.startswith("block"):
        cfg["axis"] = None
        cfg["block_sizes"] = {-1: rng.choice([16, 32, 64, 128])}
        if gran == "block_dynamic":
            cfg["block_sizes"]["type"] = "dynamic"
    else:
        cfg["axis"] = None
    if fmt != "fp8":
        cfg["narrow_range"] = rng.random() < 0.3
        cfg["unsigned"] = rng.random() < 0.15
    return {"fmt": fmt, "dtype": dtype, "shape": shape, "gran": gran, "cfg": cfg, "seed": rng.randint(0, 1 << 30),
            "scale": rng.choice([0.02, 1.0, 30.0]), "specials": rng.random() < 0.2}


def tensors(case):
    g = torch.Generator().manual_seed(case["seed"])
    dt = getattr(torch, case["dtype"])
    out = []
    for k in range(2):
        x = (torch.randn(*case["shape"], generator=g) * case["scale"] * (1.0 + k)).to(dt)
        if case["cfg"].get("unsigned"):
            x = x.abs()
        if case["specials"] and x.numel() > 8:
            flat = x.view(-1)
            flat[0], flat[1], flat[2] = 0.0, -0.0 if not case["cfg"].get("unsigned") else 0.0, flat.abs().max() * 4
        if case.get("nan") and k == 1 and x.numel() > 3:
            x.view(-1)[3] = float("nan")  # in the SECOND calibration batch and not in the quantized one
        out.append(x.to(DEV))
    return out


def run_one(make_quantizer, case):
    q = make_quantizer(case["cfg"]).to(DEV)
    xs = tensors(case)
    if case.get("pqs"):
        g = torch.Generator().manual_seed(case["seed"] + 7)
        q.pre_quant_scale = (torch.rand(case["shape"][-1], generator=g) * 2 + 0.25).to(xs[0].dtype).to(DEV)
    dynamic = case["gran"] == "block_dynamic" or case["cfg"].get("type") == "dynamic"
    constant = case["cfg"].get("use_constant_amax") or case["cfg"].get("constant_amax") is not None
    if constant:  # calibration leaves such a quantizer alone (model_calib.py:1132-1136); called directly it just quantizes
        pass
    elif not dynamic:
        q.disable_quant()
        q.enable_calib()
        for x in xs:
            q(x)
        q.load_calib_amax()
        if (case["cfg"].get("bias") or {}).get("type") == "static":
            q.load_calib_bias()  # (finish_stats_collection's order: model_calib.py:1155-1164)
        q.enable_quant()
        q.disable_calib()
    y = q(xs[0])
    amax = None if getattr(q, "_amax", None) is None else q._amax.detach().float().cpu()
    if getattr(q, "_bias_value", None) is not None:  # the calibrated offset rides along with the amax
        amax = torch.cat([amax.reshape(-1) if amax is not None else torch.zeros(0), q._bias_value.detach().float().cpu().reshape(-1)])
    return amax, y.detach().cpu()


def same_bits(a, b):
    if a is None or b is None:
        return a is None and b is None
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    iv = {torch.float32: torch.int32, torch.float16: torch.int16, torch.bfloat16: torch.int16}[a.dtype]
    return bool(((a.contiguous().view(iv) == b.contiguous().view(iv)) | (torch.isnan(a) & torch.isnan(b))).all())


def main(n_cases=200, seed=2025, verbose=True, extras=False):
    moa = load_package()
    ref_shim.install()
    from modelopt.torch.quantization.config import QuantizerAttributeConfig as RefCfg
    from modelopt.torch.quantization.nn import TensorQuantizer as RefTQ

    rng = random.Random(seed)
    stats = {"cases": 0, "equal": 0, "reference_refused": 0, "both_refused": 0, "ours_refused": [], "different": []}
    for i in range(n_cases):
        case = draw_extra(rng) if extras else draw(rng)
        try:
            want = run_one(lambda c: RefTQ(RefCfg(**c)), case)
        except Exception as e:  # the reference's own refusals (e.g. a block size that does not divide) are not cases
            want = e
        try:
            got = run_one(lambda c: moa.TensorQuantizer(moa.QuantizerAttributeConfig(**c)), case)
        except Exception as e:
            got = e
        stats["cases"] += 1
        if isinstance(want, Exception):
            stats["both_refused" if isinstance(got, Exception) else "reference_refused"] += 1
            why = f"{case['fmt']} {case['gran']}: {type(want).__name__}: {str(want)[:90]}"
            stats.setdefault("reference_refusals", {})[why] = stats.setdefault("reference_refusals", {}).get(why, 0) + 1
            continue
        if isinstance(got, Exception):
            stats["ours_refused"].append({"case": case, "error": f"{type(got).__name__}: {got}"[:200]})
            continue
        ok = same_bits(got[0], want[0]) and same_bits(got[1], want[1])
        if ok:
            stats["equal"] += 1
        else:
            stats["different"].append({"case": case, "amax_equal": same_bits(got[0], want[0]),
                                       "amax_shapes": [None if t is None else list(t.shape) for t in (got[0], want[0])],
                                       "y_dtypes": [str(got[1].dtype), str(want[1].dtype)]})
    if verbose:
        print(json.dumps({k: (v if not isinstance(v, list) else len(v)) for k, v in stats.items() if k != "reference_refusals"}))
        for why, n in sorted(stats.get("reference_refusals", {}).items(), key=lambda kv: -kv[1])[:10]:
            print(f"  reference refused {n}x: {why}")
        for d in stats["different"][:12] + stats["ours_refused"][:12]:
            print(json.dumps(d)[:600])
    return stats


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, int(sys.argv[2]) if len(sys.argv) > 2 else 2025,
         extras=len(sys.argv) > 3 and sys.argv[3] == "extras")
