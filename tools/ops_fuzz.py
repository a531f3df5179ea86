"""Seeded random calls of the path's FUNCTIONS on the device, this package against the reference's own eager implementations on
the same tensors, bit for bit: reduce_amax (any axis subset, keepdims on / off), reduce_block_amax / reduce_block_padding,
fake_tensor_quant / scaled_e4m3 with scalar, per-axis and broadcast amax, create_asp_mask (rank 1-4, planted ties),
FP8QTensor / MXFP4QTensor quantize + dequantize.  Test infrastructure (needs the staged reference).

    python tools/ops_fuzz.py [cases per family] [seed]"""
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
DT = {"bfloat16": torch.bfloat16, "float16": torch.float16, "float32": torch.float32}


def same_bits(a, b):
    if isinstance(a, (tuple, list)):
        return len(a) == len(b) and all(same_bits(x, y) for x, y in zip(a, b))
    a, b = a.detach().cpu(), b.detach().cpu()
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype in (torch.bool, torch.uint8, torch.int8, torch.float8_e4m3fn):
        return bool((a.view(torch.uint8) == b.view(torch.uint8)).all()) if a.dtype != torch.bool else bool((a == b).all())
    iv = {torch.float32: torch.int32, torch.float16: torch.int16, torch.bfloat16: torch.int16}[a.dtype]
    return bool(((a.contiguous().view(iv) == b.contiguous().view(iv)) | (torch.isnan(a) & torch.isnan(b))).all())


def rand_tensor(rng, shape, dtype, scale=None, ties=False):
    g = torch.Generator().manual_seed(rng.randint(0, 1 << 30))
    x = torch.randn(*shape, generator=g) * (scale if scale is not None else rng.choice([0.02, 1.0, 50.0]))
    if ties:  # few distinct magnitudes: equal |w| inside groups of four
        x = (x * 2).round() / 2
    return x.to(DT[dtype]).to(DEV)


def family_reduce_amax(rng, moa, ref):
    nd = rng.choice([1, 2, 3, 4])
    shape = [rng.choice([1, 2, 3, 5, 8, 16, 33, 64]) for _ in range(nd)]
    shape[-1] = rng.choice([8, 16, 40, 64, 128, 1000])
    dtype = rng.choice(list(DT))
    axis = rng.choice([None] + [tuple(sorted(rng.sample(range(nd), k))) for k in range(1, nd + 1)] + [-1])
    if isinstance(axis, tuple) and rng.random() < 0.3:
        axis = tuple(a - nd for a in axis)
    keep, squeeze = rng.random() < 0.7, rng.random() < 0.8
    x = rand_tensor(rng, shape, dtype)
    case = {"shape": shape, "dtype": dtype, "axis": axis, "keepdims": keep, "squeeze_scalar": squeeze}
    return case, (lambda: moa.ops.reduce_amax(x, axis=axis, keepdims=keep, squeeze_scalar=squeeze)), \
        (lambda: ref["core"].reduce_amax(x, axis=axis, keepdims=keep, squeeze_scalar=squeeze))


def family_block_amax(rng, moa, ref):
    nd = rng.choice([2, 2, 3, 4])
    blocks, shape = {}, []
    for d in range(nd):
        b = rng.choice([1, 2, 4, 8, 16, 32])
        shape.append(b * rng.randint(1, 6))
        if rng.random() < 0.6 or d == nd - 1:
            blocks[d if rng.random() < 0.5 else d - nd] = b
    if rng.random() < 0.3:  # ragged: padding first
        shape[-1] += rng.randint(1, 7)
    dtype = rng.choice(list(DT))
    x = rand_tensor(rng, shape, dtype)
    case = {"shape": shape, "dtype": dtype, "block_sizes": {str(k): v for k, v in blocks.items()}}

    def ours():
        p = moa.ops.reduce_block_padding(x, blocks)
        return p, moa.ops.reduce_block_amax(p, blocks)

    def theirs():
        p = ref["core"].reduce_block_padding(x, blocks)
        return p, ref["core"].reduce_block_amax(p, blocks)
    return case, ours, theirs


def family_fake_quant(rng, moa, ref):
    nd = rng.choice([2, 2, 3])
    shape = [rng.randint(1, 40) for _ in range(nd)]
    shape[-1] = rng.choice([8, 64, 256, 520, 1024])
    dtype = rng.choice(list(DT))
    x = rand_tensor(rng, shape, dtype)
    mode = rng.choice(["scalar", "axis0", "axis_last", "prefix", "apart"] if nd == 3 else ["scalar", "axis0", "axis_last"])
    xf = x.float().abs()
    am = {"scalar": lambda: xf.amax(), "axis0": lambda: xf.amax(dim=tuple(range(1, nd)), keepdim=True),
          "axis_last": lambda: xf.amax(dim=tuple(range(nd - 1)), keepdim=True),
          "prefix": lambda: xf.amax(dim=-1, keepdim=True), "apart": lambda: xf.amax(dim=1, keepdim=True)}[mode]()
    if rng.random() < 0.5:
        am = am.to(x.dtype)
    fp8 = rng.random() < 0.4
    nb, uns, narrow = rng.choice([4, 8, 6]), False, rng.random() < 0.3
    case = {"shape": shape, "dtype": dtype, "amax": mode, "amax_dtype": str(am.dtype), "fp8": fp8, "num_bits": nb, "narrow": narrow}
    if fp8:
        return case, (lambda: moa.ops.scaled_e4m3(x, am)), (lambda: ref["tq"].scaled_e4m3(x, am, None, 4, 3))
    return case, (lambda: moa.ops.fake_tensor_quant(x, am, nb, uns, narrow)), \
        (lambda: ref["tq"].fake_tensor_quant(x, am, None, nb, uns, narrow))


def family_asp_mask(rng, moa, ref):
    nd = rng.choice([1, 2, 2, 2, 3, 4])
    shape = {1: [rng.choice([16, 64, 4096])], 2: [rng.randint(1, 200), rng.choice([16, 64, 128, 1024, 4100 // 4 * 4])],
             3: [rng.randint(1, 9), rng.choice([16, 32, 64]), rng.randint(1, 5)],
             4: [rng.randint(1, 6), rng.choice([16, 32]), rng.randint(1, 3), rng.randint(1, 3)]}[nd]
    dtype = rng.choice(list(DT))
    x = rand_tensor(rng, shape, dtype, ties=rng.random() < 0.5)
    case = {"shape": shape, "dtype": dtype}
    return case, (lambda: moa.sparsity.create_asp_mask(x, "2:4 sparsity")), (lambda: ref["asp"](torch.nn.Parameter(x), "2:4 sparsity"))


def family_qtensor(rng, moa, ref):
    kind = rng.choice(["fp8_tensor", "fp8_axis", "fp8_block", "mxfp4"])
    dtype = rng.choice(list(DT))
    rows, cols = rng.randint(1, 80), rng.choice([32, 64, 256, 1024])
    x = rand_tensor(rng, [rows, cols], dtype)
    case = {"kind": kind, "shape": [rows, cols], "dtype": dtype}
    if kind == "mxfp4":
        def run(Q):
            qt, sc = Q.quantize(x, 32)
            return qt._quantized_data.view(torch.uint8), sc.view(torch.uint8), qt.dequantize(dtype=x.dtype, scale=sc, block_sizes={-1: 32})
        return case, (lambda: run(moa.qtensor.MXFP4QTensor)), (lambda: run(ref["mxfp4"]))
    kw = {"fp8_tensor": {}, "fp8_axis": {"axis": 0}, "fp8_block": {"block_sizes": {-1: rng.choice([16, 32, 128])}}}[kind]
    case["kw"] = {k: str(v) for k, v in kw.items()}

    def run8(Q):
        qt, sc = Q.quantize(x, **kw)
        return qt._quantized_data.view(torch.uint8), sc, qt.dequantize(dtype=x.dtype, scale=sc, **({"block_sizes": kw["block_sizes"]} if "block_sizes" in kw else {}))
    return case, (lambda: run8(moa.qtensor.FP8QTensor)), (lambda: run8(ref["fp8"]))


FAMILIES = {"reduce_amax": family_reduce_amax, "block_amax": family_block_amax, "fake_quant": family_fake_quant,
            "asp_mask": family_asp_mask, "qtensor": family_qtensor}


def main(n=120, seed=2025, verbose=True, families=None):
    moa = load_package()
    ref_shim.install()
    from modelopt.torch.quantization import tensor_quant as rtq
    from modelopt.torch.quantization.qtensor import FP8QTensor, MXFP4QTensor
    from modelopt.torch.quantization.utils import core_utils
    from modelopt.torch.sparsity.weight_sparsity.magnitude import create_asp_mask

    ref = {"core": core_utils, "tq": rtq, "asp": create_asp_mask, "fp8": FP8QTensor, "mxfp4": MXFP4QTensor}
    out = {}
    for fam, make in FAMILIES.items():
        if families and fam not in families:
            continue
        rng = random.Random(seed * 1000 + len(fam))
        st = {"cases": 0, "equal": 0, "both_refused": 0, "reference_refused": {}, "ours_refused": [], "different": []}
        for _ in range(n):
            case, ours, theirs = make(rng, moa, ref)
            st["cases"] += 1
            try:
                want = theirs()
            except Exception as e:
                want = e
            try:
                with moa.numerics.scale_math("device"):  # device vs device: the reference runs on this GPU too
                    got = ours()
            except Exception as e:
                got = e
            if isinstance(want, Exception):
                if isinstance(got, Exception):
                    st["both_refused"] += 1
                else:
                    why = f"{type(want).__name__}: {str(want)[:80]}"
                    st["reference_refused"][why] = st["reference_refused"].get(why, 0) + 1
                continue
            if isinstance(got, Exception):
                st["ours_refused"].append({"case": case, "error": f"{type(got).__name__}: {got}"[:200]})
            elif same_bits(got, want):
                st["equal"] += 1
            else:
                st["different"].append(case)
        out[fam] = st
        if verbose:
            print(fam, json.dumps({k: (v if not isinstance(v, list) else len(v)) for k, v in st.items()})[:400])
            for d in st["different"][:5] + st["ours_refused"][:5]:
                print("   ", json.dumps(d)[:400])
    return out


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 2025,
         families=sys.argv[3].split(",") if len(sys.argv) > 3 else None)
