"""Seeded random Hugging Face model shapes x architectures x dtypes through BOTH sparsification paths -- the reference's
`mts.sparsify(model, "sparse_magnitude")` and this package's `sparsity.sparsify` -- then, for half of the cases, through a
quantize() preset on top of the sparse weights and the masks' export (`mts.export` / `sparsity.export`: masks folded into the
weights): which modules were sparsified, every mask bit for bit, every weight after the export, every amax of the quantized
sparse model, the logits.  Build container only (the reference's CPU path against the CPU tier's stand-in).

    python tools/sparsity_fuzz.py [cases] [seed]"""
import copy
import json
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
os.environ.setdefault("MOQ_FUZZ_DEVICE", "cpu")
sys.path.insert(0, os.path.join(ROOT, "tools"))
from quantizer_fuzz import load_package  # noqa: E402

ARCHS = ["llama", "qwen2", "mistral", "opt", "gpt2", "phi3", "gemma2", "mixtral", "qwen3_moe"]
# (no SmoothQuant / AWQ preset: the reference's sparse module hands out `weight * mask` as a NEW tensor on every access, so the
# in-place fold `linear.weight.copy_(...)` of those algorithms (model_calib.py:1208-1216) writes into a temporary and is lost --
# inputs divided by the scale, weights not multiplied.  This package stores the masked weight and folds for real)
PRESETS = [None, None, "FP8_DEFAULT_CFG", "INT8_DEFAULT_CFG", "INT4_BLOCKWISE_WEIGHT_ONLY_CFG", "FP8_PER_CHANNEL_PER_TOKEN_CFG",
           "INT8_WEIGHT_ONLY_CFG"]


def draw(rng):
    heads = rng.choice([2, 4])
    hidden = heads * rng.choice([32, 64])
    return {"arch": rng.choice(ARCHS), "dtype": rng.choice(["bfloat16", "float16", "float32"]), "preset": rng.choice(PRESETS),
            "export": rng.random() < 0.5,
            "cfg": dict(hidden_size=hidden, intermediate_size=rng.choice([128, 256, 384]), num_hidden_layers=rng.choice([1, 2]),
                        num_attention_heads=heads, num_key_value_heads=rng.choice([1, heads] if heads == 2 else [1, 2, 4]),
                        vocab_size=96, max_position_embeddings=64)}


def reference(case, diff):
    import modelopt.torch.quantization as mtq
    import modelopt.torch.sparsity as mts

    dt = getattr(torch, case["dtype"])
    model = mts.sparsify(diff._model(dt, case["arch"]), "sparse_magnitude")
    masks = {n[: -len("._weight_mask")]: b.clone() for n, b in model.named_buffers() if n.endswith("_weight_mask")}
    out = {"masks": masks}
    batches = diff._batches()
    if case["preset"]:
        model = mtq.quantize(model, copy.deepcopy(getattr(mtq, case["preset"])), lambda m: [m(b) for b in batches])
        out["amax"] = {n: m._amax.detach().float().clone() for n, m in model.named_modules()
                       if type(m).__name__ == "TensorQuantizer" and m.is_enabled and getattr(m, "_amax", None) is not None}
    with torch.no_grad():
        out["logits"] = model(batches[0]).logits.clone()
    if case["export"] and not case["preset"]:
        model = mts.export(model)
        out["weights"] = {n: p.detach().clone() for n, p in model.state_dict().items()}  # (tied weights under both names)
    return out


def ours(case, diff, moa):
    dt = getattr(torch, case["dtype"])
    model = moa.sparsity.sparsify(diff._model(dt, case["arch"]), "sparse_magnitude")
    masks = {n[: -len("._weight_mask")]: b.clone() for n, b in model.named_buffers() if n.endswith("_weight_mask")}
    out = {"masks": masks}
    batches = diff._batches()
    with torch.no_grad():
        if case["preset"]:
            moa.quantize(model, copy.deepcopy(getattr(moa.model_quant, case["preset"])), lambda m: [m(b) for b in batches])
            out["amax"] = {n: m._amax.detach().float().clone() for n, m in model.named_modules()
                           if isinstance(m, moa.TensorQuantizer) and m.is_enabled and getattr(m, "_amax", None) is not None}
        out["logits"] = model(batches[0]).logits.clone()
    if case["export"] and not case["preset"]:
        model = moa.sparsity.export(model)
        out["weights"] = {n: p.detach().clone() for n, p in model.state_dict().items()}
    return out


def main(n=40, seed=2025, verbose=True):
    moa = load_package()
    import ref_shim
    import test_differential_cpu as diff

    ref_shim.install()
    rng = random.Random(seed)
    st = {"cases": 0, "equal": 0, "both_refused": 0, "reference_refused": {}, "ours_refused": [], "different": []}
    base_cfg = dict(diff.CFG)
    for _ in range(n):
        case = draw(rng)
        st["cases"] += 1
        diff.CFG.clear()
        diff.CFG.update(case["cfg"])
        try:
            want = reference(case, diff)
        except Exception as e:
            want = e
        try:
            got = ours(case, diff, moa)
        except Exception as e:
            got = e
        if isinstance(want, Exception):
            if isinstance(got, Exception):
                st["both_refused"] += 1
            else:
                why = f"{case['arch']} {case['preset']}: {type(want).__name__}: {str(want)[:90]}"
                st["reference_refused"][why] = st["reference_refused"].get(why, 0) + 1
            continue
        if isinstance(got, Exception):
            st["ours_refused"].append({"case": case, "error": f"{type(got).__name__}: {got}"[:260]})
            continue
        bad = []
        if sorted(want["masks"]) != sorted(got["masks"]):
            bad.append("sparsified modules " + str(sorted(set(want["masks"]) ^ set(got["masks"]))[:4]))
        bad += [f"mask {k}" for k, m in want["masks"].items() if k in got["masks"] and not torch.equal(got["masks"][k].bool().cpu(), m.bool())]
        for key in ("amax", "weights"):
            if key in want:
                bad += [f"{key} {k}" for k, a in want[key].items()
                        if k not in got[key] or got[key][k].shape != a.shape or not torch.equal(got[key][k].cpu(), a)]
                if key == "weights":  # (amax: the reference promotes static-block quantizers to a subclass the name filter skips)
                    bad += [f"{key} {k} (only here)" for k in got[key] if k not in want[key]]
        if not torch.equal(got["logits"].cpu(), want["logits"]):
            bad.append("logits")
        if bad:
            st["different"].append({"case": case, "first": bad[:4], "n_bad": len(bad)})
        else:
            st["equal"] += 1
    diff.CFG.clear()
    diff.CFG.update(base_cfg)
    if verbose:
        print("sparsity", json.dumps({k: (v if not isinstance(v, list) else len(v)) for k, v in st.items()})[:1500])
        for d in st["different"][:10] + st["ours_refused"][:10]:
            print("   ", json.dumps(d, default=str)[:600])
    return {"sparsity": st}


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 2025)
