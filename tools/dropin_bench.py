"""What the drop-in buys, in wall seconds (TEST INFRASTRUCTURE: needs the reference -- staged archive or checkout -- and is
not part of bench.py's timed region).  VERDICT round 5, next #1b.

Decoder layers of Llama-3-8B's width (hidden 4096, MLP 14336, 32 / 8 heads; random init with a few outlier channels, small
vocabulary) on cuda:0, the same synthetic calibration batches for every row:

  eager        the reference's own mtq.quantize(...) as it runs on ROCm today (no extension: torch eager ops)
  kernels      + modelopt_plugin.install()                  (S1 / S3 / S6: our kernels under its per-call loops)
  algorithms   + modelopt_plugin.install(algorithms=True)   (S7: this package's fused flow on the reference's objects)
  mirror       this package's own quantize() on its own model classes (what bench.py's flows time)

for FP8 (W + A + KV, max), INT8 SmoothQuant, INT4-AWQ (awq_lite) and MXFP4, plus -- per row -- the reference's
export_hf_checkpoint (unified HF export) of the quantized model.  Every `algorithms` row is also CHECKED against the eager row
(amax / pre_quant_scale of every quantizer: identical, or for AWQ the alpha picks), so a fast wrong answer cannot be a row.

    python tools/dropin_bench.py --layers 2 --out gpurun_out/r06_dropin.json
"""

import argparse
import copy
import json
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import _moa_import  # noqa: E402
import ref_shim  # noqa: E402

DEV = "cuda:0"


def wide_llama(layers: int):
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(11)
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers, num_attention_heads=32,
                      num_key_value_heads=8, vocab_size=1024, max_position_embeddings=1024, architectures=["LlamaForCausalLM"])
    m = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    with torch.no_grad():  # massive channels like a trained model's: AWQ's candidates then stand apart
        hot = torch.randperm(4096, generator=torch.Generator().manual_seed(5))[:24]
        m.model.embed_tokens.weight[:, hot] *= 30.0
        for layer in m.model.layers:
            layer.input_layernorm.weight[hot] *= 8.0
            layer.post_attention_layernorm.weight[hot] *= 8.0
    return m.to(DEV)


def batches_for(n: int, rows: int, seq: int):
    return [torch.randint(0, 1024, (rows, seq), generator=torch.Generator().manual_seed(90 + i)).to(DEV) for i in range(n)]


def quantizer_state(model):
    out = {}
    for n, mod in model.named_modules():
        if "Quantizer" not in type(mod).__name__ or not hasattr(mod, "_disabled"):
            continue
        for k in ("_amax", "_pre_quant_scale"):
            v = getattr(mod, k, None)
            if isinstance(v, torch.Tensor):
                out[f"{n}.{k}"] = v.detach().float().cpu().clone()
    return out


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0, out


FORMATS = {
    "fp8": ("FP8_DEFAULT_CFG", True),
    "int8_sq": ("INT8_SMOOTHQUANT_CFG", False),
    "int4_awq": ("INT4_AWQ_CFG", False),
    "mxfp4": ("MXFP4_DEFAULT_CFG", False),
}


def reference_row(mtq, fmt, layers, batches, mode, export: bool):
    """One run of the reference's own quantize (+ export) under `mode` in ("eager", "kernels", "algorithms")."""
    from modelopt.torch.export import export_hf_checkpoint

    from model_optimizer_amd import modelopt_plugin

    preset, with_kv = FORMATS[fmt]
    cfg = copy.deepcopy(getattr(mtq, preset))
    if with_kv:
        cfg = mtq.update_quant_cfg_with_kv_cache_quant(cfg, copy.deepcopy(mtq.FP8_KV_CFG["quant_cfg"]))
    if fmt == "int4_awq":
        cfg["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "debug": True}
    model = wide_llama(layers)

    def loop(m):
        with torch.no_grad():
            for b in batches:
                m(b)

    loop(model)  # first touch (library GEMM selection, allocator growth) outside every row's clock
    modelopt_plugin.uninstall()
    if mode != "eager":
        modelopt_plugin.install(algorithms=(mode == "algorithms"))
    modelopt_plugin.STATS.clear()
    try:
        t_q, q = timed(lambda: mtq.quantize(model, cfg, loop))
        calls = dict(modelopt_plugin.STATS)
        row = {"quantize_s": round(t_q, 3), "seam_calls_in_quantize": sum(v for k, v in calls.items() if "fallback" not in k),
               "fallbacks": sorted(k for k in calls if "fallback" in k),
               "s7": {k: v for k, v in calls.items() if k.startswith("S7")}}
        if mode == "algorithms" and fmt == "int4_awq":
            row["awq_stats"] = awq_stats()
        state = quantizer_state(q)
        alphas = {n: round(float(mod.awq_lite.best_alpha), 2) for n, mod in q.named_modules() if hasattr(mod, "awq_lite")}
        if export:
            with tempfile.TemporaryDirectory() as d:
                t_e, _ = timed(lambda: export_hf_checkpoint(q, export_dir=d))
            row["export_s"] = round(t_e, 3)
    finally:
        modelopt_plugin.uninstall()
    del q, model
    torch.cuda.empty_cache()
    return row, state, alphas


def awq_stats():
    """Where the fused AWQ flow spent the call (model_calib.AWQ_LITE_STATS of the run that just ended)."""
    from model_optimizer_amd import model_calib

    st = model_calib.AWQ_LITE_STATS
    return {k: st.get(k) for k in ("passes", "replayed_passes", "layer_local", "stages_s", "store_dropped", "rescored_linears",
                                   "rescored_candidates") if k in st}


def mirror_row(moa, fmt, layers, batches, export: bool):
    preset, with_kv = FORMATS[fmt]
    mq = moa.model_quant
    cfg = copy.deepcopy(getattr(mq, preset))
    if with_kv:
        cfg = mq.update_quant_cfg_with_kv_cache_quant(cfg, mq.FP8_KV_CFG["quant_cfg"])
    model = wide_llama(layers)

    def loop(m):
        with torch.no_grad():
            for b in batches:
                m(b)

    loop(model)
    with moa.numerics.scale_math("device"), torch.no_grad():
        t_q, _ = timed(lambda: moa.quantize(model, cfg, loop))
        row = {"quantize_s": round(t_q, 3)}
        if fmt == "int4_awq":
            row["awq_stats"] = awq_stats()
        state = quantizer_state(model)
        alphas = {n: round(float(mod.awq_lite.best_alpha), 2) for n, mod in model.named_modules() if hasattr(mod, "awq_lite")}
        if export:  # the same deliverable as the reference rows: the checkpoint DIRECTORY (tensors packed, files written)
            with tempfile.TemporaryDirectory() as d:
                t_e, _ = timed(lambda: moa.export.export_hf_checkpoint(
                    model, torch.bfloat16, export_dir=d,
                    dummy_forward_fn=lambda: model(torch.ones([1, 2], dtype=torch.long, device=DEV))))
            row["export_s"] = round(t_e, 3)
    del model
    torch.cuda.empty_cache()
    return row, state, alphas


def compare(fmt, base_state, base_alphas, state, alphas):
    """How a row's result relates to the eager row's."""
    keys = sorted(base_state)
    if sorted(state) != keys:
        return {"same_keys": False, "only_eager": sorted(set(keys) - set(state))[:4], "only_here": sorted(set(state) - set(keys))[:4]}
    same = [k for k in keys if base_state[k].shape == state[k].shape and torch.equal(base_state[k], state[k])]
    out = {"same_keys": True, "tensors": len(keys), "identical": len(same)}
    worst = 0.0
    for k in keys:
        if k not in same and base_state[k].shape == state[k].shape:
            worst = max(worst, ((base_state[k] - state[k]).abs() / base_state[k].abs().clamp_min(1e-30)).max().item())
    out["worst_relative_difference"] = worst
    if base_alphas:
        out["alpha_picks_equal"] = sum(1 for n in base_alphas if alphas.get(n) == base_alphas[n])
        out["alpha_picks"] = len(base_alphas)
    return out


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--batches", type=int, default=8)
    ap.add_argument("--rows", type=int, default=4)
    ap.add_argument("--seq", type=int, default=512)
    ap.add_argument("--formats", default="fp8,int8_sq,int4_awq,mxfp4")
    ap.add_argument("--modes", default="eager,kernels,algorithms,mirror")
    ap.add_argument("--no-export", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args(argv)
    moa = _moa_import.load()
    ref_shim.install()
    import modelopt.torch.quantization as mtq

    batches = batches_for(args.batches, args.rows, args.seq)
    result = {"what": f"{args.layers} decoder layers of Llama-3-8B width (bf16), {args.batches} x {args.rows} x {args.seq} calibration tokens, cuda:0",
              "reference": ref_shim.reference_source(), "formats": {}}
    for fmt in args.formats.split(","):
        rows, base = {}, None
        for mode in args.modes.split(","):
            if fmt == "mxfp4" and mode == "eager":
                rows[mode] = {"skipped": "the reference has no eager MX implementation (its CUDA extension only)"}
                continue
            try:
                if mode == "mirror":
                    row, state, alphas = mirror_row(moa, fmt, args.layers, batches, not args.no_export)
                else:
                    row, state, alphas = reference_row(mtq, fmt, args.layers, batches, mode, not args.no_export)
            except Exception as e:  # a row that cannot run is a row that says so
                rows[mode] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
                torch.cuda.empty_cache()
                continue
            if base is None:
                base = (mode, state, alphas)
            else:
                row[f"vs_{base[0]}"] = compare(fmt, base[1], base[2], state, alphas)
            rows[mode] = row
            print(json.dumps({fmt: {mode: row}}), flush=True)
        result["formats"][fmt] = rows
    line = json.dumps(result)
    print(line)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            f.write(line + "\n")
    return result


if __name__ == "__main__":
    main()
