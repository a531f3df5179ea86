"""End-to-end flow on a Hugging Face Llama with Llama-3-8B layer shapes (random init, `--layers` decoder layers):
quantize (FP8 W + A, FP8 KV cache, max calibration; or INT4-AWQ) on synthetic token batches, then build the checkpoint
tensors.  A timing / integration check at real shapes, not a parity test (those use the reference-run fixtures)."""

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _moa_import  # noqa: E402


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=2)
    ap.add_argument("--batches", type=int, default=4)
    ap.add_argument("--qformat", default="fp8", choices=["fp8", "int4_awq", "w4a8_awq", "mxfp4", "mxfp4_sq", "int8_sq",
                                                          "int8_mse", "fp8_mse", "int4_mse", "int4_awq_clip", "int4_awq_full",
                                                          "int8_percentile", "int8_entropy", "int4_gptq", "int4_gptq_layerwise", "int4_local_hessian", "mxfp4_gptq",
                                                          "fp8_gptq", "sparse_magnitude", "sparsegpt"],
                    help="the last rows: the other calibration algorithms of the path (MSE amax search, AWQ clip / full) and the "
                         "two sparsity modes, for wall-clock at real shapes")
    ap.add_argument("--arch", default="llama", choices=["llama", "mixtral"],
                    help="mixtral: 8 experts per layer with fused 3-D expert weights (Mixtral-8x7B layer shapes)")
    ap.add_argument("--search", default=None, choices=["auto", "gram", "gemm"], help="awq_lite search engine")
    ap.add_argument("--tie-margin", type=float, default=None, help="awq_lite re-scoring margin (inf: every candidate)")
    ap.add_argument("--dump", default=None, help="write every linear's AWQ score tables to this JSON file")
    ap.add_argument("--note", default=None)
    ap.add_argument("--defer-stats", default="auto", choices=["auto", "on", "off"],
                    help="max calibration: one statistics launch per decoder layer (calib.DeferredAmax); off = one per quantizer call")
    return ap.parse_args(argv)


def run(args, moa=None, dev=None) -> dict:
    """One flow; returns the result line.  `args`: parse_args() namespace (bench.py builds one for its `extra`)."""
    from transformers import LlamaConfig, LlamaForCausalLM, MixtralConfig, MixtralForCausalLM

    moa = moa or _moa_import.load()
    mq = moa.model_quant
    dev = dev or torch.device("cuda:0")
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=args.layers, num_attention_heads=32,
                      num_key_value_heads=8, vocab_size=128256, max_position_embeddings=8192, architectures=["LlamaForCausalLM"])
    torch.manual_seed(1234)
    if args.arch == "mixtral":
        cfg = MixtralConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=args.layers, num_attention_heads=32,
                            num_key_value_heads=8, vocab_size=32000, max_position_embeddings=8192, num_local_experts=8,
                            num_experts_per_tok=2, architectures=["MixtralForCausalLM"])
    with torch.device(dev):
        model = (MixtralForCausalLM if args.arch == "mixtral" else LlamaForCausalLM)(cfg).to(torch.bfloat16).eval()
    batches = [torch.randint(0, cfg.vocab_size, (8, 512), device=dev, generator=torch.Generator(device=dev).manual_seed(i))
               for i in range(args.batches)]

    def loop(m):
        with torch.no_grad():
            for b in batches:
                m(b)

    with torch.no_grad():
        model(batches[0])  # first touch: library GEMM selection, allocator growth (not part of the plain-loop figure)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(model)
    t_plain_host = time.perf_counter() - t0  # the host is done enqueuing here; the GPU usually is not
    torch.cuda.synchronize()
    t_plain = time.perf_counter() - t0
    host_times = {"plain_loop_host_s": round(t_plain_host, 3)}
    plain_loop = loop

    def loop(m):  # noqa: F811  (the calibration loop with a clock on its host side)
        t = time.perf_counter()
        plain_loop(m)
        host_times["calibration_loop_host_s"] = round(time.perf_counter() - t, 3)
    if args.qformat in ("sparse_magnitude", "sparsegpt"):
        t0 = time.perf_counter()
        moa.sparsity.sparsify(model, args.qformat, forward_loop=loop if args.qformat == "sparsegpt" else None)
        torch.cuda.synchronize()
        t_sp = time.perf_counter() - t0
        masks = [m._weight_mask for m in model.modules() if hasattr(m, "_weight_mask")]
        return {"arch": args.arch, "qformat": args.qformat, "layers": args.layers, "batches": args.batches, "tokens_per_batch": 4096,
                "plain_forward_loop_s": round(t_plain, 3), "sparsify_s": round(t_sp, 3), "masked_linears": len(masks),
                "kept_fraction": round(float(sum(int(m.sum()) for m in masks)) / max(1, sum(m.numel() for m in masks)), 4),
                **({"note": args.note} if args.note else {})}
    import copy as _copy

    if args.qformat in ("int8_percentile", "int8_entropy"):
        # the classic histogram flow: per-tensor INT8 input quantizers with histogram calibrators (collected on the device,
        # the threshold search on the host like the reference), weights max-calibrated
        hcfg = _copy.deepcopy(mq.INT8_DEFAULT_CFG)
        hcfg["quant_cfg"]["*input_quantizer"] = {"num_bits": 8, "axis": None, "calibrator": "histogram"}
        hcfg["algorithm"] = None
        t0 = time.perf_counter()
        moa.quantize(model, hcfg)
        moa.model_calib.histogram_calibrate(model, loop, method=args.qformat.split("_")[1])
        torch.cuda.synchronize()
        t_quant = time.perf_counter() - t0
        n_q = sum(1 for m in model.modules() if isinstance(m, moa.TensorQuantizer) and m.is_enabled)
        return {"arch": args.arch, "qformat": args.qformat, "layers": args.layers, "batches": args.batches, "tokens_per_batch": 4096,
                "plain_forward_loop_s": round(t_plain, 3), "quantize_s": round(t_quant, 3), "enabled_quantizers": n_q,
                **({"note": args.note} if args.note else {})}

    def with_alg(cfg, alg):
        c = _copy.deepcopy(cfg)
        c["algorithm"] = alg
        return c

    qcfg = {"int8_mse": lambda: with_alg(mq.INT8_DEFAULT_CFG, "mse"), "fp8_mse": lambda: with_alg(mq.FP8_DEFAULT_CFG, "mse"),
            "int4_mse": lambda: with_alg(mq.INT4_BLOCKWISE_WEIGHT_ONLY_CFG, "mse"),
            "int4_awq_clip": lambda: with_alg(mq.INT4_AWQ_CFG, "awq_clip"),
            "int4_awq_full": lambda: with_alg(mq.INT4_AWQ_CFG, "awq_full"),
            "int4_gptq": lambda: with_alg(mq.INT4_BLOCKWISE_WEIGHT_ONLY_CFG, {"method": "gptq"}),
            "int4_gptq_layerwise": lambda: with_alg(mq.INT4_BLOCKWISE_WEIGHT_ONLY_CFG, {"method": "gptq", "layerwise": {"enable": True}}),
            "fp8_gptq": lambda: with_alg(mq.FP8_DEFAULT_CFG, {"method": "gptq"}),
            "mxfp4_gptq": lambda: with_alg(mq.MXFP4_MLP_WEIGHT_ONLY_CFG, {"method": "gptq"}),
            "int4_local_hessian": lambda: with_alg(mq.INT4_BLOCKWISE_WEIGHT_ONLY_CFG,
                                                   {"method": "local_hessian", "fp8_scale_sweep": False, "block_size": 128})}.get(args.qformat)
    qcfg = qcfg() if qcfg else {"fp8": mq.update_quant_cfg_with_kv_cache_quant(mq.FP8_DEFAULT_CFG, mq.FP8_KV_CFG["quant_cfg"]),
            "int4_awq": mq.INT4_AWQ_CFG, "w4a8_awq": mq.W4A8_AWQ_BETA_CFG, "mxfp4": mq.MXFP4_DEFAULT_CFG, "mxfp4_sq": mq.MXFP4_SMOOTHQUANT_CFG,
            "int8_sq": mq.INT8_SMOOTHQUANT_CFG}[args.qformat]
    if args.search or args.tie_margin is not None:
        import copy

        qcfg = copy.deepcopy(qcfg)
        alg = qcfg["algorithm"] if isinstance(qcfg["algorithm"], dict) else {"method": qcfg["algorithm"]}
        if args.search:
            alg["search"] = args.search
        if args.tie_margin is not None:
            alg["tie_margin"] = args.tie_margin
        qcfg["algorithm"] = alg
    if getattr(args, "defer_stats", "auto") != "auto" and (qcfg.get("algorithm") == "max" or (isinstance(qcfg.get("algorithm"), dict) and qcfg["algorithm"].get("method") == "max")):
        import copy

        qcfg = copy.deepcopy(qcfg)
        alg = qcfg["algorithm"] if isinstance(qcfg["algorithm"], dict) else {"method": "max"}
        alg["defer_stats"] = args.defer_stats == "on"
        qcfg["algorithm"] = alg
    t0 = time.perf_counter()
    moa.quantize(model, qcfg, loop)
    torch.cuda.synchronize()
    t_quant = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.no_grad():
        logits = model(batches[0]).logits
    torch.cuda.synchronize()
    t_fq = time.perf_counter() - t0
    t0 = time.perf_counter()
    state = moa.export.export_state_dict(model, torch.bfloat16,
                                         (lambda: model(torch.ones([1, 2], dtype=torch.long, device=dev)))
                                         if args.qformat in ("int4_awq", "w4a8_awq", "int4_awq_clip", "int4_awq_full") else None)
    torch.cuda.synchronize()
    t_export = time.perf_counter() - t0
    n_q = sum(1 for m in model.modules() if isinstance(m, moa.TensorQuantizer) and m.is_enabled)
    awq = [m.awq_lite for m in model.modules() if hasattr(m, "awq_lite")]
    extra = {}
    if args.note:
        extra["note"] = args.note
    if awq and args.dump:
        os.makedirs(os.path.dirname(os.path.abspath(args.dump)), exist_ok=True)
        named = [(n, m) for n, m in model.named_modules() if hasattr(m, "awq_lite")]
        with open(args.dump, "w") as f:
            json.dump({"search": args.search, "tie_margin": args.tie_margin, "alphas": awq[0].alphas,
                       "planes": {str(k): v for k, v in moa.model_calib.GRAM_SCORE_PLANES.items()},
                       "linears": [{"name": n, "shape": list(m.weight.shape), "best_alpha": float(m.awq_lite.best_alpha),
                                    "loss": [float(v) for v in m.awq_lite.loss_buf.tolist()],
                                    "gram_loss": m.awq_lite.gram_loss, "contenders": m.awq_lite.contenders}
                                   for n, m in named]}, f)
    extra["quantize_stages_s"] = dict(moa.model_quant.QUANTIZE_STATS.get("stages_s") or {})
    extra["host_side"] = host_times  # loop wall-clock == host time: the HOST is the bottleneck of that loop
    if moa.model_calib.MAX_CALIBRATE_STATS:
        extra["max_calibrate_s"] = dict(moa.model_calib.MAX_CALIBRATE_STATS)
    if "gptq" in args.qformat:
        st = dict(moa.gptq.GPTQ_STATS)
        mse = st.pop("relative_mse", {})
        st["relative_mse_max"] = max(mse.values()) if mse else None
        extra["gptq_stats"] = st
    if awq:
        extra["awq_stats"] = dict(moa.model_calib.AWQ_LITE_STATS)
        alphas = [round(float(h.best_alpha), 1) for h in awq if h.best_alpha is not None]
        extra.update({"awq_rescored_linears": sum(1 for h in awq if h.contenders is not None),
                      "awq_rescored_candidates": sum(len(h.contenders) for h in awq if h.contenders is not None),
                      "awq_best_alpha_hist": {str(a): alphas.count(a) for a in sorted(set(alphas))}})
    return ({"arch": args.arch, "qformat": args.qformat, "layers": args.layers, "batches": args.batches, "tokens_per_batch": 4096,
                      "plain_forward_loop_s": round(t_plain, 3), "quantize_s": round(t_quant, 3),
                      "fake_quant_forward_s": round(t_fq, 3), "export_state_dict_s": round(t_export, 3),
                      "enabled_quantizers": n_q, "exported_tensors": len(state),
                      "logits_finite": bool(torch.isfinite(logits).all()),
                      "kv_cache_quant_algo": moa.export.hf_quant_config(model)["quantization"]["kv_cache_quant_algo"], **extra})


def main():
    print(json.dumps(run(parse_args())), flush=True)


if __name__ == "__main__":
    main()
