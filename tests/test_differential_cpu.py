"""LIVE differential test against the reference on CPU (build container only; skipped where /root/reference is absent):
the same tiny random Llama goes through the reference's `mtq.quantize` + `export_hf_checkpoint` and through this
package's `quantize` + `export_state_dict` (host on CPU via tests/hostmem_backend.py), preset by preset; every
quantizer amax and every checkpoint tensor must be identical.  No stored fixture is involved, so a preset or a model
shape can be added here in one line."""

import copy
import os
import sys
import tempfile

import pytest
import torch

import _moa_import
import hostmem_backend
from conftest import GOLDEN

moa = _moa_import.load()
sys.path.insert(0, GOLDEN)
import ref_shim  # noqa: E402

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference checkout not present (GPU box)")

CFG = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
           vocab_size=96, max_position_embeddings=64)


def _model(dtype, arch="llama"):
    from transformers import LlamaConfig, LlamaForCausalLM, MixtralConfig, MixtralForCausalLM

    torch.manual_seed(7)
    if arch == "llama-eager":  # eager attention: the KV quantizers are reached through eager_attention_forward
        cfg = LlamaConfig(architectures=["LlamaForCausalLM"], **CFG)
        cfg._attn_implementation = "eager"
        return LlamaForCausalLM(cfg).to(dtype).eval()
    if arch == "llama-ragged":  # Cin = 192 / 320: not multiples of the INT4 block (128) -> zero-padded last block
        cfg = LlamaConfig(architectures=["LlamaForCausalLM"], **{**CFG, "hidden_size": 192, "intermediate_size": 320})
        return LlamaForCausalLM(cfg).to(dtype).eval()
    if arch == "opt":  # BASELINE configs[0] family: biased linears, LayerNorm, learned positions
        from transformers import OPTConfig, OPTForCausalLM

        cfg = OPTConfig(hidden_size=128, ffn_dim=256, num_hidden_layers=2, num_attention_heads=4, vocab_size=96,
                        max_position_embeddings=64, word_embed_proj_dim=128, architectures=["OPTForCausalLM"])
        return OPTForCausalLM(cfg).to(dtype).eval()
    if arch == "qwen2":  # biased q / k / v projections, tied embeddings by default
        from transformers import Qwen2Config, Qwen2ForCausalLM

        return Qwen2ForCausalLM(Qwen2Config(architectures=["Qwen2ForCausalLM"], tie_word_embeddings=True, **CFG)).to(dtype).eval()
    if arch == "gpt2":  # Conv1D projections ([in, out] weights), LayerNorm, tied embeddings
        from transformers import GPT2Config, GPT2LMHeadModel

        return GPT2LMHeadModel(GPT2Config(n_embd=128, n_layer=2, n_head=4, vocab_size=96, n_positions=64,
                                          architectures=["GPT2LMHeadModel"])).to(dtype).eval()
    if arch in ("mistral", "phi3", "gemma2", "qwen3_moe", "falcon"):
        import transformers as tf

        if arch == "mistral":
            return tf.MistralForCausalLM(tf.MistralConfig(architectures=["MistralForCausalLM"], **CFG)).to(dtype).eval()
        if arch == "phi3":  # fused qkv_proj / gate_up_proj linears
            return tf.Phi3ForCausalLM(tf.Phi3Config(architectures=["Phi3ForCausalLM"], pad_token_id=0, **CFG)).to(dtype).eval()
        if arch == "gemma2":  # soft-capped attention, tied embeddings
            return tf.Gemma2ForCausalLM(tf.Gemma2Config(architectures=["Gemma2ForCausalLM"], head_dim=32, **CFG)).to(dtype).eval()
        if arch == "falcon":  # FalconLinear (nn.Linear subclass), fused query_key_value
            return tf.FalconForCausalLM(tf.FalconConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, vocab_size=96,
                                                         architectures=["FalconForCausalLM"])).to(dtype).eval()
        return tf.Qwen3MoeForCausalLM(tf.Qwen3MoeConfig(architectures=["Qwen3MoeForCausalLM"], moe_intermediate_size=64,
                                                        num_experts=4, num_experts_per_tok=2, head_dim=32, **CFG)).to(dtype).eval()
    if arch in _MORE_ARCHITECTURES:
        return _MORE_ARCHITECTURES[arch]().to(dtype).eval()
    if arch == "mixtral":
        cfg = MixtralConfig(architectures=["MixtralForCausalLM"], num_local_experts=4, num_experts_per_tok=2, **CFG)
        return MixtralForCausalLM(cfg).to(dtype).eval()
    return LlamaForCausalLM(LlamaConfig(architectures=["LlamaForCausalLM"], **CFG)).to(dtype).eval()


def _tf(cls, cfg_cls, **extra):
    import transformers as tf

    return lambda: getattr(tf, cls)(getattr(tf, cfg_cls)(architectures=[cls], **{**CFG, **extra}))


# round 5: thirty-four more decoder families through the same two flows (attention modules patched on the fly for the KV quantizers,
# q / k norms, parallel attention + MLP blocks, LayerNorm with bias, multi-head latent attention, a head whose checkpoint name
# differs from its module name)
_MORE_ARCHITECTURES = {
    "qwen3": _tf("Qwen3ForCausalLM", "Qwen3Config", head_dim=32), "gemma": _tf("GemmaForCausalLM", "GemmaConfig", head_dim=32),
    "starcoder2": _tf("Starcoder2ForCausalLM", "Starcoder2Config"), "olmo2": _tf("Olmo2ForCausalLM", "Olmo2Config"),
    "cohere": _tf("CohereForCausalLM", "CohereConfig"), "phi": _tf("PhiForCausalLM", "PhiConfig"),
    "granite": _tf("GraniteForCausalLM", "GraniteConfig"), "glm": _tf("GlmForCausalLM", "GlmConfig", head_dim=32, pad_token_id=0),
    # GPT-NeoX: the head is `lm_head` in the module tree and `embed_out` in the checkpoint (a class-specific renaming)
    # attention classes outside the attention interface: the KV quantizers are written into the class's own products
    "gptj": lambda: __import__("transformers").GPTJForCausalLM(__import__("transformers").GPTJConfig(
        architectures=["GPTJForCausalLM"], n_embd=128, n_layer=2, n_head=4, vocab_size=96, n_positions=64, rotary_dim=16)),
    "codegen": lambda: __import__("transformers").CodeGenForCausalLM(__import__("transformers").CodeGenConfig(
        architectures=["CodeGenForCausalLM"], n_embd=128, n_layer=2, n_head=4, vocab_size=96, n_positions=64, rotary_dim=16)),
    "mpt": lambda: __import__("transformers").MptForCausalLM(__import__("transformers").MptConfig(
        architectures=["MptForCausalLM"], d_model=128, n_heads=4, n_layers=2, vocab_size=96, max_seq_len=64)),
    "stablelm": _tf("StableLmForCausalLM", "StableLmConfig"), "nemotron": _tf("NemotronForCausalLM", "NemotronConfig"),
    "glm4": _tf("Glm4ForCausalLM", "Glm4Config", head_dim=32, pad_token_id=0), "exaone4": _tf("Exaone4ForCausalLM", "Exaone4Config"),
    "ernie4_5": _tf("Ernie4_5ForCausalLM", "Ernie4_5Config"),
    "gpt_bigcode": lambda: __import__("transformers").GPTBigCodeForCausalLM(__import__("transformers").GPTBigCodeConfig(
        architectures=["GPTBigCodeForCausalLM"], n_embd=128, n_layer=2, n_head=4, vocab_size=96, n_positions=64)),
    # more mixture-of-experts families on the generic fused-experts rule
    "qwen2_moe": _tf("Qwen2MoeForCausalLM", "Qwen2MoeConfig", moe_intermediate_size=64, shared_expert_intermediate_size=64, num_experts=4, num_experts_per_tok=2),
    "olmoe": _tf("OlmoeForCausalLM", "OlmoeConfig", num_experts=4, num_experts_per_tok=2),
    "granitemoe": _tf("GraniteMoeForCausalLM", "GraniteMoeConfig", num_local_experts=4, num_experts_per_tok=2),
    "phimoe": _tf("PhimoeForCausalLM", "PhimoeConfig", num_local_experts=4, num_experts_per_tok=2),
    "helium": _tf("HeliumForCausalLM", "HeliumConfig", head_dim=32), "arcee": _tf("ArceeForCausalLM", "ArceeConfig"),
    "apertus": _tf("ApertusForCausalLM", "ApertusConfig"), "seed_oss": _tf("SeedOssForCausalLM", "SeedOssConfig", head_dim=32),
    "hunyuan": _tf("HunYuanDenseV1ForCausalLM", "HunYuanDenseV1Config", head_dim=32), "gemma3": _tf("Gemma3ForCausalLM", "Gemma3TextConfig", head_dim=32),
    "bitnet": _tf("BitNetForCausalLM", "BitNetConfig"),
    "glm4_moe": _tf("Glm4MoeForCausalLM", "Glm4MoeConfig", moe_intermediate_size=64, n_routed_experts=4, num_experts_per_tok=2,
                    n_shared_experts=1, first_k_dense_replace=1, head_dim=32, n_group=1, topk_group=1),
    "ernie4_5_moe": _tf("Ernie4_5_MoeForCausalLM", "Ernie4_5_MoeConfig", moe_intermediate_size=64, moe_num_experts=4, moe_k=2),
    "dots1": _tf("Dots1ForCausalLM", "Dots1Config", moe_intermediate_size=64, n_routed_experts=4, num_experts_per_tok=2, n_shared_experts=1,
                 first_k_dense_replace=1, n_group=1, topk_group=1),
    "minimax": _tf("MiniMaxForCausalLM", "MiniMaxConfig", num_local_experts=4, num_experts_per_tok=2, head_dim=32),
    "gpt_neox": lambda: __import__("transformers").GPTNeoXForCausalLM(__import__("transformers").GPTNeoXConfig(
        architectures=["GPTNeoXForCausalLM"], hidden_size=CFG["hidden_size"], intermediate_size=CFG["intermediate_size"],
        num_hidden_layers=CFG["num_hidden_layers"], num_attention_heads=CFG["num_attention_heads"], vocab_size=96, max_position_embeddings=64)),
    "deepseek_v3": lambda: __import__("transformers").DeepseekV3ForCausalLM(__import__("transformers").DeepseekV3Config(
        architectures=["DeepseekV3ForCausalLM"], hidden_size=CFG["hidden_size"], intermediate_size=CFG["intermediate_size"],
        moe_intermediate_size=64, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=4, n_routed_experts=4,
        num_experts_per_tok=2, n_shared_experts=1, first_k_dense_replace=1, vocab_size=96, max_position_embeddings=64, q_lora_rank=32,
        kv_lora_rank=32, qk_rope_head_dim=16, qk_nope_head_dim=16, v_head_dim=32, n_group=1, topk_group=1)),
}


def _batches():
    return [torch.randint(0, CFG["vocab_size"], (3, 24), generator=torch.Generator().manual_seed(40 + i)) for i in range(3)]


def _reference_run(preset, dtype, with_kv, arch="llama", algorithm=None, device=None, edit=None, export=True, inspect=None):
    """`inspect`: called with the quantized model before anything is exported (search tables, module state).
    `device`: None = host tensors (this tier); "cuda" = the same run with the model and the batches on the GPU
    (tests/test_gpu_reference_live.py: the reference's eager path with device tensors, staged or checked out).
    `edit`: a callable applied to the preset's copy before the KV-cache entries are merged (per-layer overrides)."""
    ref_shim.install()
    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint
    from safetensors import safe_open

    model = _model(dtype, arch)
    if device is not None:
        model = model.to(device)
    cfg = copy.deepcopy(getattr(mtq, preset))
    if algorithm is not None:
        cfg["algorithm"] = copy.deepcopy(algorithm)
    if edit is not None:
        edit(cfg)
    if with_kv:  # True: FP8 key / value quantizers; "affine": the same with a per-head per-channel offset
        kv = mtq.FP8_AFFINE_KV_CFG if with_kv == "affine" else mtq.FP8_KV_CFG
        if with_kv == "cast":  # configs/ptq/units/kv_fp8_cast.yaml (hf_ptq.py's default KV format): amax fixed at 448
            kv = {"quant_cfg": [{"quantizer_name": "*[kv]_bmm_quantizer", "cfg": {"num_bits": (4, 3), "axis": None, "use_constant_amax": True}}]}
        if with_kv == "int8":  # an INT8 KV cache: calibrated like any other, named "INT8" in the tables (quant_utils.py:453-456)
            kv = {"quant_cfg": [{"quantizer_name": "*[kv]_bmm_quantizer", "cfg": {"num_bits": 8, "axis": None}}]}
        cfg = mtq.update_quant_cfg_with_kv_cache_quant(cfg, copy.deepcopy(kv["quant_cfg"]))
    batches = [b.to(device) if device is not None else b for b in _batches()]
    loop = (lambda m: [m(b) for b in batches]) if cfg.get("algorithm") else None
    q = mtq.quantize(model, cfg, loop)
    if inspect is not None:
        inspect(q)
    amax = {n: m._amax.detach().float().cpu().clone() for n, m in q.named_modules()
            if type(m).__name__ == "TensorQuantizer" and m.is_enabled and getattr(m, "_amax", None) is not None}
    logits = None
    if "MXFP" not in preset:  # the reference's MX fake quant has no CPU implementation
        with torch.no_grad():
            logits = q(batches[0]).logits.cpu().clone()
    out = {"__logits__": logits}
    if arch == "llama-ragged" or not export:  # the reference's INT4 packer indexes past its scale tensor for a padded last block
        return amax, out
    with tempfile.TemporaryDirectory() as d:
        export_hf_checkpoint(q, export_dir=d)
        with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
            for k in f.keys():
                out[k] = f.get_tensor(k)
        import json

        # the two JSON tables a deployment framework reads next to the tensors
        hfq = json.load(open(os.path.join(d, "hf_quant_config.json")))["quantization"] if os.path.exists(os.path.join(d, "hf_quant_config.json")) else None
        qc = json.load(open(os.path.join(d, "config.json"))).get("quantization_config")
        out["__quant_json__"] = (hfq, qc)
    return amax, out


def _our_run(preset, dtype, with_kv, arch="llama", algorithm=None, device=None, edit=None, export=True, inspect=None):
    mq = moa.model_quant
    model = _model(dtype, arch)
    if device is not None:
        model = model.to(device)
    cfg = copy.deepcopy(getattr(mq, preset))
    if algorithm is not None:
        cfg["algorithm"] = copy.deepcopy(algorithm)
    if edit is not None:
        edit(cfg)
    if with_kv:
        kv = {"affine": mq.FP8_AFFINE_KV_CFG, "cast": mq.FP8_CAST_KV_CFG,
              "int8": {"quant_cfg": {"*[kv]_bmm_quantizer": {"num_bits": 8, "axis": None, "enable": True}}}}.get(with_kv, mq.FP8_KV_CFG)
        cfg = mq.update_quant_cfg_with_kv_cache_quant(cfg, kv["quant_cfg"])
    batches = [b.to(device) if device is not None else b for b in _batches()]
    with torch.no_grad():
        moa.quantize(model, cfg, (lambda m: [m(b) for b in batches]) if cfg.get("algorithm") else None)
    if inspect is not None:
        inspect(model)
    amax = {n: m._amax.detach().float().cpu().clone() for n, m in model.named_modules()
            if isinstance(m, moa.TensorQuantizer) and m.is_enabled and getattr(m, "_amax", None) is not None}
    with torch.no_grad():
        logits = model(batches[0]).logits.cpu().clone()
    if arch == "llama-ragged" or not export:
        return amax, {"__logits__": logits}
    state = moa.export.export_state_dict(model, dtype, lambda: model(torch.ones([1, 2], dtype=torch.long, device=device)))
    state["__logits__"] = logits
    quant = moa.export.hf_quant_config(model)
    state["__quant_json__"] = (quant["quantization"], moa.export.convert_hf_quant_config_format(quant))
    return amax, state


SQ_HALF = {"method": "smoothquant", "alpha": 0.5}


def _assert_same_quant_json(ours, ref, what=""):
    """hf_quant_config.json's quantization table (algorithm, group size, exclude_modules wildcards, routers, per-layer
    table, KV cache) and config.json's quantization_config, up to each library's own producer entry."""
    assert ours[0] == ref[0], f"{what}: hf_quant_config {ours[0]} vs {ref[0]}"
    ours_qc, ref_qc = dict(ours[1]), dict(ref[1])
    ours_qc.pop("producer", None), ref_qc.pop("producer", None)
    assert ours_qc == ref_qc, f"{what}: quantization_config {ours_qc} vs {ref_qc}"


@pytest.mark.parametrize("preset,dtype,with_kv,arch,algorithm", [
    ("FP8_DEFAULT_CFG", torch.bfloat16, False, "llama", None), ("FP8_DEFAULT_CFG", torch.float32, True, "llama", None),
    ("FP8_DEFAULT_CFG", torch.float16, True, "llama", None),
    ("INT8_SMOOTHQUANT_CFG", torch.bfloat16, False, "llama", None), ("INT8_SMOOTHQUANT_CFG", torch.float32, False, "llama", None),
    ("INT8_SMOOTHQUANT_CFG", torch.bfloat16, False, "llama", SQ_HALF),
    ("INT8_DEFAULT_CFG", torch.bfloat16, False, "llama", None), ("INT8_WEIGHT_ONLY_CFG", torch.float16, False, "llama", None),
    ("INT8_DEFAULT_CFG", torch.float32, False, "mixtral", None),
    ("FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG", torch.bfloat16, False, "llama", None), ("MXFP4_DEFAULT_CFG", torch.bfloat16, False, "llama", None),
    ("MXFP4_DEFAULT_CFG", torch.float16, False, "llama", None), ("INT4_BLOCKWISE_WEIGHT_ONLY_CFG", torch.bfloat16, False, "llama", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "mixtral", None), ("FP8_DEFAULT_CFG", torch.float32, False, "mixtral", None),
    ("MXFP4_DEFAULT_CFG", torch.bfloat16, False, "mixtral", None),
    ("INT8_DEFAULT_CFG", torch.float32, False, "opt", None), ("INT8_SMOOTHQUANT_CFG", torch.float16, False, "opt", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "opt", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "qwen2", None), ("INT8_SMOOTHQUANT_CFG", torch.float32, False, "qwen2", None),
    ("FP8_DEFAULT_CFG", torch.float32, False, "gpt2", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "mistral", None), ("FP8_DEFAULT_CFG", torch.bfloat16, True, "phi3", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "gemma2", None), ("FP8_DEFAULT_CFG", torch.bfloat16, True, "qwen3_moe", None),
    ("INT8_DEFAULT_CFG", torch.float32, False, "qwen3_moe", None), ("FP8_DEFAULT_CFG", torch.bfloat16, False, "falcon", None),
    ("FP8_DEFAULT_CFG", torch.float32, True, "llama-eager", None), ("FP8_DEFAULT_CFG", torch.float16, True, "mixtral", None),
    # AWQ-lite search + resmooth + norm fusion + INT4 packing, end to end
    ("INT4_AWQ_CFG", torch.bfloat16, False, "llama", None), ("INT4_AWQ_CFG", torch.bfloat16, False, "opt", None),
    ("INT4_AWQ_CFG", torch.bfloat16, False, "qwen2", None), ("INT4_AWQ_CFG", torch.bfloat16, False, "phi3", None),
    ("INT4_AWQ_CFG", torch.bfloat16, False, "gpt2", None), ("INT4_AWQ_CFG", torch.bfloat16, False, "gemma2", None),
    ("INT4_AWQ_CFG", torch.float16, True, "mistral", None), ("INT4_AWQ_CFG", torch.bfloat16, False, "llama-ragged", None),
    # W4A8 AWQ: INT4 -> FP8 sequential weight quantizers, per-channel input calibration collapsed after the search
    ("W4A8_AWQ_BETA_CFG", torch.bfloat16, False, "llama", None), ("W4A8_AWQ_BETA_CFG", torch.float16, True, "qwen2", None),
    ("MXFP8_DEFAULT_CFG", torch.bfloat16, False, "llama", None), ("MXFP8_DEFAULT_CFG", torch.float16, True, "qwen2", None),
    ("MXFP8_DEFAULT_CFG", torch.float32, False, "opt", None),
    # the presets of round 4 and more formats on fused expert containers (Mixtral / Qwen3-MoE); with a KV-cache config the
    # algorithm becomes "max", so W4A8_MXFP4_FP8's FP8 input quantizers are calibrated and `input_scale` is written
    ("W4A8_MXFP4_FP8_CFG", torch.bfloat16, False, "llama", None), ("MXFP4_MLP_WEIGHT_ONLY_CFG", torch.bfloat16, False, "llama", None),
    ("MXFP4_MLP_WEIGHT_ONLY_CFG", torch.bfloat16, False, "mixtral", None), ("W4A8_MXFP4_FP8_CFG", torch.float16, True, "mixtral", None),
    ("MXFP8_DEFAULT_CFG", torch.bfloat16, False, "mixtral", None), ("INT8_SMOOTHQUANT_CFG", torch.bfloat16, False, "mixtral", None),
    ("FP8_PER_CHANNEL_PER_TOKEN_CFG", torch.bfloat16, False, "mixtral", None), ("INT8_WEIGHT_ONLY_CFG", torch.bfloat16, False, "qwen3_moe", None),
    ("MXFP4_DEFAULT_CFG", torch.bfloat16, False, "qwen3_moe", None),
    # 2-D FP8 blocks on fused experts: the reference's name reversal refuses the 4-D expert scales and the whole checkpoint
    # keeps the module tree's names (mirrored: export.rename_to_checkpoint_keys)
    ("FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG", torch.bfloat16, False, "mixtral", None),
    ("FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG", torch.bfloat16, False, "qwen3_moe", None),
    ("W4A8_AWQ_BETA_CFG", torch.bfloat16, True, "gemma2", None), ("INT8_SMOOTHQUANT_CFG", torch.bfloat16, False, "qwen3_moe", None),
    ("W4A8_MXFP4_FP8_CFG", torch.bfloat16, True, "phi3", None), ("INT4_BLOCKWISE_WEIGHT_ONLY_CFG", torch.float16, False, "opt", None),
    ("MXFP8_DEFAULT_CFG", torch.bfloat16, True, "gpt2", None), ("INT8_DEFAULT_CFG", torch.bfloat16, True, "phi3", None),
    # affine KV cache (FP8_AFFINE_KV_CFG): offsets calibrated before the abs-max, exported as k_proj.k_bias / v_proj.v_bias
    ("FP8_DEFAULT_CFG", torch.bfloat16, "affine", "llama", None), ("FP8_DEFAULT_CFG", torch.float32, "affine", "llama-eager", None),
    ("FP8_DEFAULT_CFG", torch.float16, "affine", "qwen2", None), ("INT4_AWQ_CFG", torch.bfloat16, "affine", "mistral", None),
    # cast-style KV cache (use_constant_amax: the reference example's default KV format)
    ("FP8_DEFAULT_CFG", torch.bfloat16, "cast", "llama", None), ("INT4_AWQ_CFG", torch.bfloat16, "cast", "qwen2", None),
    ("FP8_DEFAULT_CFG", torch.float32, "cast", "mixtral", None),
    # MX inputs under a KV-cache config: the max calibration leaves `_amax` buffers on the E8M0 input quantizers, whose `amax`
    # still reads None (tensor_quantizer.py:358-363), so no `input_scale` is written (found by tools/hf_flow_fuzz.py)
    ("MXFP4_DEFAULT_CFG", torch.float16, "cast", "llama", None), ("MXFP4_DEFAULT_CFG", torch.bfloat16, True, "mixtral", None),
    # the MSE weight search reaches the per-expert quantizers of fused expert containers too (iter_weights_for_calibration,
    # quant_module.py:123-129 / huggingface.py:1084-1100; found by tools/hf_flow_fuzz.py)
    ("FP8_DEFAULT_CFG", torch.bfloat16, False, "qwen3_moe", {"method": "mse"}), ("INT8_DEFAULT_CFG", torch.float32, False, "mixtral", {"method": "mse"}),
    # more decoder families (_MORE_ARCHITECTURES)
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "qwen3", None), ("FP8_DEFAULT_CFG", torch.bfloat16, True, "gemma", None),
    ("FP8_DEFAULT_CFG", torch.float16, True, "starcoder2", None), ("FP8_DEFAULT_CFG", torch.bfloat16, True, "gpt_neox", None),
    ("INT4_AWQ_CFG", torch.bfloat16, False, "gpt_neox", None), ("FP8_DEFAULT_CFG", torch.bfloat16, "cast", "olmo2", None),
    ("INT4_AWQ_CFG", torch.bfloat16, False, "cohere", None), ("FP8_DEFAULT_CFG", torch.bfloat16, True, "phi", None),
    ("INT8_SMOOTHQUANT_CFG", torch.bfloat16, False, "granite", None), ("FP8_DEFAULT_CFG", torch.bfloat16, "affine", "glm", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "deepseek_v3", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "gptj", None), ("FP8_DEFAULT_CFG", torch.float16, "cast", "codegen", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, "affine", "mpt", None), ("INT4_AWQ_CFG", torch.bfloat16, True, "gptj", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "stablelm", None), ("FP8_DEFAULT_CFG", torch.bfloat16, True, "nemotron", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "glm4", None), ("FP8_DEFAULT_CFG", torch.bfloat16, True, "exaone4", None),
    ("INT4_AWQ_CFG", torch.bfloat16, False, "ernie4_5", None), ("FP8_DEFAULT_CFG", torch.bfloat16, True, "gpt_bigcode", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "qwen2_moe", None), ("FP8_DEFAULT_CFG", torch.bfloat16, True, "olmoe", None),
    ("FP8_DEFAULT_CFG", torch.float16, "cast", "granitemoe", None), ("INT8_DEFAULT_CFG", torch.bfloat16, False, "phimoe", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "helium", None), ("INT4_AWQ_CFG", torch.bfloat16, False, "arcee", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "apertus", None), ("FP8_DEFAULT_CFG", torch.float16, "cast", "seed_oss", None),
    ("INT8_SMOOTHQUANT_CFG", torch.bfloat16, False, "hunyuan", None), ("FP8_DEFAULT_CFG", torch.bfloat16, True, "gemma3", None),
    ("INT4_AWQ_CFG", torch.bfloat16, True, "bitnet", None), ("FP8_DEFAULT_CFG", torch.bfloat16, True, "glm4_moe", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "ernie4_5_moe", None), ("FP8_DEFAULT_CFG", torch.bfloat16, "affine", "dots1", None),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "minimax", None),
])
def test_quantize_and_export_equal_the_reference_live(monkeypatch, preset, dtype, with_kv, arch, algorithm):
    ref_amax, ref_state = _reference_run(preset, dtype, with_kv, arch, algorithm)
    hostmem_backend.install(monkeypatch, moa)
    our_amax, our_state = _our_run(preset, dtype, with_kv, arch, algorithm)
    # every enabled quantizer the reference calibrated exists here under the same name with the same amax
    for n, a in ref_amax.items():
        assert n in our_amax, f"{preset}: quantizer {n} has no amax here"
        assert torch.equal(our_amax[n].reshape(-1), a.reshape(-1)), f"{preset}: amax of {n} differs"
    ref_json, our_json = ref_state.pop("__quant_json__", None), our_state.pop("__quant_json__", None)
    ref_logits, our_logits = ref_state.pop("__logits__"), our_state.pop("__logits__")
    if ref_logits is not None:  # the forward with fake quantization active, after calibration
        assert torch.equal(our_logits, ref_logits), f"{preset}: logits of the fake-quantized model differ"
    if arch == "falcon":
        return  # the reference's exporter leaves FalconLinear weights unpacked; calibration and fake quant are compared
    if arch == "llama-ragged":
        return  # calibration (alpha search on padded blocks), amax and the fake-quantized forward are compared
    assert sorted(our_state) == sorted(ref_state), set(our_state) ^ set(ref_state)
    if ref_json is not None and ref_json[0] is not None:
        _assert_same_quant_json(our_json, ref_json, f"{preset} {arch}")
    for k, want in ref_state.items():
        got = our_state[k].detach().cpu()
        assert got.dtype == want.dtype and tuple(got.shape) == tuple(want.shape), f"{preset} {k}: {got.dtype} {tuple(got.shape)} vs {want.dtype} {tuple(want.shape)}"
        assert torch.equal(got.contiguous().reshape(-1).view(torch.uint8), want.contiguous().reshape(-1).view(torch.uint8)), f"{preset}: {k} differs"


def _override(extra):
    """Per-layer entries appended after a preset's own (the last matching entry wins in both libraries)."""
    def edit(cfg):
        qc = cfg["quant_cfg"]
        for pattern, value in extra:
            if isinstance(qc, dict):
                qc.pop(pattern, None)  # (re-inserted at the END: a later entry wins, like the appended entry of the list form)
                qc[pattern] = dict(value)
            elif value == {"enable": False}:
                qc.append({"quantizer_name": pattern, "enable": False})
            else:
                qc.append({"quantizer_name": pattern, "cfg": dict(value)})
    return edit


_FP8_ATTENTION = [("*self_attn*weight_quantizer", {"num_bits": (4, 3), "axis": None}), ("*self_attn*input_quantizer", {"num_bits": (4, 3), "axis": None})]
_INT8_MLP = [("*mlp*weight_quantizer", {"num_bits": 8, "axis": 0}), ("*mlp*input_quantizer", {"num_bits": 8, "axis": None})]
_FP8_2D_ATTENTION = [("*self_attn*weight_quantizer", {"num_bits": (4, 3), "block_sizes": {-1: 64, -2: 64}}), ("*self_attn*input_quantizer", {"enable": False})]
_NO_FIRST_LAYER = [("*layers.0.*", {"enable": False})]
_INT8_PER_TENSOR_WEIGHTS = [("*layers.*weight_quantizer", {"num_bits": 8, "axis": None})]
_QUANTIZED_HEAD = [("*lm_head*weight_quantizer", {"num_bits": 8, "axis": 0})]
_DYNAMIC_INPUTS = [("*input_quantizer", {"num_bits": 8, "axis": None, "type": "dynamic"})]  # (re-enables the head's input quantizer too)
_FP8_OUTPUT = [("*o_proj*output_quantizer", {"num_bits": (4, 3), "axis": None})]
_MIXED_OUTPUTS = _FP8_OUTPUT + [("*down_proj*output_quantizer", {"num_bits": 8, "axis": None})]


@pytest.mark.parametrize("preset,dtype,with_kv,arch,extra", [
    # a second format on part of an AWQ model: awq_lite smooths and searches EVERY linear whose weight quantizer is enabled
    # (model_calib.py:1563-1567), per-tensor FP8 / per-channel INT8 / 2-D FP8 blocks included -- the scaled weight goes through
    # the quantizer itself with the amax of its own layout, weight scale over whole rows (found by tools/hf_flow_fuzz.py:
    # this package searched the INT-k block linears only and max-calibrated the rest)
    ("W4A8_AWQ_BETA_CFG", torch.float16, True, "llama-eager", _FP8_ATTENTION), ("INT4_AWQ_CFG", torch.bfloat16, False, "llama", _INT8_MLP),
    ("INT4_AWQ_CFG", torch.bfloat16, False, "qwen2", _FP8_2D_ATTENTION), ("INT4_AWQ_CFG", torch.float16, "cast", "mistral", _FP8_ATTENTION),
    # layers left out (exclude_modules of the checkpoint's tables), a second format under max calibration (the per-layer table)
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "llama", _NO_FIRST_LAYER), ("FP8_DEFAULT_CFG", torch.bfloat16, False, "qwen2", _INT8_MLP),
    ("INT8_SMOOTHQUANT_CFG", torch.bfloat16, False, "llama", _FP8_ATTENTION),
    # per-tensor INT8 weights (`wsf[:, None]` of a one-element factor broadcasts over the rows, export/quant_utils.py:868-869), a
    # quantized head, dynamic per-tensor inputs, and an enabled OUTPUT quantizer -- which the exporter's KV-cache rule counts
    # (get_kv_cache_dtype reads any module's output quantizer, :426-435: `kv_cache_quant_algo` "FP8" without a KV quantizer)
    ("INT8_DEFAULT_CFG", torch.bfloat16, False, "llama", _INT8_PER_TENSOR_WEIGHTS), ("FP8_DEFAULT_CFG", torch.float16, True, "qwen2", _QUANTIZED_HEAD),
    ("FP8_DEFAULT_CFG", torch.float16, True, "mistral", _DYNAMIC_INPUTS), ("FP8_DEFAULT_CFG", torch.bfloat16, False, "mistral", _FP8_OUTPUT),
])
def test_per_layer_overrides_of_a_preset_equal_the_reference_live(monkeypatch, preset, dtype, with_kv, arch, extra):
    ref_amax, ref_state = _reference_run(preset, dtype, with_kv, arch, None, edit=_override(extra))
    hostmem_backend.install(monkeypatch, moa)
    our_amax, our_state = _our_run(preset, dtype, with_kv, arch, None, edit=_override(extra))
    for n, a in ref_amax.items():
        assert n in our_amax, f"{preset}: quantizer {n} has no amax here"
        assert torch.equal(our_amax[n].reshape(-1), a.reshape(-1)), f"{preset}: amax of {n} differs"
    ref_json, our_json = ref_state.pop("__quant_json__", None), our_state.pop("__quant_json__", None)
    ref_logits, our_logits = ref_state.pop("__logits__"), our_state.pop("__logits__")
    assert torch.equal(our_logits, ref_logits), f"{preset}: logits of the fake-quantized model differ"
    assert sorted(our_state) == sorted(ref_state), set(our_state) ^ set(ref_state)
    if ref_json is not None and ref_json[0] is not None:
        _assert_same_quant_json(our_json, ref_json, f"{preset} {arch}")
    for k, want in ref_state.items():
        got = our_state[k].detach().cpu()
        assert got.dtype == want.dtype and tuple(got.shape) == tuple(want.shape), f"{preset} {k}: {got.dtype} {tuple(got.shape)} vs {want.dtype} {tuple(want.shape)}"
        assert torch.equal(got.contiguous().reshape(-1).view(torch.uint8), want.contiguous().reshape(-1).view(torch.uint8)), f"{preset}: {k} differs"


@pytest.mark.parametrize("preset,dtype,arch,shard", [
    ("FP8_DEFAULT_CFG", torch.bfloat16, "llama", None), ("INT4_AWQ_CFG", torch.float16, "qwen2", None), ("FP8_DEFAULT_CFG", torch.bfloat16, "mixtral", None),
    # max_shard_size splits the checkpoint: WHICH tensor lands in which file follows the order of the exported dict (the
    # model's state_dict order with the exporter's buffers in place, expert containers expanded where they stood), and the index
    # counts the parameters of a model that holds the packed tensors
    ("FP8_DEFAULT_CFG", torch.bfloat16, "mixtral", "100KB"), ("INT4_AWQ_CFG", torch.float16, "qwen2", "100KB"),
    ("W4A8_AWQ_BETA_CFG", torch.bfloat16, "gpt2", "100KB")])
def test_the_checkpoint_files_on_disk_are_the_references_live(monkeypatch, preset, dtype, arch, shard):
    """export_hf_checkpoint's directory against this package's export_hf_checkpoint's: the same four files; `model.safetensors`
    (header, key order, metadata, every tensor) and `generation_config.json` byte for byte; `hf_quant_config.json` and
    `config.json` (its `quantization_config` included) the same documents up to each library's own `producer` entry."""
    import json

    ref_shim.install()
    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint

    batches = _batches()
    cfg = mtq.update_quant_cfg_with_kv_cache_quant(copy.deepcopy(getattr(mtq, preset)), copy.deepcopy(mtq.FP8_KV_CFG["quant_cfg"]))
    ref = mtq.quantize(_model(dtype, arch), cfg, lambda m: [m(b) for b in batches])
    with tempfile.TemporaryDirectory() as there, tempfile.TemporaryDirectory() as here:
        export_hf_checkpoint(ref, export_dir=there, **({"max_shard_size": shard} if shard else {}))
        hostmem_backend.install(monkeypatch, moa)
        mq = moa.model_quant
        ours = _model(dtype, arch)
        cfg = mq.update_quant_cfg_with_kv_cache_quant(copy.deepcopy(getattr(mq, preset)), mq.FP8_KV_CFG["quant_cfg"])
        with torch.no_grad():
            moa.quantize(ours, cfg, lambda m: [m(b) for b in batches])
        moa.export.export_hf_checkpoint(ours, dtype, here, **({"max_shard_size": shard} if shard else {}))
        assert sorted(os.listdir(here)) == sorted(os.listdir(there))
        weights = [f for f in os.listdir(there) if f.endswith(".safetensors") or f.endswith(".index.json")]
        assert (len(weights) > 2) == bool(shard) and ("model.safetensors" in weights) == (not shard)
        for name in weights + ["generation_config.json"]:
            assert open(os.path.join(here, name), "rb").read() == open(os.path.join(there, name), "rb").read(), name
        mine, theirs = (json.load(open(os.path.join(d, "hf_quant_config.json"))) for d in (here, there))
        assert mine.pop("producer")["name"] != theirs.pop("producer")["name"] and mine == theirs
        mine, theirs = (json.load(open(os.path.join(d, "config.json"))) for d in (here, there))
        assert mine["quantization_config"].pop("producer") != theirs["quantization_config"].pop("producer") and mine == theirs


@pytest.mark.parametrize("preset", ["INT8_WEIGHT_ONLY_CFG", "INT4_BLOCKWISE_WEIGHT_ONLY_CFG", "FP8_DEFAULT_CFG", "INT8_SMOOTHQUANT_CFG",
                                    "INT4_AWQ_CFG", "FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG"])
def test_quantize_without_a_forward_loop_behaves_like_the_reference_live(monkeypatch, preset):
    """forward_loop=None: weight-only calibration for the max presets (inputs stay uncalibrated), awq_lite warns and skips
    (model_calib.py:1412-1414), smoothquant stops with its assertion (:1298)."""
    import warnings

    ref_shim.install()
    import modelopt.torch.quantization as mtq

    def both(run):
        try:
            with warnings.catch_warnings(record=True) as seen:
                warnings.simplefilter("always")
                amax, logits = run()
            return amax, logits, sorted({str(w.message) for w in seen if "forward_loop" in str(w.message)})
        except AssertionError as e:
            return str(e)

    def reference():
        q = mtq.quantize(_model(torch.bfloat16), copy.deepcopy(getattr(mtq, preset)), None)
        amax = {n: m._amax.float().clone() for n, m in q.named_modules() if type(m).__name__.endswith("Quantizer") and getattr(m, "_amax", None) is not None}
        with torch.no_grad():
            return amax, q(_batches()[0]).logits

    def ours():
        model = _model(torch.bfloat16)
        with torch.no_grad():
            moa.quantize(model, copy.deepcopy(getattr(moa.model_quant, preset)), None)
            amax = {n: m._amax.float().clone() for n, m in model.named_modules() if isinstance(m, moa.TensorQuantizer) and getattr(m, "_amax", None) is not None}
            return amax, model(_batches()[0]).logits

    want = both(reference)
    hostmem_backend.install(monkeypatch, moa)
    got = both(ours)
    if isinstance(want, str):
        assert got == want and "forward_loop must be provided" in want
        return
    assert sorted(got[0]) == sorted(want[0])
    for n, a in want[0].items():
        assert torch.equal(got[0][n].reshape(-1), a.reshape(-1)), n
    assert torch.equal(got[1], want[1]) and got[2] == want[2]


def test_the_functions_used_on_a_quantized_model_equal_the_reference_live(monkeypatch):
    """disable_quantizer / enable_quantizer by wildcard, calibrate() again on other data, postprocess_amax -- and the
    state_dict of the quantized model (keys, shapes, dtypes, values) -- step by step beside the reference."""
    ref_shim.install()
    import modelopt.torch.quantization as mtq

    batches = _batches()
    more = [torch.randint(0, CFG["vocab_size"], (2, 30), generator=torch.Generator().manual_seed(99 + i)) for i in range(2)]

    def walk(lib, model, ours):
        seen = []
        with torch.no_grad():
            cfg = lib.update_quant_cfg_with_kv_cache_quant(copy.deepcopy(lib.FP8_DEFAULT_CFG), copy.deepcopy(lib.FP8_KV_CFG["quant_cfg"])) \
                if not ours else moa.model_quant.update_quant_cfg_with_kv_cache_quant(copy.deepcopy(lib.FP8_DEFAULT_CFG), lib.FP8_KV_CFG["quant_cfg"])
            q = lib.quantize(model, cfg, lambda m: [m(b) for b in batches]) or model
            seen.append({k: v.detach().clone() for k, v in q.state_dict().items()})
            lib.disable_quantizer(q, "*mlp*")
            seen.append(q(batches[0]).logits.clone())
            lib.enable_quantizer(q, "*mlp*input_quantizer")
            seen.append(q(batches[0]).logits.clone())
            lib.calibrate(q, algorithm="max", forward_loop=lambda m: [m(b) for b in more])
            seen.append({k: v.detach().clone() for k, v in q.state_dict().items() if k.endswith("_amax")})
            lib.postprocess_amax(q, "*input_quantizer", lambda a: torch.clamp(a, min=0.5))
            seen.append({k: v.detach().clone() for k, v in q.state_dict().items() if k.endswith("_amax")})
            seen.append(q(batches[0]).logits.clone())
        return seen

    want = walk(mtq, _model(torch.bfloat16), False)
    hostmem_backend.install(monkeypatch, moa)
    got = walk(moa.model_quant, _model(torch.bfloat16), True)
    for i, (a, b) in enumerate(zip(want, got)):
        if isinstance(a, dict):
            assert sorted(a) == sorted(b), (i, set(a) ^ set(b))
            for k in a:
                assert a[k].dtype == b[k].dtype and a[k].shape == b[k].shape and torch.equal(a[k], b[k]), (i, k)
        else:
            assert torch.equal(a, b), i


def test_the_attribute_setters_equal_the_reference_live(monkeypatch):
    """set_quantizer_attributes_partial (a merge: the calibrated amax stays, also under other num_bits), set_quantizer_by_cfg_context
    (attributes put back on exit), set_quantizer_attributes_full (one config, or a list -> a quantizer chain) and their refusals,
    step by step beside the reference (conversion.py:373-512, :568-599)."""
    ref_shim.install()
    import modelopt.torch.quantization as mtq
    from modelopt.torch.quantization.config import QuantizerAttributeConfig as RefConfig

    batches = _batches()

    def states(q, ours):
        return {n: (m.is_enabled, str(m.num_bits), str(m.axis), str(m.block_sizes)) for n, m in q.named_modules()
                if (isinstance(m, moa.TensorQuantizer) if ours else type(m).__name__ == "TensorQuantizer") and "embed" not in n}

    def walk(lib, config, ours):
        seen = []

        def refused(fn):
            try:
                fn()
            except (AssertionError, ValueError) as e:
                return f"{type(e).__name__}: {str(e)[:40]}"

        with torch.no_grad():
            model = _model(torch.bfloat16)
            q = lib.quantize(model, copy.deepcopy(lib.FP8_DEFAULT_CFG), lambda m: [m(b) for b in batches]) or model
            lib.set_quantizer_attributes_partial(q, "*mlp*input_quantizer", {"enable": False})
            seen += [states(q, ours), q(batches[0]).logits.clone()]
            lib.set_quantizer_attributes_partial(q, lambda n: n.endswith("o_proj.weight_quantizer"), {"num_bits": 8})
            seen += [states(q, ours), q(batches[0]).logits.clone()]
            entry = {"*self_attn*input_quantizer": {"enable": False}} if ours else [{"quantizer_name": "*self_attn*input_quantizer", "enable": False}]
            with lib.set_quantizer_by_cfg_context(q, entry):
                seen += [states(q, ours), q(batches[0]).logits.clone()]
            seen += [states(q, ours), q(batches[0]).logits.clone()]
            seen.append(refused(lambda: lib.set_quantizer_attributes_partial(q, "*gate_proj.weight_quantizer", [{"num_bits": 8}])))
            seen.append(refused(lambda: lib.set_quantizer_attributes_partial(q, "*gate_proj.weight_quantizer", {"no_such": 1})))
            lib.set_quantizer_attributes_full(q, "*down_proj.weight_quantizer", config(num_bits=8, axis=0))
            lib.set_quantizer_attributes_full(q, "*up_proj.weight_quantizer", [config(num_bits=4, block_sizes={-1: 128}), config(num_bits=(4, 3), axis=None)])
            seen += [states(q, ours), {n: type(m).__name__ for n, m in q.named_modules() if n.endswith("up_proj.weight_quantizer")}]
        return seen

    want = walk(mtq, RefConfig, False)
    hostmem_backend.install(monkeypatch, moa)
    got = walk(moa.model_quant, moa.QuantizerAttributeConfig, True)
    assert len(want) == len(got)
    for i, (a, b) in enumerate(zip(want, got)):
        assert torch.equal(a, b) if isinstance(a, torch.Tensor) else a == b, (i, a if not isinstance(a, (dict, torch.Tensor)) else "", b if not isinstance(b, (dict, torch.Tensor)) else "")


def test_need_calibration_answers_like_the_reference_for_every_preset_live():
    ref_shim.install()
    import modelopt.torch.quantization as mtq

    names = [n for n in dir(moa.model_quant) if n.endswith("_CFG") and hasattr(mtq, n) and "algorithm" in getattr(mtq, n)]
    assert len(names) >= 15
    for n in names:
        assert moa.model_quant.need_calibration(getattr(moa.model_quant, n)) == mtq.need_calibration(getattr(mtq, n)), n
        assert getattr(moa.model_quant, n)["algorithm"] == getattr(mtq, n)["algorithm"], n  # the presets' algorithm literals


@pytest.mark.parametrize("arch", ["llama", "mixtral"])
def test_every_shared_model_preset_configures_the_quantizers_like_the_reference_live(monkeypatch, arch):
    """Every model preset both libraries name, applied without calibration: the same quantizers exist under the same names with
    the same switches, bits, axes, block sizes, dynamic flag, offsets and constant-amax flag.  (The KV presets are compared
    merged into model presets everywhere else; on their own they leave the reference's never-used q / p bmm quantizers at the
    class default -- enabled -- where this package creates them switched off.)"""
    ref_shim.install()
    import modelopt.torch.quantization as mtq

    def state(q, ours):
        return {n: (m.is_enabled, str(m.num_bits), str(m.axis), str(m.block_sizes), bool(m._dynamic), str(getattr(m, "_bias", None)),
                    bool(getattr(m, "_use_constant_amax", False)))
                for n, m in q.named_modules() if "embed" not in n
                and (isinstance(m, moa.TensorQuantizer) if ours else type(m).__name__ == "TensorQuantizer")}

    names = [n for n in dir(moa.model_quant) if n.endswith("_CFG") and hasattr(mtq, n) and "algorithm" in getattr(mtq, n) and "_KV_" not in n]
    want = {n: state(mtq.quantize(_model(torch.bfloat16, arch), {**copy.deepcopy(getattr(mtq, n)), "algorithm": None}, None), False) for n in names}
    hostmem_backend.install(monkeypatch, moa)
    for n in names:
        model = _model(torch.bfloat16, arch)
        moa.quantize(model, {**copy.deepcopy(getattr(moa.model_quant, n)), "algorithm": None}, None)
        assert state(model, True) == want[n], n


@pytest.mark.parametrize("preset,with_kv", [("INT4_AWQ_CFG", "cast"), ("FP8_DEFAULT_CFG", "affine"), ("W4A8_AWQ_BETA_CFG", False),
                                            ("INT8_SMOOTHQUANT_CFG", True), ("MXFP4_DEFAULT_CFG", True)])
def test_print_quant_summary_prints_the_references_lines_live(monkeypatch, preset, with_kv):
    """What a user reads after quantize(): one line per quantizer in the reference's order (input, output, weight per linear),
    each quantizer in the reference's words -- `disabled`, bits, `per-tensor` / axis / block sizes, amax as value or
    `[min, max](count)`, `dynamic`, the constant, the smoothing scale, calibrator, offset, switches; a quantizer promoted to
    static block scales under its subclass name.  Up to the reference's (switched-off) embedding quantizers and the count."""
    import contextlib
    import io

    ref_shim.install()
    import modelopt.torch.quantization as mtq

    batches = _batches()

    def lines(lib, model):
        out = io.StringIO()
        with contextlib.redirect_stdout(out):
            lib.print_quant_summary(model)
        return [ln for ln in out.getvalue().splitlines() if "embed_tokens" not in ln]

    def config(lib, ours):
        cfg = copy.deepcopy(getattr(lib, preset))
        if not with_kv:
            return cfg
        if ours:
            kv = {"affine": lib.FP8_AFFINE_KV_CFG, "cast": lib.FP8_CAST_KV_CFG}.get(with_kv, lib.FP8_KV_CFG)["quant_cfg"]
        elif with_kv == "cast":
            kv = [{"quantizer_name": "*[kv]_bmm_quantizer", "cfg": {"num_bits": (4, 3), "axis": None, "use_constant_amax": True}}]
        else:
            kv = copy.deepcopy((lib.FP8_AFFINE_KV_CFG if with_kv == "affine" else lib.FP8_KV_CFG)["quant_cfg"])
        return lib.update_quant_cfg_with_kv_cache_quant(cfg, kv)

    loop = (lambda m: [m(b) for b in batches]) if getattr(mtq, preset).get("algorithm") or with_kv else None
    ref = mtq.quantize(_model(torch.bfloat16), config(mtq, False), loop)
    want = lines(mtq, ref)
    hostmem_backend.install(monkeypatch, moa)
    ours = _model(torch.bfloat16)
    with torch.no_grad():
        moa.quantize(ours, config(moa.model_quant, True), loop)
    got = lines(moa.model_quant, ours)
    assert len(got) == len(want) > 20
    assert got[:-1] == want[:-1]
    assert got[-1].endswith("TensorQuantizers found in model") and want[-1].endswith("TensorQuantizers found in model")


@pytest.mark.parametrize("case", ["all_off", "kv_only", "other_dtype_fp8", "other_dtype_int4_awq"])
def test_corner_exports_write_the_references_directory_live(monkeypatch, case):
    """A model whose quantizers are all off (three plain files, no quantization tables), a model with only the KV cache
    quantized, and exports in a dtype other than the model's: the directory is the reference's (producer entries aside)."""
    import json

    ref_shim.install()
    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint

    batches = _batches()
    model_dtype, export_dtype = {"other_dtype_fp8": (torch.bfloat16, torch.float16), "other_dtype_int4_awq": (torch.float16, torch.bfloat16)}.get(
        case, (torch.bfloat16, torch.bfloat16))

    def config(lib, ours):
        off = {"*": {"enable": False}} if ours else [{"quantizer_name": "*", "enable": False}]
        if case == "all_off":
            return {"quant_cfg": off, "algorithm": "max"}
        if case == "kv_only":
            kv = lib.FP8_KV_CFG["quant_cfg"]
            return {"quant_cfg": {**off, **kv} if ours else off + copy.deepcopy(kv), "algorithm": "max"}
        return copy.deepcopy(lib.FP8_DEFAULT_CFG if case == "other_dtype_fp8" else lib.INT4_AWQ_CFG)

    with tempfile.TemporaryDirectory() as there, tempfile.TemporaryDirectory() as here:
        ref = mtq.quantize(_model(model_dtype), config(mtq, False), lambda m: [m(b) for b in batches])
        export_hf_checkpoint(ref, dtype=export_dtype, export_dir=there)
        hostmem_backend.install(monkeypatch, moa)
        ours = _model(model_dtype)
        with torch.no_grad():
            moa.quantize(ours, config(moa.model_quant, True), lambda m: [m(b) for b in batches])
        moa.export.export_hf_checkpoint(ours, export_dtype, here)
        assert sorted(os.listdir(here)) == sorted(os.listdir(there))
        assert ("hf_quant_config.json" in os.listdir(there)) == (case != "all_off")
        for name in os.listdir(there):
            mine, theirs = open(os.path.join(here, name), "rb").read(), open(os.path.join(there, name), "rb").read()
            if name.endswith(".json") and mine != theirs:
                mine, theirs = json.loads(mine), json.loads(theirs)
                for doc in (mine, theirs):
                    doc.pop("producer", None)
                    (doc.get("quantization_config") or {}).pop("producer", None)
            assert mine == theirs, name


def test_expert_containers_the_reference_has_a_class_of_its_own_for_are_refused(monkeypatch):
    """GPT-OSS' experts ([E, H, 2I] with biases) have `_QuantGptOssExperts` in the reference (plugins/huggingface.py:1467-1557);
    the generic per-expert rule would take them and quantize them differently, so quantize() stops by name."""
    import transformers as tf

    hostmem_backend.install(monkeypatch, moa)
    model = tf.GptOssForCausalLM(tf.GptOssConfig(num_local_experts=4, num_experts_per_tok=2, head_dim=32, **CFG)).to(torch.bfloat16).eval()
    with pytest.raises(moa.MoquantUnsupported, match="GptOssExperts"):
        moa.quantize(model, copy.deepcopy(moa.model_quant.FP8_DEFAULT_CFG), lambda m: [m(b) for b in _batches()])


@pytest.mark.parametrize("with_kv", [False, True])
def test_output_quantizers_of_two_formats_stop_the_export_like_the_reference_live(monkeypatch, with_kv):
    """unified_export_hf's table (quant_utils.py:1675-1690): every module with an enabled k / v bmm OR output quantizer must
    name the same KV-cache format."""
    with pytest.raises(AssertionError, match="mixed precision kv cache"):
        _reference_run("FP8_DEFAULT_CFG", torch.bfloat16, with_kv, "mistral", None, edit=_override(_MIXED_OUTPUTS))
    hostmem_backend.install(monkeypatch, moa)
    with pytest.raises(AssertionError, match="mixed precision kv cache"):
        _our_run("FP8_DEFAULT_CFG", torch.bfloat16, with_kv, "mistral", None, edit=_override(_MIXED_OUTPUTS))


@pytest.mark.parametrize("preset,dtype", [("FP8_DEFAULT_CFG", torch.bfloat16), ("INT8_DEFAULT_CFG", torch.float32),
                                          ("INT4_BLOCKWISE_WEIGHT_ONLY_CFG", torch.bfloat16),
                                          ("INT8_SMOOTHQUANT_CFG", torch.float16)])
def test_fold_weight_equals_the_reference_live(monkeypatch, preset, dtype):
    """mtq.fold_weight (model_quant.py:728-736, quant_module.py:132-186) against model_quant.fold_weight on the same
    calibrated model: folded weights bit-identical, weight quantizers disabled and stripped of amax / pre_quant_scale,
    logits of the folded model equal."""
    ref_shim.install()
    import modelopt.torch.quantization as mtq

    batches = _batches()
    ref = mtq.quantize(_model(dtype), copy.deepcopy(getattr(mtq, preset)), lambda m: [m(b) for b in batches])
    mtq.fold_weight(ref)
    with torch.no_grad():
        ref_logits = ref(batches[0]).logits
    ref_w = {n: p.detach().clone() for n, p in ref.named_parameters()}
    hostmem_backend.install(monkeypatch, moa)
    ours = _model(dtype)
    with torch.no_grad():
        moa.quantize(ours, copy.deepcopy(getattr(moa.model_quant, preset)), lambda m: [m(b) for b in batches])
        moa.model_quant.fold_weight(ours)
        logits = ours(batches[0]).logits
    for n, p in ours.named_parameters():
        assert torch.equal(p.detach(), ref_w[n]), f"{preset}: folded {n} differs from the reference's"
    for n, m in ours.named_modules():
        if n.endswith("weight_quantizer"):
            assert not m.is_enabled and not hasattr(m, "_amax") and not hasattr(m, "_pre_quant_scale"), n
    assert torch.equal(logits, ref_logits)


def test_magnitude_sparsity_equals_the_reference_live(monkeypatch):
    """mts.sparsify(model, "sparse_magnitude") of the reference against sparsity.sparsify on the same model: masks equal."""
    ref_shim.install()
    import modelopt.torch.sparsity as mts

    ref = mts.sparsify(_model(torch.bfloat16), "sparse_magnitude")
    ref_masks = {n[: -len("._weight_mask")]: b.clone() for n, b in ref.named_buffers() if n.endswith("_weight_mask")}
    hostmem_backend.install(monkeypatch, moa)
    ours = moa.sparsity.sparsify(_model(torch.bfloat16), "sparse_magnitude")
    our_masks = {n[: -len("._weight_mask")]: b for n, b in ours.named_buffers() if n.endswith("_weight_mask")}
    assert set(our_masks) == set(ref_masks) and len(ref_masks) >= 14
    for n, m in ref_masks.items():
        assert torch.equal(our_masks[n].bool(), m.bool()), n


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_awq_lite_picks_the_reference_alphas_live(monkeypatch, dtype):
    """INT4_AWQ_CFG on the tiny Llama: the reference (debug=True keeps its per-linear state) and this package pick the
    same alpha for every linear; exported tensors agree within 2 ulp of the model dtype (the activation statistics are
    reduced in different orders, which can move a scale by one ulp)."""
    ref_shim.install()
    import modelopt.torch.quantization as mtq

    batches = _batches()
    cfg = copy.deepcopy(mtq.INT4_AWQ_CFG)
    cfg["algorithm"]["debug"] = True
    ref = mtq.quantize(_model(dtype), cfg, lambda m: [m(b) for b in batches])
    ref_alpha = {n: float(m.awq_lite.best_alpha) for n, m in ref.named_modules() if hasattr(m, "awq_lite")}
    ref_w = {n: m.weight.detach().float().clone() for n, m in ref.named_modules() if hasattr(m, "awq_lite")}
    hostmem_backend.install(monkeypatch, moa)
    ours = _model(dtype)
    with torch.no_grad():
        moa.quantize(ours, moa.model_quant.INT4_AWQ_CFG, lambda m: [m(b) for b in batches])
    our_alpha = {n: float(m.awq_lite.best_alpha) for n, m in ours.named_modules() if hasattr(m, "awq_lite")}
    assert set(our_alpha) == set(ref_alpha) and len(ref_alpha) == 14
    assert our_alpha == ref_alpha, {n: (our_alpha[n], ref_alpha[n]) for n in ref_alpha if our_alpha[n] != ref_alpha[n]}
    ulp = torch.finfo(dtype).eps
    for n, w in ref_w.items():
        got = ours.get_submodule(n).weight.detach().float()
        assert ((got - w).abs() <= max(2 * ulp, 1e-5) * w.abs() + 1e-30).all(), f"{n}: folded weight off by more than 2 ulp / 1e-5"


@pytest.mark.parametrize("preset,dtype", [("INT8_DEFAULT_CFG", torch.float32), ("FP8_DEFAULT_CFG", torch.bfloat16),
                                          ("INT4_BLOCKWISE_WEIGHT_ONLY_CFG", torch.bfloat16)])
def test_mse_calibration_equals_the_reference_live(monkeypatch, preset, dtype):
    """algorithm "mse" (max calibration, then the 39-candidate amax sweep per weight quantizer): same refined amax
    everywhere, up to candidates whose losses tie within the summation order (at most 1 % of the entries may sit on a
    neighbouring candidate)."""
    ref_shim.install()
    import modelopt.torch.quantization as mtq

    batches = _batches()
    cfg = copy.deepcopy(getattr(mtq, preset))
    cfg["algorithm"] = "mse"
    ref = mtq.quantize(_model(dtype), cfg, lambda m: [m(b) for b in batches])
    ref_amax = {n: m._amax.detach().float().reshape(-1).clone() for n, m in ref.named_modules()
                if type(m).__name__.endswith("Quantizer") and getattr(m, "_amax", None) is not None and m.is_enabled}
    hostmem_backend.install(monkeypatch, moa)
    ours = _model(dtype)
    cfg2 = copy.deepcopy(getattr(moa.model_quant, preset))
    cfg2["algorithm"] = "mse"
    with torch.no_grad():
        moa.quantize(ours, cfg2, lambda m: [m(b) for b in batches])
    our_amax = {n: m._amax.detach().float().reshape(-1) for n, m in ours.named_modules()
                if isinstance(m, moa.TensorQuantizer) and getattr(m, "_amax", None) is not None and m.is_enabled}
    assert set(our_amax) == set(ref_amax)
    total = differing = 0
    for n, a in ref_amax.items():
        assert our_amax[n].shape == a.shape, n
        total += a.numel()
        differing += int((our_amax[n] != a).sum())
    assert differing <= 0.01 * total, f"{preset}: {differing} of {total} amax entries differ"


def test_awq_clip_equals_the_reference_live(monkeypatch):
    ref_shim.install()
    import modelopt.torch.quantization as mtq

    batches = _batches()
    cfg = copy.deepcopy(mtq.INT4_AWQ_CFG)
    cfg["algorithm"] = {"method": "awq_clip"}
    ref = mtq.quantize(_model(torch.float32), cfg, lambda m: [m(b) for b in batches])
    ref_amax = {n: m.weight_quantizer._amax.detach().float().reshape(-1).clone() for n, m in ref.named_modules()
                if hasattr(m, "weight_quantizer") and m.weight_quantizer.is_enabled}
    hostmem_backend.install(monkeypatch, moa)
    ours = _model(torch.float32)
    cfg2 = copy.deepcopy(moa.model_quant.INT4_AWQ_CFG)
    cfg2["algorithm"] = {"method": "awq_clip"}
    with torch.no_grad():
        moa.quantize(ours, cfg2, lambda m: [m(b) for b in batches])
    total = differing = 0
    for n, a in ref_amax.items():
        got = ours.get_submodule(n).weight_quantizer._amax.detach().float().reshape(-1)
        assert got.shape == a.shape, n
        total += a.numel()
        differing += int(((got - a).abs() > 1e-6 * a.abs()).sum())
    assert total > 0 and differing <= 0.01 * total, f"awq_clip: {differing} of {total} clipped block amax values differ"


@pytest.mark.parametrize("arch,dtype,preset", [("llama", torch.bfloat16, None), ("gemma2", torch.float16, None),
                                               ("mixtral", torch.float32, "FP8_DEFAULT_CFG"), ("opt", torch.bfloat16, "INT4_BLOCKWISE_WEIGHT_ONLY_CFG")])
def test_sparse_models_exported_or_quantized_equal_the_reference_live(monkeypatch, arch, dtype, preset):
    """mts.sparsify + mts.export (masks folded into the weights, no mask buffers left; sparsification.py:100-123) or
    mts.sparsify + mtq.quantize on top of the sparse weights, against this package's: masks, exported state dict / amax, logits
    (tools/sparsity_fuzz.py runs the same over random shapes)."""
    ref_shim.install()
    import modelopt.torch.quantization as mtq
    import modelopt.torch.sparsity as mts

    batches = _batches()
    ref = mts.sparsify(_model(dtype, arch), "sparse_magnitude")
    ref_masks = {n[: -len("._weight_mask")]: b.clone() for n, b in ref.named_buffers() if n.endswith("_weight_mask")}
    if preset:
        ref = mtq.quantize(ref, copy.deepcopy(getattr(mtq, preset)), lambda m: [m(b) for b in batches])
        ref_amax = {n: m._amax.detach().float().clone() for n, m in ref.named_modules()
                    if type(m).__name__ == "TensorQuantizer" and m.is_enabled and getattr(m, "_amax", None) is not None}
    with torch.no_grad():
        ref_logits = ref(batches[0]).logits.clone()
    if not preset:
        ref_state = {k: v.detach().clone() for k, v in mts.export(ref).state_dict().items()}
    hostmem_backend.install(monkeypatch, moa)
    ours = moa.sparsity.sparsify(_model(dtype, arch), "sparse_magnitude")
    our_masks = {n[: -len("._weight_mask")]: b.clone() for n, b in ours.named_buffers() if n.endswith("_weight_mask")}
    assert sorted(our_masks) == sorted(ref_masks) and ref_masks
    for n, m in ref_masks.items():
        assert torch.equal(our_masks[n].bool(), m.bool()), n
    with torch.no_grad():
        if preset:
            moa.quantize(ours, copy.deepcopy(getattr(moa.model_quant, preset)), lambda m: [m(b) for b in batches])
            for n, a in ref_amax.items():
                assert torch.equal(dict(ours.named_modules())[n]._amax.float().reshape(-1), a.reshape(-1)), n
        assert torch.equal(ours(batches[0]).logits, ref_logits)
    if not preset:
        state = moa.sparsity.export(ours).state_dict()
        assert sorted(state) == sorted(ref_state) and not any(k.endswith("_weight_mask") for k in state)
        for k, v in ref_state.items():
            assert torch.equal(state[k], v), k


def test_sparsegpt_equals_the_reference_live(monkeypatch):
    """mts.sparsify(model, "sparsegpt") against sparsity.sparsify on the same model and batches (fp32 model: both sides
    accumulate the Hessian with an fp32 library GEMM): masks agree on at least 99 % of the weights of every linear
    (Cholesky / trailing-update summation order), every mask is 2:4."""
    ref_shim.install()
    import modelopt.torch.sparsity as mts

    batches = _batches()
    ref = mts.sparsify(_model(torch.float32), "sparsegpt",
                       config={"data_loader": batches, "collect_func": lambda b: b})
    ref_masks = {n[: -len("._weight_mask")]: b.clone() for n, b in ref.named_buffers() if n.endswith("_weight_mask")}
    hostmem_backend.install(monkeypatch, moa)
    ours = moa.sparsity.sparsify(_model(torch.float32), "sparsegpt", lambda m: [m(b) for b in batches])
    our_masks = {n[: -len("._weight_mask")]: b for n, b in ours.named_buffers() if n.endswith("_weight_mask")}
    assert set(our_masks) == set(ref_masks) and len(ref_masks) == 14
    for n, m in ref_masks.items():
        got = our_masks[n].bool()
        assert (got.view(got.shape[0], -1, 4).sum(-1) <= 2).all(), n
        same = (got == m.bool()).float().mean().item()
        assert same >= 0.99, f"{n}: only {same:.4f} of the mask equals the reference's"


def test_awq_full_equals_the_reference_live(monkeypatch):
    """algorithm awq_full = awq_lite, then awq_clip on the scaled weights: same alphas, clipped block amax equal up to
    ties, pre_quant_scale within the statistics' summation order."""
    ref_shim.install()
    import modelopt.torch.quantization as mtq

    batches = _batches()
    cfg = copy.deepcopy(mtq.INT4_AWQ_CFG)
    cfg["algorithm"] = {"method": "awq_full", "debug": True}
    ref = mtq.quantize(_model(torch.float32), cfg, lambda m: [m(b) for b in batches])
    ref_state = {n: (float(m.awq_lite.best_alpha), m.weight_quantizer._amax.detach().float().reshape(-1).clone(),
                     m.input_quantizer._pre_quant_scale.detach().float().clone())
                 for n, m in ref.named_modules() if hasattr(m, "awq_lite")}
    hostmem_backend.install(monkeypatch, moa)
    ours = _model(torch.float32)
    cfg2 = copy.deepcopy(moa.model_quant.INT4_AWQ_CFG)
    cfg2["algorithm"] = {"method": "awq_full"}
    with torch.no_grad():
        moa.quantize(ours, cfg2, lambda m: [m(b) for b in batches])
    assert len(ref_state) == 14
    total = differing = 0
    for n, (alpha, amax, pqs) in ref_state.items():
        lin = ours.get_submodule(n)
        assert float(lin.awq_lite.best_alpha) == alpha, n
        got = lin.weight_quantizer._amax.detach().float().reshape(-1)
        total += amax.numel()
        differing += int(((got - amax).abs() > 1e-5 * amax.abs()).sum())
        assert ((lin.input_quantizer._pre_quant_scale.float() - pqs).abs() <= 1e-5 * pqs.abs()).all(), n
    assert differing <= 0.01 * total, f"awq_full: {differing} of {total} block amax values differ"


@pytest.mark.parametrize("search", ["gram", "gemm"])
def test_awq_lite_with_kv_cache_quantizers_both_search_modes_live(monkeypatch, search):
    """INT4_AWQ + FP8 KV cache: the KV quantizers calibrate during the cache pass and quantize during the search pass
    (model_calib.py:1574-1586), so the Gram search accumulates in the second pass; both search modes give the reference's
    KV amax, alphas and checkpoint."""
    ref_amax, ref_state = _reference_run("INT4_AWQ_CFG", torch.bfloat16, True)
    hostmem_backend.install(monkeypatch, moa)
    algo = {"method": "awq_lite", "alpha_step": 0.1, "search": search}
    our_amax, our_state = _our_run("INT4_AWQ_CFG", torch.bfloat16, True, algorithm=algo)
    assert any("k_bmm_quantizer" in n for n in ref_amax)
    for n, a in ref_amax.items():
        assert torch.equal(our_amax[n].reshape(-1), a.reshape(-1)), n
    ref_state.pop("__logits__"), our_state.pop("__logits__")
    _assert_same_quant_json(our_state.pop("__quant_json__"), ref_state.pop("__quant_json__"))  # FP8 KV cache entry included
    assert sorted(our_state) == sorted(ref_state)
    for k, want in ref_state.items():
        got = our_state[k].detach().cpu()
        assert torch.equal(got.contiguous().reshape(-1).view(torch.uint8), want.contiguous().reshape(-1).view(torch.uint8)), k


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fp8_per_channel_per_token_calibration_forward_and_export_equal_the_reference_live(monkeypatch, dtype):
    """FP8_PER_CHANNEL_PER_TOKEN_CFG: per-channel weight amax, the logits of the fake-quantized model (FP8 inputs with a
    dynamic per-token abs-max, block_sizes {-1: None} -> axis) and, since round 4, every byte of the fp8_pc_pt checkpoint
    (E4M3 weights from the fp32-promoted quotient, fp32 [Cout] weight_scale, no input_scale) equal the reference run live."""
    ref_amax, ref_state = _reference_run("FP8_PER_CHANNEL_PER_TOKEN_CFG", dtype, False)
    hostmem_backend.install(monkeypatch, moa)
    our_amax, our_state = _our_run("FP8_PER_CHANNEL_PER_TOKEN_CFG", dtype, False)
    assert len(ref_amax) == 14 and set(ref_amax) <= set(our_amax)
    for n, a in ref_amax.items():
        assert a.numel() > 1 and torch.equal(our_amax[n].reshape(-1), a.reshape(-1)), n
    assert torch.equal(our_state.pop("__logits__"), ref_state.pop("__logits__"))
    _assert_same_quant_json(our_state.pop("__quant_json__"), ref_state.pop("__quant_json__"))
    assert sorted(our_state) == sorted(ref_state)
    assert not any(k.endswith("input_scale") for k in ref_state)
    for k, want in ref_state.items():
        got = our_state[k].detach().cpu()
        assert got.dtype == want.dtype and tuple(got.shape) == tuple(want.shape), k
        assert torch.equal(got.contiguous().reshape(-1).view(torch.uint8), want.contiguous().reshape(-1).view(torch.uint8)), k


# ------------------------------------------------------------------------------------------------------------------
# The drop-in boundary under the REAL reference with real tensors: modelopt_plugin.install() + the
# reference's own mtq.quantize().  The reference only hands GPU tensors to its extension modules (`inputs.is_cuda`
# gates in tensor_quant.py:80, :374 and its device guard), and this tier has no GPU: for the duration of the test every
# tensor answers is_cuda = True, the reference's CUDA device guard is a no-op, and the C-ABI is served by the host-memory
# stand-in -- so the reference's UNMODIFIED call sites (tensor_quant.py:83-91, :103-111, :184-191, calib/max.py:63-64)
# drive our adapters with real data.  Everything must equal the un-installed reference run.
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("preset,expected", [
    ("FP8_DEFAULT_CFG", ["S1:fake_e4m3fy", "S6:reduce_amax"]),
    ("INT8_DEFAULT_CFG", ["S1:fake_tensor_quant", "S1:fake_tensor_quant_with_axis", "S6:reduce_amax"]),
    ("INT4_AWQ_CFG", ["S1:fake_tensor_quant_with_axis", "S6:reduce_amax"]),
    ("MXFP4_DEFAULT_CFG", ["S1:fused_amax_convert"]),
])
def test_reference_quantize_through_installed_seams(monkeypatch, preset, expected):
    import contextlib

    ref_shim.install()
    import modelopt.torch.quantization as mtq
    import modelopt.torch.quantization.extensions as ext
    from modelopt.torch.quantization.nn.modules import tensor_quantizer as ref_tq_mod
    from modelopt.torch.quantization.utils import core_utils
    from transformers import LlamaConfig, LlamaForCausalLM

    from model_optimizer_amd import modelopt_plugin

    def tiny():
        torch.manual_seed(0)
        cfg = LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                          num_key_value_heads=2, vocab_size=64, max_position_embeddings=32,
                          architectures=["LlamaForCausalLM"])
        return LlamaForCausalLM(cfg).to(torch.bfloat16).eval()

    batches = [torch.randint(0, 64, (2, 16), generator=torch.Generator().manual_seed(i)) for i in range(2)]

    def run():
        m = tiny()
        with torch.no_grad():
            q = mtq.quantize(m, copy.deepcopy(getattr(mtq, preset)), lambda mm: [mm(b) for b in batches])
            logits = q(batches[0]).logits
        state = {n: t.clone() for n, t in q.state_dict().items()}
        return state, logits

    if preset == "MXFP4_DEFAULT_CFG":
        base = None  # the reference has no CPU implementation of the MX kernels: nothing to compare a baseline with
    else:
        base = run()
    # -- install: seams + host-memory backend + "every tensor is a GPU tensor"
    saved = {fn: getattr(fn, "extension", None) for fn in (ext.get_cuda_ext, ext.get_cuda_ext_fp8, ext.get_cuda_ext_mx)}
    saved_reduce = (core_utils.reduce_amax, )
    hostmem_backend.install(monkeypatch, moa)
    try:
        installed = modelopt_plugin.install()
        assert "S1:extensions" in installed and "S6:reduce_amax" in installed
        monkeypatch.setattr(torch.Tensor, "is_cuda", property(lambda self: True), raising=False)
        monkeypatch.setattr(ref_tq_mod, "same_device_as", lambda t: contextlib.nullcontext())
        modelopt_plugin.STATS.clear()
        ours = run()
        stats = dict(modelopt_plugin.STATS)
    finally:
        monkeypatch.undo()
        for fn, v in saved.items():
            if v is None and hasattr(fn, "extension"):
                del fn.extension
            elif v is not None:
                fn.extension = v
    for key in expected:
        assert stats.get(key, 0) > 0, f"{preset}: the reference never reached {key}: {stats}"
    assert not [k for k in stats if "fallback" in k], f"{preset}: unexpected fallbacks {stats}"
    if base is not None:
        assert set(base[0]) == set(ours[0])
        for n in base[0]:
            assert torch.equal(base[0][n], ours[0][n]), f"{preset}: {n} differs from the un-installed reference run"
        assert torch.equal(base[1], ours[1]), f"{preset}: logits differ"
    else:
        # MXFP4 under the reference = its own TensorQuantizer / QuantLinear code calling our MX kernel through S1; this
        # package's quantize() of the same model calls the same kernel from its own host code: identical logits
        hostmem_backend.install(monkeypatch, moa)
        m = tiny()
        with torch.no_grad():
            q = moa.quantize(m, copy.deepcopy(moa.model_quant.MXFP4_DEFAULT_CFG), None)
            mine = q(batches[0]).logits
        assert torch.equal(mine, ours[1]), "MXFP4 logits: reference through the seams vs this package"


@pytest.mark.parametrize("rows,cols,cfg,fmt,bits", [
    (96, 1024, dict(num_bits=4, block_sizes={-1: 128, "type": "static"}), 1, 4),
    (64, 768, dict(num_bits=(4, 3), axis=None), 2, 8),
    (80, 640, dict(num_bits=8, axis=0), 1, 8)])
def test_gptq_blockwise_update_equals_the_reference_live(monkeypatch, rows, cols, cfg, fmt, bits):
    """The reference's own gptq_blockwise_update (utils/calib_utils.py:241-276) run HERE -- its quantizer fake-quantizing the
    whole matrix once per column -- against the oracle's column sweep + defined-order trailing update, from the same inverse
    factor, over several column blocks of a 640-1024 wide weight: the same weights, every one."""
    ref_shim.install()
    from modelopt.torch.quantization.config import QuantizerAttributeConfig
    from modelopt.torch.quantization.model_calib import max_calibrate
    from modelopt.torch.quantization.nn import TensorQuantizer
    from modelopt.torch.quantization.utils import calib_utils

    from oracle import oracle

    gen = torch.Generator().manual_seed(rows + cols)
    w = (torch.randn(rows, cols, generator=gen) * 0.05).float()
    mix = torch.randn(cols, cols, generator=gen) / cols ** 0.5 * torch.linspace(2, 0.05, cols)[:, None]
    x = torch.randn(4 * cols, cols, generator=gen) @ mix
    hinv = calib_utils.compute_hessian_inverse((2.0 / x.shape[0]) * x.t() @ x, w, 0.01).contiguous()
    q = TensorQuantizer(QuantizerAttributeConfig(**cfg))
    max_calibrate(q, lambda qq: qq(w), distributed_sync=False)
    want = w.clone()
    calib_utils.gptq_blockwise_update(want, hinv, 128, q)
    amax = q._amax.float().reshape(-1)
    stride, g = (0, cols) if amax.numel() == 1 else ((1, cols) if amax.numel() == rows else (cols // 128, 128))
    got = w.clone().contiguous()
    for i1 in range(0, cols, 128):
        delta = oracle.gptq_block_sweep(got, i1, 128, hinv, amax, stride, g, fmt, bits)
        if i1 + 128 < cols:
            oracle.sgpt_trailing_update(got, i1, delta, hinv)
    assert torch.equal(got, want), f"{int((got != want).sum())} of {got.numel()} weights differ"
    # and this package's own functions on the same inputs (host side through the oracle-backed C-ABI)
    hostmem_backend.install(monkeypatch, moa)
    ours_hinv = moa.gptq.compute_hessian_inverse((2.0 / x.shape[0]) * x.t() @ x, w, 0.01)
    assert torch.equal(ours_hinv, hinv)


def test_histogram_mse_threshold_equals_the_reference_live():
    """`HistogramCalibrator.compute_amax("mse")` for integer formats: the reference's call passes the bit width in the
    `bias` slot of fake_tensor_quant and the signedness in `num_bits` (calib/histogram.py:305-307 against
    tensor_quant.py:349-360); calib._compute_amax_mse returns the amax THAT computes, bit for bit, over widths,
    signedness, strides, start bins and value scales (count-weighted mean of the centres below / above the bit width)."""
    ref_shim.install()
    from modelopt.torch.quantization.calib import histogram as ref_hist

    checked = 0
    for seed in range(10):
        gen = torch.Generator().manual_seed(seed)
        nb = [2048, 300, 512, 257, 64][seed % 5]
        scale = [1.0, 300.0, 0.01, 20.0][seed % 4]
        x = (torch.randn(1 << 14, generator=gen) * torch.exp(0.7 * torch.randn(1 << 14, generator=gen))).abs() * scale
        hist = torch.histc(x, bins=nb, min=0, max=float(x.max()))
        edges = torch.linspace(0, float(x.max()), nb + 1)
        for bits in (8, 4, 0):
            for unsigned in (False, True):
                for stride, start in ((1, 128 if nb > 128 else 8), (3, 16), (7, 1)):
                    want = ref_hist._compute_amax_mse(hist.numpy(), edges.numpy(), bits, unsigned, stride, start)
                    got = moa.calib._compute_amax_mse(hist.to(torch.int64), edges, bits, unsigned, stride, start)
                    assert torch.equal(want.float().reshape(()), got.float().reshape(())), (seed, nb, bits, unsigned, stride, start)
                    checked += 1
    assert checked == 180
    # (4, 3): the reference's call is one argument short
    with pytest.raises(TypeError):
        ref_hist._compute_amax_mse(hist.numpy(), edges.numpy(), (4, 3), False, 1, 16)


@pytest.mark.parametrize("method,kwargs", [("percentile", {"percentile": 99.9}), ("entropy", {}), ("mse", {})])
def test_histogram_calibrated_inputs_of_a_model_equal_the_reference_live(monkeypatch, method, kwargs):
    """The manual flow of the histogram calibrators (config `"calibrator": "histogram"` on the input quantizers,
    enable_stats_collection, forward passes, `load_calib_amax(method, ...)` per quantizer): every amax of the tiny Llama
    equals the reference's for the three threshold searches."""
    ref_shim.install()
    import modelopt.torch.quantization as mtq
    from modelopt.torch.quantization import model_calib as ref_calib

    def with_histograms(cfg):
        cfg = copy.deepcopy(cfg)
        entries = cfg["quant_cfg"]
        spec = {"num_bits": 8, "axis": None, "calibrator": "histogram"}
        if isinstance(entries, list):
            for e in entries:
                if e.get("quantizer_name") == "*input_quantizer":
                    e["cfg"] = spec
        else:
            entries["*input_quantizer"] = spec
        cfg["algorithm"] = None
        return cfg

    def run(quantize, cfg, calib_mod, is_quantizer):
        model = _model(torch.bfloat16, "llama")
        with torch.no_grad():
            quantize(model, with_histograms(cfg), None)
            calib_mod.enable_stats_collection(model)
            for b in _batches():
                model(b)
            for _, m in model.named_modules():
                if is_quantizer(m) and not m._disabled and m._calibrator is not None:
                    if type(m._calibrator).__name__ == "MaxCalibrator":
                        m.load_calib_amax()
                    else:
                        m.load_calib_amax(method, **kwargs)
                    m.enable_quant()
                    m.disable_calib()
        return {n: m._amax.detach().float().clone() for n, m in model.named_modules()
                if is_quantizer(m) and m.is_enabled and getattr(m, "_amax", None) is not None}

    ref = run(mtq.quantize, mtq.INT8_DEFAULT_CFG, ref_calib, lambda m: type(m).__name__ == "TensorQuantizer")
    hostmem_backend.install(monkeypatch, moa)
    ours = run(moa.quantize, moa.model_quant.INT8_DEFAULT_CFG, moa.model_calib, lambda m: isinstance(m, moa.TensorQuantizer))
    assert len(ref) == 28 and set(ref) == set(ours)
    for n, a in ref.items():
        assert torch.equal(a.reshape(-1), ours[n].reshape(-1)), f"{method}: amax of {n}: {ours[n]} vs {a}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_constant_amax_quantizers_skip_calibration_like_the_reference_live(monkeypatch, dtype):
    """`constant_amax` pins `_amax` at configuration time (config.py:674-709, tensor_quantizer.py:256-261): the input
    quantizers neither calibrate nor disturb the statistics of the others, the forward and the exported input_scale use
    the pinned value.  Invalid combinations are refused at configuration time on both sides."""
    ref_shim.install()
    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint
    from safetensors import safe_open

    batches = _batches()
    loop = lambda m: [m(b) for b in batches]  # noqa: E731
    ref_cfg = copy.deepcopy(mtq.FP8_DEFAULT_CFG)
    ref_cfg["quant_cfg"] = list(ref_cfg["quant_cfg"]) + [{"quantizer_name": "*mlp*input_quantizer", "cfg": {"num_bits": (4, 3), "axis": None, "constant_amax": 96.0}}]
    ref = mtq.quantize(_model(dtype), ref_cfg, loop)
    ref_amax = {n: m._amax.detach().float().clone() for n, m in ref.named_modules()
                if type(m).__name__ == "TensorQuantizer" and m.is_enabled and getattr(m, "_amax", None) is not None}
    with torch.no_grad():
        ref_logits = ref(batches[0]).logits.clone()
    ref_state = {}
    with tempfile.TemporaryDirectory() as d:
        export_hf_checkpoint(ref, export_dir=d)
        with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
            ref_state = {k: f.get_tensor(k) for k in f.keys()}
    hostmem_backend.install(monkeypatch, moa)
    mq = moa.model_quant
    cfg = copy.deepcopy(mq.FP8_DEFAULT_CFG)
    cfg["quant_cfg"] = mq.normalize_quant_cfg_list(cfg["quant_cfg"]) + [{"quantizer_name": "*mlp*input_quantizer", "cfg": {"num_bits": (4, 3), "axis": None, "constant_amax": 96.0}}]
    ours = _model(dtype)
    with torch.no_grad():
        moa.quantize(ours, cfg, loop)
        logits = ours(batches[0]).logits.clone()
    our_amax = {n: m._amax.detach().float().clone() for n, m in ours.named_modules()
                if isinstance(m, moa.TensorQuantizer) and m.is_enabled and getattr(m, "_amax", None) is not None}
    assert set(our_amax) == set(ref_amax)
    for n, a in ref_amax.items():
        assert torch.equal(our_amax[n].reshape(-1), a.reshape(-1)), n
        if ".mlp." in n and n.endswith("input_quantizer"):
            assert float(a) == 96.0
    assert torch.equal(logits, ref_logits)
    state = moa.export.export_state_dict(ours, dtype, lambda: ours(torch.ones([1, 2], dtype=torch.long)))
    assert sorted(state) == sorted(ref_state)
    for k, want in ref_state.items():
        got = state[k].detach().cpu()
        assert got.dtype == want.dtype and torch.equal(got.reshape(-1).view(torch.uint8), want.reshape(-1).view(torch.uint8)), k
    for bad in ({"constant_amax": -1.0}, {"constant_amax": 3.0, "use_constant_amax": True}):
        with pytest.raises(AssertionError):
            moa.QuantizerAttributeConfig(num_bits=(4, 3), **bad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_block_grids_on_any_axes_equal_the_reference_live(monkeypatch, dtype):
    """Static block_sizes on any set of axes of tensors of rank 2-4 (tensor_quantizer.py:975-1043), whole and ragged:
    amax buffer (shape and values, running maximum over two calls), fake-quantized output and the dynamic-amax output."""
    ref_shim.install()
    from modelopt.torch.quantization.config import QuantizerAttributeConfig as RefCfg
    from modelopt.torch.quantization.nn import TensorQuantizer as RefQuantizer

    hostmem_backend.install(monkeypatch, moa)
    cases = [((4, 3), {0: 8}, (32, 24)), (8, {0: 8}, (20, 24)), (8, {0: 4, -1: 8}, (6, 5, 20)), ((4, 3), {1: 4}, (6, 10, 3, 3)),
             (4, {-2: 16}, (40, 16)), (8, {0: 2, 1: 3, 2: 4}, (4, 7, 8)), ((4, 3), {0: 64}, (8, 16)), (8, {0: 8, 1: 8}, (3, 8, 8)),
             ((4, 3), {-1: 32, -2: 16}, (3, 32, 64)), (8, {-1: 16, -2: 8}, (2, 3, 20, 40)), (8, {-1: 8, -2: 4}, (5, 8, 8)),
             (8, {-1: 16}, (3, 5, 40))]
    for nb, grid, shape in cases:
        x = (torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape))) * 0.3).to(dtype)
        got = []
        for Q, C in ((RefQuantizer, RefCfg), (moa.TensorQuantizer, moa.QuantizerAttributeConfig)):
            q = Q(C(num_bits=nb, block_sizes=dict(grid)))
            q.disable_quant(); q.enable_calib()
            q(x); q(x * 0.5)
            q.load_calib_amax()
            q.enable_quant(); q.disable_calib()
            got.append((q(x), q._amax, Q(C(num_bits=nb, block_sizes=dict(grid)))(x)))
        (ry, ra, rd), (y, a, d) = got
        what = f"{nb} {grid} {shape}"
        assert a.shape == ra.shape and a.dtype == ra.dtype and torch.equal(a.float(), ra.float()), what
        assert torch.equal(y, ry) and torch.equal(d, rd), what


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_quantization_axes_that_lie_apart_equal_the_reference_live(monkeypatch, dtype):
    """TensorQuantizer(axis=(0, 2)) on a rank-3 tensor, axis=(1, 3) on rank 4, dynamic and calibrated, INT8 / INT4 / FP8: the kept
    axes are not one adjacent block (found by tools/quantizer_fuzz.py on the device; the reference's eager path broadcasts such
    an amax, tensor_quant.py:607-645).  Also the assertion both sides raise for dynamic blocks without scale_bits."""
    ref_shim.install()
    from modelopt.torch.quantization.config import QuantizerAttributeConfig as RefCfg
    from modelopt.torch.quantization.nn import TensorQuantizer as RefQuantizer

    hostmem_backend.install(monkeypatch, moa)
    for nb, axis, shape in [(8, (0, 2), (3, 20, 16)), ((4, 3), (0, 2), (2, 7, 40)), (4, (1, 3), (2, 3, 5, 8)), (8, (0, -1), (4, 9, 24))]:
        x = (torch.randn(*shape, generator=torch.Generator().manual_seed(sum(shape))) * 0.4).to(dtype)
        got = []
        for Q, C in ((RefQuantizer, RefCfg), (moa.TensorQuantizer, moa.QuantizerAttributeConfig)):
            q = Q(C(num_bits=nb, axis=axis))
            q.disable_quant(); q.enable_calib()
            q(x); q(x * 0.5)
            q.load_calib_amax()
            q.enable_quant(); q.disable_calib()
            got.append((q(x), q._amax, Q(C(num_bits=nb, axis=axis))(x)))
        (ry, ra, rd), (y, a, d) = got
        what = f"{nb} axis {axis} {shape}"
        assert a.shape == ra.shape and a.dtype == ra.dtype and torch.equal(a.float(), ra.float()), what
        assert torch.equal(y, ry) and torch.equal(d, rd), what
    for Q, C in ((RefQuantizer, RefCfg), (moa.TensorQuantizer, moa.QuantizerAttributeConfig)):
        with pytest.raises(AssertionError):
            Q(C(num_bits=8, block_sizes={-1: 16, "type": "dynamic"}))(torch.randn(4, 32))


@pytest.mark.parametrize("preset,dtype,with_kv", [("INT4_AWQ_CFG", torch.bfloat16, False), ("FP8_DEFAULT_CFG", torch.float16, True),
                                                  ("INT8_SMOOTHQUANT_CFG", torch.bfloat16, False)])
def test_scale_math_on_the_tensors_own_device_equals_the_reference_on_that_device_live(monkeypatch, preset, dtype, with_kv):
    """numerics mode "device": the flows' small-vector scale math (AWQ scale tables, `amax / maxbound`, ...) as torch
    evaluates the reference's expressions where the statistics live.  In this tier that device is the host, so the result
    must equal the reference's (CPU) run like the default mode does; on the GPU the same test runs against the reference's
    GPU run (tests/test_gpu_reference_live.py, section B), which differs from its CPU run in the last bit of the scales."""
    ref_amax, ref_state = _reference_run(preset, dtype, with_kv)
    hostmem_backend.install(monkeypatch, moa)
    with moa.numerics.scale_math("device"):
        assert moa.numerics.mode() == "device"
        our_amax, our_state = _our_run(preset, dtype, with_kv)
    assert moa.numerics.mode() == "host"
    for n, a in ref_amax.items():
        assert torch.equal(our_amax[n].reshape(-1), a.reshape(-1)), n
    ref_state.pop("__quant_json__", None), our_state.pop("__quant_json__", None)
    assert torch.equal(our_state.pop("__logits__"), ref_state.pop("__logits__"))
    assert sorted(our_state) == sorted(ref_state)
    for k, want in ref_state.items():
        got = our_state[k].detach().cpu()
        assert torch.equal(got.contiguous().reshape(-1).view(torch.uint8), want.contiguous().reshape(-1).view(torch.uint8)), k
