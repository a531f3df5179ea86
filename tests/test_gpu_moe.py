"""Fused MoE experts (BASELINE configs[3]: Mixtral FP8) against the reference run on a tiny fp32 Mixtral
(tests/golden/moe_fp8.npz): which quantizers exist and are enabled, per-expert weight amax (bit-exact: a max of
representable values, computed from the weights alone -- also for the expert no token was routed to), shared input
amax (1e-5: CPU vs GPU fp32 GEMM order), logits with fake-quant active, and the exported per-expert checkpoint tensors
(FP8 bytes of gate / up / down projections cut out of the fused tensors, scales, checkpoint key names)."""

import numpy as np
import pytest
import torch

import _moa_import
from conftest import from_bits

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import hf_experts  # noqa: E402

DEV = "cuda:0"


def _build(g, cases):
    from transformers import MixtralConfig, MixtralForCausalLM

    cfg = MixtralConfig(architectures=["MixtralForCausalLM"], **cases["config"])
    with torch.device("cpu"):
        model = MixtralForCausalLM(cfg).to(torch.float32)
    sd = {k[len("orig/"):]: from_bits(g.raw(k), torch.float32) for k in g.z.files if k.startswith("orig/")}
    assert not model.load_state_dict(sd, strict=False).unexpected_keys
    return model.to(DEV).eval()


def test_mixtral_fp8_calibration_and_export_match_reference(golden):
    g = golden("moe_fp8")
    cases = g.cases
    mq = moa.model_quant
    model = _build(g, cases)
    batches = [torch.from_numpy(g.raw(f"tokens{i}")).to(DEV) for i in range(cases["n_batches"])]
    with torch.no_grad():
        mq.quantize(model, mq.FP8_DEFAULT_CFG, lambda m: [m(b) for b in batches])
    ours = {n: m for n, m in model.named_modules() if isinstance(m, moa.TensorQuantizer)}
    # the reference's quantizer set, name by name (its attention p/q/k/v bmm quantizers included), and what is enabled
    # (the reference also wraps nn.Embedding; those quantizers are disabled by every preset and not mirrored)
    ref_q = {n: en for n, en in cases["quantizers"].items() if "embed_tokens" not in n}
    assert all(not en for n, en in cases["quantizers"].items() if "embed_tokens" in n)
    assert set(ours) == set(ref_q), set(ours) ^ set(ref_q)
    for n, en in ref_q.items():
        assert ours[n].is_enabled == en, n
    n_weight = n_act = 0
    for key in g.z.files:
        if not key.startswith("amax/"):
            continue
        n = key[len("amax/"):]
        want = from_bits(g.raw(key), torch.float32)
        got = ours[n]._amax.float().cpu()
        assert got.shape == want.shape, n
        if "weight_quantizer" in n:
            assert torch.equal(got, want), f"{n}: {got} vs {want}"
            n_weight += 1
        else:
            assert ((got - want).abs() <= 1e-5 * want.abs()).all(), f"{n}: {got} vs {want}"
            n_act += 1
    assert n_weight == 2 * (4 + 2 * 4) and n_act >= 2 * (4 + 2)
    with torch.no_grad():
        logits = model(batches[0]).logits.float().cpu()
    want = from_bits(g.raw("logits"), torch.float32)
    assert ((logits - want).pow(2).mean().sqrt() <= 0.02 * want.pow(2).mean().sqrt())

    state = moa.export.export_state_dict(model, torch.float32)
    assert sorted(state) == sorted(cases["dtypes"]), set(state) ^ set(cases["dtypes"])
    checked = 0
    for key, dts in cases["dtypes"].items():
        if f"exp/{key}" not in g.z.files:
            continue
        got = state[key].detach().cpu().contiguous()
        if dts == "torch.float8_e4m3fn":
            assert got.dtype == torch.float8_e4m3fn
            want_b = g.raw(f"exp/{key}")
            same = (got.view(torch.uint8).numpy() == want_b).mean()
            # bytes depend on weight_scale only (bit-exact amax) -> identical
            assert same == 1.0, f"{key}: {same:.4f} of the fp8 bytes equal"
        else:
            want = from_bits(g.raw(f"exp/{key}"), torch.float32)
            assert got.dtype == torch.float32 and got.shape == want.shape, key
            if key.endswith("weight_scale") or key.endswith("gate.weight"):
                assert torch.equal(got, want), key
            else:
                assert ((got - want).abs() <= 1e-5 * want.abs()).all(), key
        checked += 1
    assert checked == 2 * (4 * 9 + 1)
    qc = moa.export.hf_quant_config(model)["quantization"]
    assert qc == cases["hf_quant_config"]["quantization"] and qc["quant_algo"] == "FP8"  # (incl. the routers in exclude_modules)


def test_expert_slices_join_the_multi_tensor_amax_launch():
    """weight_only_quantize puts every expert slice of every fused container into ONE abs-max launch; the result is
    each slice's own abs-max."""
    from transformers import MixtralConfig, MixtralForCausalLM

    torch.manual_seed(3)
    cfg = MixtralConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=1, num_attention_heads=4,
                        num_key_value_heads=2, vocab_size=64, num_local_experts=8, num_experts_per_tok=2)
    model = MixtralForCausalLM(cfg).to(torch.bfloat16).to(DEV)
    mq = moa.model_quant
    moa.nn.replace_quant_module(model)
    mq.set_quantizer_by_cfg(model, mq.FP8_DEFAULT_CFG["quant_cfg"])
    calls = []
    orig = moa.multi_tensor.SegmentTable.calibrate_amax
    moa.multi_tensor.SegmentTable.calibrate_amax = lambda self, *a, **k: (calls.append(self.n_seg), orig(self, *a, **k))[1]
    try:
        moa.model_calib.max_calibrate(model, None)
    finally:
        moa.multi_tensor.SegmentTable.calibrate_amax = orig
    assert calls == [4 + 2 * 8]  # q, k, v, o + 8 experts x (gate_up, down); lm_head is disabled
    ex = model.model.layers[0].mlp.experts
    assert hf_experts.is_quant_fused_experts(ex)
    for idx in range(8):
        assert ex.gate_up_proj_weight_quantizers[idx].amax.float().item() == ex.gate_up_proj[idx].abs().max().float().item()
        assert ex.down_proj_weight_quantizers[idx].amax.float().item() == ex.down_proj[idx].abs().max().float().item()
