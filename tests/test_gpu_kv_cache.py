"""FP8 KV-cache quantizers ("KV-cache quantizers (k/v_bmm_quantizer) use the same amax path") against
the reference run on the tiny Llama (tests/golden/export_llama_fp8_kv.npz: FP8_DEFAULT_CFG + FP8_KV_CFG, sdpa and
eager attention): which bmm quantizers are enabled, their calibrated amax, the logits with KV fake-quant active,
the exported k_scale / v_scale and hf_quant_config.

Tolerance: the model runs in fp32; the reference's numbers come from CPU GEMMs, ours from GPU GEMMs, so the key /
value states differ in the last bits -- amax is compared to 1e-5 relative, logits to 2 % of their RMS (FP8 rounding
of activations can flip on last-bit differences)."""

import pytest
import torch

import _moa_import
from conftest import assert_bits_equal, from_bits

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import hf_attention, ops  # noqa: E402

DEV = "cuda:0"


def _build(g, cases, impl):
    from transformers import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(architectures=["LlamaForCausalLM"], **cases["config"])
    cfg._attn_implementation = impl
    with torch.device("cpu"):
        model = LlamaForCausalLM(cfg).to(torch.float32)
    sd = {k[len("orig/"):]: from_bits(g.raw(k), torch.float32) for k in g.z.files if k.startswith("orig/")}
    assert not model.load_state_dict(sd, strict=False).unexpected_keys
    return model.to(DEV).eval()


@pytest.mark.parametrize("impl", ["sdpa", "eager"])
def test_fp8_kv_cache_calibration_and_export_match_reference(golden, impl):
    g = golden("export_llama_fp8_kv")
    cases = g.cases
    mq = moa.model_quant
    model = _build(g, cases, impl)
    batches = [torch.from_numpy(g.raw(f"tokens{i}")).to(DEV) for i in range(cases["n_batches"])]
    cfg = mq.update_quant_cfg_with_kv_cache_quant(mq.FP8_DEFAULT_CFG, mq.FP8_KV_CFG["quant_cfg"])
    with torch.no_grad():
        mq.quantize(model, cfg, lambda m: [m(b) for b in batches])
    attns = [n for n, m in model.named_modules() if hf_attention.is_quantized_attention(m)]
    assert attns == cases["attentions"]
    for n in attns:
        m = model.get_submodule(n)
        for which in "qkv":
            tq = getattr(m, f"{which}_bmm_quantizer")
            assert tq.is_enabled == bool(g.raw(f"{impl}/{n}.{which}_enabled")), f"{n}.{which} enabled"
            if tq.is_enabled:
                want = from_bits(g.raw(f"{impl}/{n}.{which}_amax"), torch.float32)
                got = tq._amax.float().cpu()
                assert got.shape == want.shape
                assert ((got - want).abs() <= 1e-5 * want.abs()).all(), f"{n}.{which}_amax {got} vs {want}"
    with torch.no_grad():
        logits = model(batches[0]).logits.float().cpu()
    want = from_bits(g.raw(f"{impl}/logits"), torch.float32)
    assert ((logits - want).pow(2).mean().sqrt() <= 0.02 * want.pow(2).mean().sqrt())
    if impl != "sdpa":
        return
    state = moa.export.export_state_dict(model, torch.float32)
    for key in cases["dtypes"]:
        got, want = state[key].cpu(), from_bits(g.raw(f"exp/{key}"), torch.float32)
        assert got.dtype == torch.float32 and got.shape == want.shape, key
        assert ((got - want).abs() <= 1e-5 * want.abs()).all(), key
        attn = model.get_submodule(key.rsplit(".", 2)[0])
        amax = getattr(attn, f"{key[-7]}_bmm_quantizer")._amax.float().cpu()
        assert_bits_equal(got, amax / 448.0, f"{key} == amax / 448")
    assert sorted(state) == cases["exported_keys"]
    assert moa.export.hf_quant_config(model)["quantization"] == cases["hf_quant_config"]["quantization"]
    assert cases["hf_quant_config"]["quantization"]["kv_cache_quant_algo"] == "FP8"


def test_per_tensor_entries_walk_permuted_dense_tensors_in_place():
    """Key / value states are `[B, S, heads, D].transpose(1, 2)`: dense but not contiguous.  The per-tensor amax and
    QDQ entries must give the contiguous result, keep the input's strides, and not depend on the permutation."""
    torch.manual_seed(5)
    base = torch.randn(4, 48, 6, 64, device=DEV, dtype=torch.bfloat16)
    kv = base.transpose(1, 2)
    assert not kv.is_contiguous() and ops._is_dense(kv)
    a = ops.reduce_amax(kv)
    assert_bits_equal(a, ops.reduce_amax(kv.contiguous()), "amax of a permuted view")
    buf = torch.zeros(1, device=DEV)
    ops.reduce_amax(kv, out=buf, accumulate=True)
    assert buf.item() == a.float().item()
    for fn in (lambda x: ops.scaled_e4m3(x, a.float()), lambda x: ops.scaled_e4m3(x, None),
               lambda x: ops.fake_tensor_quant(x, a.float(), 8, False, True)):
        y = fn(kv)
        assert y.stride() == kv.stride() and y.shape == kv.shape
        assert_bits_equal(y, fn(kv.contiguous()), "QDQ of a permuted view")
    # a sliced (non-dense) view still works through the copying path
    sl = base[:, :, :, ::2]
    assert not ops._is_dense(sl)
    assert_bits_equal(ops.scaled_e4m3(sl, a.float()), ops.scaled_e4m3(sl.contiguous(), a.float()), "strided view")
    assert_bits_equal(ops.reduce_amax(sl), ops.reduce_amax(sl.contiguous()), "strided amax")
    # per-channel amax on a permuted view goes through the copying path too
    am = ops.reduce_amax(kv, axis=[0, 2, 3])
    assert_bits_equal(am, ops.reduce_amax(kv.contiguous(), axis=[0, 2, 3]), "axis amax")
