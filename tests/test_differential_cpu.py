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


def _model(dtype):
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(7)
    return LlamaForCausalLM(LlamaConfig(architectures=["LlamaForCausalLM"], **CFG)).to(dtype).eval()


def _batches():
    return [torch.randint(0, CFG["vocab_size"], (3, 24), generator=torch.Generator().manual_seed(40 + i)) for i in range(3)]


def _reference_run(preset, dtype, with_kv):
    ref_shim.install()
    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint
    from safetensors import safe_open

    model = _model(dtype)
    cfg = copy.deepcopy(getattr(mtq, preset))
    if with_kv:
        cfg = mtq.update_quant_cfg_with_kv_cache_quant(cfg, copy.deepcopy(mtq.FP8_KV_CFG["quant_cfg"]))
    batches = _batches()
    loop = (lambda m: [m(b) for b in batches]) if cfg.get("algorithm") else None
    q = mtq.quantize(model, cfg, loop)
    amax = {n: m._amax.detach().float().clone() for n, m in q.named_modules()
            if type(m).__name__ == "TensorQuantizer" and m.is_enabled and getattr(m, "_amax", None) is not None}
    out = {}
    with tempfile.TemporaryDirectory() as d:
        export_hf_checkpoint(q, export_dir=d)
        with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
            for k in f.keys():
                out[k] = f.get_tensor(k)
    return amax, out


def _our_run(preset, dtype, with_kv):
    mq = moa.model_quant
    model = _model(dtype)
    cfg = copy.deepcopy(getattr(mq, preset))
    if with_kv:
        cfg = mq.update_quant_cfg_with_kv_cache_quant(cfg, mq.FP8_KV_CFG["quant_cfg"])
    batches = _batches()
    with torch.no_grad():
        moa.quantize(model, cfg, (lambda m: [m(b) for b in batches]) if cfg.get("algorithm") else None)
    amax = {n: m._amax.detach().float().clone() for n, m in model.named_modules()
            if isinstance(m, moa.TensorQuantizer) and m.is_enabled and getattr(m, "_amax", None) is not None}
    state = moa.export.export_state_dict(model, dtype, lambda: model(torch.ones([1, 2], dtype=torch.long)))
    return amax, state


@pytest.mark.parametrize("preset,dtype,with_kv", [
    ("FP8_DEFAULT_CFG", torch.bfloat16, False), ("FP8_DEFAULT_CFG", torch.float32, True), ("FP8_DEFAULT_CFG", torch.float16, True),
    ("INT8_SMOOTHQUANT_CFG", torch.bfloat16, False), ("INT8_SMOOTHQUANT_CFG", torch.float32, False),
    ("INT8_DEFAULT_CFG", torch.bfloat16, False),
    ("FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG", torch.bfloat16, False), ("MXFP4_DEFAULT_CFG", torch.bfloat16, False),
    ("MXFP4_DEFAULT_CFG", torch.float16, False), ("INT4_BLOCKWISE_WEIGHT_ONLY_CFG", torch.bfloat16, False),
])
def test_quantize_and_export_equal_the_reference_live(monkeypatch, preset, dtype, with_kv):
    ref_amax, ref_state = _reference_run(preset, dtype, with_kv)
    hostmem_backend.install(monkeypatch, moa)
    our_amax, our_state = _our_run(preset, dtype, with_kv)
    # every enabled quantizer the reference calibrated exists here under the same name with the same amax
    for n, a in ref_amax.items():
        assert n in our_amax, f"{preset}: quantizer {n} has no amax here"
        assert torch.equal(our_amax[n].reshape(-1), a.reshape(-1)), f"{preset}: amax of {n} differs"
    assert sorted(our_state) == sorted(ref_state), set(our_state) ^ set(ref_state)
    for k, want in ref_state.items():
        got = our_state[k].detach().cpu()
        assert got.dtype == want.dtype and tuple(got.shape) == tuple(want.shape), f"{preset} {k}: {got.dtype} {tuple(got.shape)} vs {want.dtype} {tuple(want.shape)}"
        assert torch.equal(got.contiguous().reshape(-1).view(torch.uint8), want.contiguous().reshape(-1).view(torch.uint8)), f"{preset}: {k} differs"
