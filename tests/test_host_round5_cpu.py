"""Host logic added in round 5, pinned WITHOUT the reference (the live comparisons are in tests/test_differential_cpu.py, which
needs the reference checkout): the order of the exported checkpoint dict, the module-name mapper of the quantization tables,
the partial attribute setter, need_calibration, the quantizer's printed form."""

import re

import pytest
import torch
from torch import nn

import _moa_import

moa = _moa_import.load()
ex, mq = moa.export, moa.model_quant
Cfg, TQ = moa.QuantizerAttributeConfig, moa.TensorQuantizer


class _Attn(nn.Module):
    def __init__(self):
        super().__init__()
        self.q_proj, self.k_proj = nn.Linear(8, 8, bias=True), nn.Linear(8, 8, bias=False)
        self.k_bmm_quantizer = TQ(Cfg(num_bits=(4, 3), axis=None))
        self.k_bmm_quantizer.amax = 1.0


class _Experts(nn.Module):
    def __init__(self):
        super().__init__()
        self.gate_up_proj, self.down_proj = nn.Parameter(torch.zeros(2, 16, 8)), nn.Parameter(torch.zeros(2, 8, 8))


class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        self.embed = nn.Embedding(4, 8)
        self.attn, self.experts, self.norm = _Attn(), _Experts(), nn.LayerNorm(8)


def test_the_exported_dict_follows_the_module_tree_with_scales_behind_their_linear_and_experts_expanded_in_place():
    model = _Toy()
    keys = ["norm.bias", "attn.k_proj.k_scale", "experts.1.down_proj.weight_scale", "experts.0.up_proj.weight", "attn.q_proj.input_scale",
            "experts.1.gate_proj.weight", "attn.q_proj.weight_scale", "something.else", "experts.0.gate_proj.weight", "attn.k_proj.weight",
            "attn.q_proj.pre_quant_scale", "embed.weight", "attn.q_proj.bias", "experts.0.down_proj.weight", "attn.q_proj.weight",
            "experts.1.down_proj.weight", "norm.weight", "experts.0.gate_proj.weight_scale"]
    got = list(ex._in_module_tree_order({k: i for i, k in enumerate(keys)}, model))
    assert got == ["embed.weight",
                   "attn.q_proj.weight", "attn.q_proj.bias", "attn.q_proj.weight_scale", "attn.q_proj.input_scale", "attn.q_proj.pre_quant_scale",
                   "attn.k_proj.weight", "attn.k_proj.k_scale",  # (the KV scale where <attn>.k_bmm_quantizer._amax stood)
                   "experts.0.gate_proj.weight", "experts.0.gate_proj.weight_scale", "experts.0.up_proj.weight", "experts.0.down_proj.weight",
                   "experts.1.gate_proj.weight", "experts.1.down_proj.weight", "experts.1.down_proj.weight_scale",
                   "norm.weight", "norm.bias", "something.else"]


def test_the_name_mapper_applies_anchored_rules_to_bare_names_and_keeps_a_trailing_wildcard(monkeypatch):
    rules = [("expert", "gate_proj", "w1"), ("regex", re.compile(r"^lm_head\."), "embed_out.", ()),
             ("regex", re.compile(r"\.mlp\."), ".block_sparse_moe.", ())]
    monkeypatch.setattr(ex, "_checkpoint_rename_rules", lambda model: rules)
    monkeypatch.setattr(ex, "_keeps_module_names", lambda model: False)
    name = ex._module_name_mapper(object())
    assert name("lm_head") == "embed_out" and name("lm_head*") == "embed_out*"  # a rule that ends in the separator still matches
    assert name("model.layers.0.mlp.experts.3.gate_proj") == "model.layers.0.block_sparse_moe.experts.3.w1"
    assert name("model.layers.0.mlp.gate.*") == "model.layers.0.block_sparse_moe.gate.*" and name("model.norm") == "model.norm"
    assert ex._rename_key("lm_head.weight", rules) == "embed_out.weight" and ex._rename_key("x.lm_head.weight", rules) == "x.lm_head.weight"


def test_partial_attributes_merge_into_a_quantizer_and_keep_what_was_calibrated():
    q = TQ(Cfg(num_bits=8, axis=0))
    q.amax = torch.ones(4, 1)
    q.update_attributes({"num_bits": 4, "enable": False})
    assert q.num_bits == 4 and not q.is_enabled and q.axis == 0 and torch.equal(q._amax, torch.ones(4, 1))
    q.update_attributes({"block_sizes": {-1: 16}, "enable": True})
    assert q.axis is None and q._calibrator._axis is None and q.block_sizes == {-1: 16} and q.is_enabled
    with pytest.raises(RuntimeError, match="Changing shape"):  # the constant is pinned on the buffer, which keeps its shape
        q.update_attributes({"constant_amax": 2.5})
    fresh = TQ(Cfg(num_bits=(4, 3), axis=None))
    fresh.update_attributes({"constant_amax": 2.5})
    assert fresh._constant_amax == 2.5 and float(fresh._amax) == 2.5
    with pytest.raises(AssertionError, match="not a valid"):
        q.update_attributes({"no_such_attribute": 1})
    with pytest.raises(moa.MoquantUnsupported):
        q.update_attributes({"rotate": True})


def test_need_calibration_reads_the_algorithm_and_the_non_weight_entries():
    assert mq.need_calibration({"quant_cfg": {"*weight_quantizer": {"num_bits": 8, "axis": 0}, "*input_quantizer": {"enable": False}}, "algorithm": "max"}) is False
    assert mq.need_calibration({"quant_cfg": {"*weight_quantizer": {"num_bits": 8}, "*input_quantizer": {"num_bits": 8}}, "algorithm": "max"}) is True
    assert mq.need_calibration({"quant_cfg": {"*input_quantizer": {"num_bits": 8, "type": "dynamic"}}, "algorithm": None}) is False
    assert mq.need_calibration({"quant_cfg": {"*input_quantizer": {"enable": False}}, "algorithm": "awq_lite"}) is True
    assert mq.need_calibration({"quant_cfg": [{"quantizer_name": "*x_quantizer", "cfg": [{"num_bits": 4}, {"num_bits": (4, 3)}]}], "algorithm": None}) is True


def test_a_quantizer_prints_in_the_references_words():
    q = TQ(Cfg(num_bits=(4, 3), axis=None))
    assert repr(q) == "TensorQuantizer((4, 3) bit fake per-tensor amax=dynamic calibrator=MaxCalibrator quant)"
    q.amax = 0.5
    assert "amax=5.00e-01" in repr(q)
    p = TQ(Cfg(num_bits=8, axis=0, unsigned=True, narrow_range=True))
    p.amax = torch.tensor([[1.0], [3.0]])
    assert repr(p) == "TensorQuantizer(unsigned 8 bit narrow fake axis=0 amax=[1.00e+00, 3.00e+00](2) calibrator=MaxCalibrator quant)"
    p.disable()
    assert repr(p) == "TensorQuantizer(disabled)"
    mx = TQ(Cfg(num_bits=(2, 1), block_sizes={-1: 32, "type": "dynamic", "scale_bits": (8, 0)}))
    assert "block_sizes={-1: 32, 'type': 'dynamic', 'scale_bits': (8, 0)}, amax=None" in repr(mx)
    const = TQ(Cfg(num_bits=(4, 3), axis=None, use_constant_amax=True))
    assert "amax=4.48e+02(const)" in repr(const)
