"""Host logic added or corrected in round 6, pinned on this tier through the host-memory stand-in for the C-ABI:
SmoothQuant's fold riding inside the MX quantize-dequantize (one launch), attribute-only reconfiguration of a quantizer,
calibration loops under torch.inference_mode(), the probation of the automatic per-layer statistics, the all-or-nothing
checkpoint renaming and the cap of the automatic activation store."""

import copy
import os
import sys
import types
import warnings

import pytest
import torch
from torch import nn

import _moa_import

moa = _moa_import.load()
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import hostmem_backend  # noqa: E402

mq, mc, ex = moa.model_quant, moa.model_calib, moa.export
Cfg, TQ = moa.QuantizerAttributeConfig, moa.TensorQuantizer


@pytest.fixture
def hostmem(monkeypatch):
    return hostmem_backend.install(monkeypatch, moa)


def _llama(dtype=torch.bfloat16):
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(7)
    cfg = LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                      vocab_size=96, max_position_embeddings=64, architectures=["LlamaForCausalLM"])
    return LlamaForCausalLM(cfg).to(dtype).eval()


def _batches():
    return [torch.randint(0, 96, (3, 24), generator=torch.Generator().manual_seed(40 + i)) for i in range(3)]


# ---------------------------------------------------------------------------------------- the fold inside the MX launch
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_smoothquant_with_the_fold_inside_the_mx_launch_equals_smooth_then_fold_weight(hostmem, monkeypatch, dtype):
    """smoothquant(fold_weights=True) on an MXFP4 model (BASELINE configs[4]): ONE fused launch over all smoothed linears instead
    of fold + re-calibration + QDQ per linear -- and the same model afterwards as smoothquant() followed by fold_weight(), bit
    for bit: weights, quantizer state, logits."""
    launches = []
    original = moa.multi_tensor.SegmentTable.fold_mx_fused

    def counted(self, *a, **k):
        launches.append(self.n_seg)
        return original(self, *a, **k)

    monkeypatch.setattr(moa.multi_tensor.SegmentTable, "fold_mx_fused", counted)

    def run(inside):
        m, batches = _llama(dtype), _batches()
        cfg = copy.deepcopy(mq.MXFP4_SMOOTHQUANT_CFG)
        cfg["algorithm"] = {**cfg["algorithm"], "fold_weights": inside}
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            moa.quantize(m, cfg, lambda mm: [mm(b) for b in batches])
            if not inside:
                mq.fold_weight(m)
            return {k: v.clone() for k, v in m.state_dict().items()}, m(batches[0]).logits, m

    two_steps = run(False)
    assert launches == []
    one_launch = run(True)
    assert launches == [14], launches  # every smoothed linear of the two layers, one table
    assert sorted(two_steps[0]) == sorted(one_launch[0])
    for k, v in two_steps[0].items():
        assert torch.equal(v, one_launch[0][k]), k
    assert torch.equal(two_steps[1], one_launch[1])
    for (n, a), (_, b) in zip(two_steps[2].named_modules(), one_launch[2].named_modules()):
        if isinstance(a, TQ):
            assert a._disabled == b._disabled and sorted(a._buffers) == sorted(b._buffers), n


def test_a_linear_whose_format_needs_recalibration_keeps_the_two_step_fold(hostmem):
    """INT8 SmoothQuant with fold_weights=True: nothing rides (the per-channel weight amax must be taken from the FOLDED weight),
    the call is smoothquant() + fold_weight()."""
    def run(inside):
        m, batches = _llama(torch.float32), _batches()
        cfg = copy.deepcopy(mq.INT8_SMOOTHQUANT_CFG)
        cfg["algorithm"] = {"method": "smoothquant", "alpha": 0.5, "fold_weights": inside}
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            moa.quantize(m, cfg, lambda mm: [mm(b) for b in batches])
            if not inside:
                mq.fold_weight(m)
            return {k: v.clone() for k, v in m.state_dict().items()}

    a, b = run(False), run(True)
    assert sorted(a) == sorted(b) and all(torch.equal(a[k], b[k]) for k in a)


# ---------------------------------------------------------------------------------------- ADVICE round 5, #1
def test_a_temporary_configuration_keeps_calibration_and_smoothing(hostmem):
    """set_quantizer_by_cfg_context / set_quantizer_attributes_full go through set_from_attribute_config, which -- like the
    reference's (nn/modules/tensor_quantizer.py:228-290) -- writes ATTRIBUTES only: `_amax`, `_pre_quant_scale` and the affine
    offset survive, and the block-layout caches filled under the temporary layout do not."""
    lin = moa.nn.QuantLinear.convert(nn.Linear(64, 32))
    lin.weight_quantizer.set_from_attribute_config(Cfg(num_bits=8, axis=0))
    lin.weight_quantizer.amax = torch.full((32, 1), 0.5)
    lin.input_quantizer.amax = 2.0
    lin.input_quantizer._enable_pre_quant_scale = True
    lin.input_quantizer.pre_quant_scale = torch.full((64,), 1.5)
    model = nn.Sequential(lin)
    with mq.set_quantizer_by_cfg_context(model, {"*input_quantizer": {"num_bits": (2, 1), "block_sizes": {-1: 16, "type": "dynamic", "scale_bits": (8, 0)}}}):
        assert lin.input_quantizer.num_bits == (2, 1)
        lin.input_quantizer(torch.randn(3, 64))  # (fills the layout caches of the temporary block layout)
    iq, wq = lin.input_quantizer, lin.weight_quantizer
    assert iq.num_bits == 8 and iq.block_sizes is None and iq.is_enabled
    assert iq.amax is not None and float(iq.amax) == 2.0
    assert iq.pre_quant_scale is not None and torch.equal(iq.pre_quant_scale, torch.full((64,), 1.5))
    assert wq.amax is not None and wq.amax.shape == (32, 1)
    assert not any(k in iq.__dict__ for k in TQ._LAYOUT_CACHES)
    mq.set_quantizer_attributes_full(model, "*weight_quantizer", Cfg(num_bits=8, axis=0, narrow_range=True))
    assert wq.narrow_range and wq.amax is not None and torch.equal(wq.amax, torch.full((32, 1), 0.5))


# ---------------------------------------------------------------------------------------- ADVICE round 5, #2
def test_calibration_loops_under_inference_mode(hostmem):
    """A forward_loop that runs under torch.inference_mode() hands the quantizers tensors WITHOUT a version counter: the
    per-layer deferred statistics, AWQ's input store and GPTQ's shared-input shortcut all recognise an unchanged tensor by
    that counter and must simply not be used for such tensors -- same results as under no_grad."""
    def run(preset, algorithm, ctx):
        m, batches = _llama(torch.float32), _batches()
        cfg = copy.deepcopy(getattr(mq, preset))
        if algorithm is not None:
            cfg["algorithm"] = algorithm

        def loop(mm):
            with ctx():
                for b in batches:
                    mm(b)

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            moa.quantize(m, cfg, loop)
        return {n: q._amax.clone() for n, q in m.named_modules() if isinstance(q, TQ) and getattr(q, "_amax", None) is not None}

    for preset, algorithm in (("FP8_DEFAULT_CFG", None), ("INT4_AWQ_CFG", None),
                              ("INT4_BLOCKWISE_WEIGHT_ONLY_CFG", {"method": "gptq", "block_size": 128})):
        a = run(preset, algorithm, torch.no_grad)
        b = run(preset, algorithm, torch.inference_mode)
        assert sorted(a) == sorted(b) and len(a) > 0, preset
        for n in a:
            assert torch.equal(a[n], b[n]), (preset, n)


def test_deferred_statistics_never_note_an_inference_tensor():
    batch = moa.calib.DeferredAmax("cpu")
    with torch.inference_mode():
        x = torch.randn(4, 8)
    assert batch.add(x, 0, torch.zeros(1)) is False and not batch.entries


def test_the_automatic_deferral_watches_first_and_backs_off_on_an_in_place_write():
    """defer_stats=None (nobody vouched for the model): the first pass through every flush point answers each request with its
    own launch and only watches the version counters.  A model that writes a noted tensor in place before its layer ends
    switches deferral OFF for the calibration -- no error, no lost statistic; a clean first pass switches it on."""
    dirty = moa.calib.DeferredAmax("cpu", probation=True)
    x = torch.randn(4, 8)
    assert dirty.add(x, 0, torch.zeros(1)) is False  # (the caller launches this request itself)
    x.mul_(2.0)
    dirty.flush(key=1)
    assert dirty.disabled and dirty.stats.get("disabled_by_inplace_write")
    assert dirty.add(torch.randn(4, 8), 0, torch.zeros(1)) is False
    clean = moa.calib.DeferredAmax("cpu", probation=True)
    for key in (1, 2):  # a pass in which nobody asked (every calibrator's first collect takes the general path): proves nothing
        clean.flush(key=key)
    for key in (1, 2):  # second batch: layers 1 and 2, requests watched
        assert clean.add(torch.randn(4, 8), 0, torch.zeros(1)) is False
        clean.flush(key=key)
        assert clean.probation and not clean.disabled
    clean.add(torch.randn(4, 8), 0, torch.zeros(1))
    clean.flush(key=1)  # layer 1 again: a whole pass went through every flush point with its tensors untouched
    assert not clean.probation and not clean.disabled
    counted = moa.calib.DeferredAmax("cpu", probation=True, flush_points=2)  # the caller knows how many layers a pass has
    for key in (1, 2):
        counted.flush(key=key)
    assert counted.probation  # (nothing was watched in that pass)
    for key in (1, 2):
        counted.add(torch.randn(4, 8), 0, torch.zeros(1))
        counted.flush(key=key)
    assert not counted.probation and not counted.disabled  # on from the next batch's FIRST layer
    strict = moa.calib.DeferredAmax("cpu")  # defer_stats=True: deferred from the first request, a write is an error
    y = torch.randn(4, 8)
    assert strict.add(y, 0, torch.zeros(1)) is True
    y.add_(1.0)
    with pytest.raises(RuntimeError, match="written in place"):
        strict.flush()


# ---------------------------------------------------------------------------------------- ADVICE round 5, #3 / #4
def test_checkpoint_renaming_is_all_or_nothing():
    class WeightRenaming:  # (named like transformers' class: the rules are read by type name)
        def __init__(self, ok):
            self.ok = ok
            self.source_patterns, self.target_patterns = ["^a\\."], ["b."]

        def reverse_transform(self):
            if not self.ok:
                raise ValueError("cannot be reversed")
            return types.SimpleNamespace(source_patterns=["^b\\."], target_patterns=["a."], scope_prefix=None)

    class WeightConverter:
        source_patterns, target_patterns = ["qkv.weight"], ["q.weight", "k.weight", "v.weight"]

    good = types.SimpleNamespace(_weight_conversions=[WeightRenaming(True)])
    assert len(ex._checkpoint_rename_rules(good)) == 1
    for bad in ([WeightRenaming(True), WeightRenaming(False)], [WeightRenaming(True), WeightConverter()]):
        with pytest.warns(UserWarning, match="keeps the module names"):
            assert ex._checkpoint_rename_rules(types.SimpleNamespace(_weight_conversions=bad)) == []


def test_expert_anchors_are_looked_up_not_searched():
    """_in_module_tree_order on 64 per-expert projections: one pass over the model's keys, then dictionary look-ups."""
    class Experts(nn.Module):
        def __init__(self):
            super().__init__()
            self.gate_up_proj, self.down_proj = nn.Parameter(torch.zeros(2, 4, 4)), nn.Parameter(torch.zeros(2, 4, 4))

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.norm, self.experts = nn.LayerNorm(4), Experts()

    model = nn.Sequential(Block(), Block())
    keys = [f"{b}.experts.{e}.{p}.weight" for b in (1, 0) for e in (1, 0) for p in ("down_proj", "gate_proj", "up_proj")]
    keys += ["1.norm.weight", "0.norm.weight"]
    got = list(ex._in_module_tree_order({k: 0 for k in keys}, model))
    assert got == ["0.norm.weight"] + [f"0.experts.{e}.{p}.weight" for e in (0, 1) for p in ("gate_proj", "up_proj", "down_proj")] + \
        ["1.norm.weight"] + [f"1.experts.{e}.{p}.weight" for e in (0, 1) for p in ("gate_proj", "up_proj", "down_proj")]


# ---------------------------------------------------------------------------------------- ADVICE round 5, #5
def test_the_automatic_activation_store_stops_at_its_cap_and_the_search_is_a_real_pass(hostmem, monkeypatch):
    """awq_lite's automatic input store (store_activations="auto") is capped at half of what the HBM budget has left: beyond it the
    stores are dropped and the exact pass runs forward_loop again -- same alphas and scales as with room for everything."""
    def run(budget_bytes):
        monkeypatch.setattr(mc._WeightCacheBudget, "host_bytes", budget_bytes)
        m, batches = _llama(torch.float32), _batches()
        calls = []

        def loop(mm):
            calls.append(1)
            with torch.no_grad():
                for b in batches:
                    mm(b)

        cfg = copy.deepcopy(mq.INT4_AWQ_CFG)
        cfg["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "search": "auto", "layer_local": False, "tie_margin": float("inf")}
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            moa.quantize(m, cfg, loop)
        return len(calls), {n: (float(mod.awq_lite.best_alpha), mod.input_quantizer._pre_quant_scale.clone())
                            for n, mod in m.named_modules() if hasattr(mod, "awq_lite")}

    roomy_passes, roomy = run(1 << 30)
    tight_passes, tight = run(1 << 20)  # 1 MiB: Gram matrices fit, half of the rest does not hold the inputs of every linear
    assert roomy_passes == 1 and tight_passes == 2, (roomy_passes, tight_passes)
    assert sorted(roomy) == sorted(tight)
    for n in roomy:
        assert roomy[n][0] == tight[n][0] and torch.equal(roomy[n][1], tight[n][1]), n


def test_histogram_mse_threshold_search_is_bounded_however_far_the_histogram_grew(monkeypatch):
    """`HistogramCalibrator.compute_amax("mse")` on integer formats evaluates the reference's loop (calib/histogram.py:286-323)
    as [candidates, bins] passes.  A calibrator whose range grew 800-fold holds ~8e5 bins: the one-pass form asked the host for
    2.7 TB and took a GPU box down (tools/calib_fuzz.py, seed 6 case 27).  Now: chunks of 2^24 elements, every candidate up to
    2^30 candidate x bin products, beyond that the candidates within 1e-4 of the float64 parabola's minimum -- and that
    screened form picks what the exhaustive one picks."""
    from model_optimizer_amd import calib

    g = torch.Generator().manual_seed(5)
    for bins, unsigned, num_bits in [(700, False, 8), (5000, False, 4), (5000, True, 8), (2048, False, 8)]:
        x = torch.randn(100000, generator=g).abs() * 3
        top = float(x.max())
        edges = torch.linspace(0, top, bins + 1)
        counts = torch.histc(x, bins=bins, min=0, max=top).to(torch.int64)
        counts[::5] = 0
        monkeypatch.setattr(calib, "_MSE_SEARCH_BUDGET", 1 << 40)
        monkeypatch.setattr(calib, "_MSE_SEARCH_CHUNK", 1 << 40)
        one_pass = calib._compute_amax_mse(counts, edges, num_bits, unsigned)           # the former form: everything at once
        monkeypatch.setattr(calib, "_MSE_SEARCH_CHUNK", 1 << 14)
        chunked = calib._compute_amax_mse(counts, edges, num_bits, unsigned)
        monkeypatch.setattr(calib, "_MSE_SEARCH_BUDGET", 1 << 16)
        screened = calib._compute_amax_mse(counts, edges, num_bits, unsigned)
        assert torch.equal(one_pass, chunked) and torch.equal(one_pass, screened), (bins, unsigned, num_bits)
    # a grown histogram: 3e5 bins = 9e10 products -- seconds and a bounded footprint, not 360 GB
    monkeypatch.undo()
    bins = 300000
    x = torch.cat([torch.randn(20000, generator=g).abs() * 0.05, torch.randn(20000, generator=g).abs() * 40])
    top = float(x.max())
    amax = calib._compute_amax_mse(torch.histc(x, bins=bins, min=0, max=top).to(torch.int64), torch.linspace(0, top, bins + 1), 8, False)
    assert 0 < float(amax) <= top
