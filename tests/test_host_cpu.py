"""CPU tests of host-side logic that needs no kernel: histogram threshold searches against the reference's
own results, block-quant bookkeeping, config plumbing, and (build container only) the modelopt seams."""

import os
import sys

import numpy as np
import pytest
import torch

import _moa_import
from conftest import DT, GOLDEN

moa = _moa_import.load()
from model_optimizer_amd import QuantizerAttributeConfig, TensorQuantizer, calib, model_quant  # noqa: E402
from oracle import oracle  # noqa: E402


def test_histogram_threshold_searches_match_reference(golden):
    """percentile / entropy reductions (calib/histogram.py:210-343) on the reference's own histograms."""
    g = golden("hist")
    for k, c in g.cases.items():
        hist = g.t(f"{k}_h2").numpy().astype(np.int64)
        edges = g.t(f"{k}_e2").numpy()
        pct = calib._compute_amax_percentile(hist, edges, 99.9)
        ent = calib._compute_amax_entropy(hist, edges, 8, False, 1, 64)
        assert torch.equal(pct.reshape(1), g.t(f"{k}_pct")), f"percentile {k}"
        assert torch.equal(ent.reshape(1), g.t(f"{k}_ent")), f"entropy {k}"
        # "mse" as the reference computes it (bit width in the bias slot, calib._compute_amax_mse): signed, unsigned (all
        # NaN: the first candidate), and scaled edges that reach the non-degenerate branch
        h_t, e_t = torch.from_numpy(hist), torch.from_numpy(edges)
        for tag, args in (("mse", (h_t, e_t, 8, False, 1, 64)), ("mseu", (h_t, e_t, 8, True, 1, 64)),
                          ("mse100", (h_t, e_t * 100, 8, False, 1, 16)), ("mse4s3", (h_t, e_t * 37, 4, False, 3, 16))):
            assert torch.equal(calib._compute_amax_mse(*args).reshape(1), g.t(f"{k}_{tag}")), f"{tag} {k}"
    with pytest.raises(ValueError):
        calib._compute_amax_percentile(hist, edges, 101)


def _entropy_divergences_textbook(hist, num_bits, unsigned, stride, start_bin):
    """The reference's loop as written (digitize / add.at / Counter), returning every divergence."""
    from collections import Counter

    from scipy.stats import entropy
    bins = hist.astype(np.int64).copy()
    bins[0] = bins[1]
    nbins = 1 << (num_bits - 1 + int(unsigned))
    out = []
    for i in range(start_bin, len(bins) + 1, stride):
        space = np.linspace(0, i, num=nbins + 1)
        dig = np.digitize(range(i), space) - 1
        dig[bins[:i] == 0] = -1
        valid = dig != -1
        ndc = np.zeros(nbins, dtype=np.float64)
        np.add.at(ndc, dig[valid], bins[:i][valid])
        for key, val in Counter(dig.tolist()).items():
            if key != -1:
                ndc[key] = ndc[key] / val
        new_density = np.zeros(i, dtype=np.float64)
        new_density[valid] = ndc[dig[valid]]
        ref = np.array(bins[:i], dtype=np.float64)
        ref[-1] += np.sum(bins[i:])
        out.append(entropy(ref, new_density))
    return np.array(out)


@pytest.mark.parametrize("dense", [False, True], ids=["with_empty_bins", "no_empty_bins"])
@pytest.mark.parametrize("num_bits,unsigned,nb,stride,start", [(8, False, 2048, 1, 128), (8, True, 2048, 3, 128),
                                                                 (4, False, 777, 1, 16), (6, False, 1500, 7, 100)])
def test_entropy_search_is_the_reference_loop_bit_for_bit(num_bits, unsigned, nb, stride, start, dense):
    """calib._compute_amax_entropy replaces digitize / add.at / Counter by exact integer equivalents: every
    divergence must equal the textbook loop's bit for bit (zeros, gaps and a heavy tail included)."""
    rng = np.random.default_rng(nb + num_bits)
    hist = (rng.exponential(1.0, nb) * 1e6 * np.exp(-np.arange(nb) / (nb / 6))).astype(np.int64)
    if dense:
        hist += 1                                     # every bin occupied: the search's gather-free branch
    else:
        hist[rng.integers(0, nb, nb // 10)] = 0      # empty bins
        hist[nb // 2: nb // 2 + 40] = 0               # a gap wider than a bucket
    hist[-1] = 12345
    got = []
    edges = np.linspace(0, 1.0, nb + 1, dtype=np.float32)
    amax = calib._compute_amax_entropy(hist, edges, num_bits, unsigned, stride, start, divergences_out=got)
    want = _entropy_divergences_textbook(hist, num_bits, unsigned, stride, start)
    assert np.array_equal(np.array(got).view(np.uint64), want.view(np.uint64))
    last_argmin = len(want) - 1 - np.argmin(want[::-1])
    assert amax.item() == edges[last_argmin * stride + start].item()


def test_oracle_awq_weight_scale_matches_reference(golden):
    g = golden("awq")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        got = oracle.awq_weight_scale(g.t(f"{k}_w", dt), c["g"])
        want = g.t(f"{k}_wscale")
        ulp = want.abs() * (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10)
        assert ((got - want).abs() <= ulp).all(), f"oracle awq_weight_scale {k}"


def test_tensor_quantizer_config_and_state_without_gpu():
    q = TensorQuantizer(QuantizerAttributeConfig(num_bits=4, block_sizes={-1: 128}))
    assert q.is_static_block_quant and not q._dynamic and q.maxbound == 7.0 and q.amax is None
    q.amax = torch.tensor([[1.0], [2.0]])
    with pytest.raises(RuntimeError, match="Changing shape"):
        q.amax = torch.tensor(1.0)
    q.reset_amax()
    assert q.amax is None
    with pytest.raises(RuntimeError, match="Calibrator returned None"):
        q.load_calib_amax()
    mx = TensorQuantizer(QuantizerAttributeConfig(num_bits=(2, 1), block_sizes={-1: 32, "type": "dynamic", "scale_bits": (8, 0)}))
    assert mx.is_mx_format and mx._block_dynamic and not mx._dynamic and not mx.is_static_block_quant
    q.pre_quant_scale = torch.ones(4)
    assert q.pre_quant_scale is not None
    q._enable_pre_quant_scale = False
    assert q.pre_quant_scale is None
    # block bookkeeping: padding / slices exactly like tensor_quantizer.py:975-1016 (no kernel involved)
    x = torch.zeros(3, 200)
    q2 = TensorQuantizer(QuantizerAttributeConfig(num_bits=4, block_sizes={-1: 128}))
    q2._setup_for_blockquant(x)
    v = q2._process_for_blockquant(x)
    assert v.shape == (6, 128) and q2.axis == (0,) and q2._amax_shape_for_export == (3, -1)
    assert q2._reset_to_original_shape(v).shape == (3, 200)
    with pytest.raises(ValueError, match="shape has changed"):
        q2._process_for_blockquant(torch.zeros(3, 100))
    # blocks on both axes: ragged shapes are zero-padded to whole tiles and cut back (tensor_quantizer.py:1018-1043)
    q3 = TensorQuantizer(QuantizerAttributeConfig(num_bits=4, block_sizes={-1: 16, -2: 16}))
    q3._setup_for_blockquant(x)
    v3 = q3._process_for_blockquant(x)
    assert v3.shape == (1, 16, 13, 16) and q3.axis == (0, 2) and q3._reset_to_original_shape(v3).shape == (3, 200)


def test_quantize_config_plumbing_without_gpu():
    model = torch.nn.Sequential(torch.nn.Linear(128, 128), torch.nn.GELU(), torch.nn.Linear(128, 128))
    model_quant.replace_quant_module(model)
    model_quant.set_quantizer_by_cfg(model, model_quant.INT4_AWQ_CFG["quant_cfg"])
    lin = model[0]
    assert lin.weight_quantizer.block_sizes == {-1: 128, "type": "static"} and lin.weight_quantizer.num_bits == 4
    assert not lin.input_quantizer.is_enabled and lin.weight_quantizer.is_enabled
    # the product path has no CPU fallback: running it on CPU tensors must fail loudly
    with pytest.raises(moa.MoquantError, match="must live on the GPU"):
        model(torch.randn(2, 128))
    with pytest.raises(ValueError, match="outside this path"):
        moa.quantize(torch.nn.Linear(4, 4), {"quant_cfg": {}, "algorithm": "svdquant"})


def test_modelopt_seams_install():
    sys.path.insert(0, GOLDEN)
    import ref_shim

    if not ref_shim.reference_available():
        pytest.skip("reference checkout not present (GPU box)")
    ref_shim.install()
    installed = moa.modelopt_plugin.install()
    assert installed == ["S1:extensions", "S3:backend=mi355x", "S6:reduce_amax", "S5:create_asp_mask",
                         "S5:create_sgpt_mask"]
    import modelopt.torch.quantization.extensions as ext
    from modelopt.torch.quantization.nn.modules import tensor_quantizer as ref_tq
    from modelopt.torch.quantization.utils import core_utils

    for fn, names in [(ext.get_cuda_ext, ["fake_tensor_quant", "fake_tensor_quant_", "fake_tensor_quant_with_axis",
                                          "INT4_quantize", "INT4_dequantize", "NF4_quantize", "NF4_dequantize"]),
                      (ext.get_cuda_ext_fp8, ["fake_e4m3fy", "fake_e4m3fy_with_axis"]),
                      (ext.get_cuda_ext_mx, ["fused_amax_convert", "convert_to_exmy", "Types"])]:
        for n in names:
            assert hasattr(fn(), n), f"adapter lacks {n}"
    assert ext.get_cuda_ext_mx().Types.E2M1 == 6 and ext.get_cuda_ext_mx().Types.E8M0 == 9  # tensor_quant_mx.h:39
    assert ref_tq.is_registered_quant_backend("mi355x")
    # CPU tensors keep flowing through the reference's own eager code (the seams only take GPU tensors)
    x = torch.randn(4, 8)
    assert torch.equal(core_utils.reduce_amax(x), x.abs().max())
    import modelopt.torch.quantization as mtq

    m = torch.nn.Sequential(torch.nn.Linear(16, 16))
    mtq.quantize(m, mtq.INT8_DEFAULT_CFG, lambda mod: mod(torch.randn(2, 16)))
    assert m[0].weight_quantizer.amax is not None
    # S2 (opt-in): the module-level operators are re-pointed; CPU tensors still reach the reference's implementation
    from modelopt.torch.quantization import tensor_quant as ref_tensor_quant
    assert "S2:library_ops" in moa.modelopt_plugin.install(library_ops=True)
    assert getattr(ref_tensor_quant.quantize_op, "_moq_seam", False)
    # a CPU tensor still takes the reference's own operator -- which, with an extension module present (S1), hands it
    # to the extension and gets the extension's "must be a GPU tensor" RuntimeError, exactly as with the CUDA build
    with pytest.raises(RuntimeError, match="GPU"):
        ref_tensor_quant.quantize_op(x, x.abs().max(), 8, 0, False, True)
    m2 = torch.nn.Sequential(torch.nn.Linear(16, 16))
    mtq.quantize(m2, mtq.INT8_DEFAULT_CFG, lambda mod: mod(torch.randn(2, 16)))
    assert torch.isfinite(m2(torch.randn(2, 16))).all()
    # SparseGPT seams: CPU tensors keep running the reference's own code through the re-pointed functions
    from modelopt.torch.sparsity.weight_sparsity import sparsegpt

    lin = type("Linear", (), {})()
    lin.hessian, lin.samples = torch.zeros(16, 16), 0
    sparsegpt.SparseGPTSearcher._hook_compute_hessian(lin, (torch.randn(1, 8, 16),), None)
    assert lin.samples == 1 and lin.hessian.abs().sum() > 0
    mask = sparsegpt.create_sgpt_mask(torch.randn(8, 16), lin.hessian + torch.eye(16),
                                      {"pattern": "2:4 sparsity", "col_block_size": 128, "row_block_size": -1,
                                       "hessian_damp": 0.1})
    assert mask.dtype == torch.bool and (mask.view(8, -1, 4).sum(-1) <= 2).all()


def test_hf_attention_registration_patches_and_restores_the_interface():
    """hf_attention (plugins/huggingface.py:283-334 mirror) without any kernel: with all bmm quantizers disabled the
    converted model computes exactly what the original did, the attention interface is restored after every forward
    -- also when the forward raises -- and only modules that call the interface are converted."""
    from transformers import LlamaConfig, LlamaForCausalLM
    from transformers.models.llama import modeling_llama

    from model_optimizer_amd import hf_attention, nn as mnn
    cfgd = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, vocab_size=64, max_position_embeddings=32)
    tokens = torch.randint(0, 64, (2, 16), generator=torch.Generator().manual_seed(0))
    for impl in ("sdpa", "eager"):
        cfg = LlamaConfig(**cfgd)
        cfg._attn_implementation = impl
        torch.manual_seed(0)
        model = LlamaForCausalLM(cfg).eval()
        with torch.no_grad():
            want = model(tokens).logits
        eager_fn, sdpa_fn = modeling_llama.eager_attention_forward, modeling_llama.ALL_ATTENTION_FUNCTIONS["sdpa"]
        assert hf_attention.register_hf_attentions_on_the_fly(model) == 2
        assert hf_attention.register_hf_attentions_on_the_fly(model) == 0  # idempotent
        attn = model.model.layers[0].self_attn
        assert type(attn).__name__ == "QuantLlamaAttention" and isinstance(attn, modeling_llama.LlamaAttention)
        for name in ("q_bmm_quantizer", "k_bmm_quantizer", "v_bmm_quantizer"):
            assert not getattr(attn, name).is_enabled
        seen = []
        attn.k_bmm_quantizer.register_forward_hook(lambda m, i, o: seen.append(tuple(i[0].shape)))
        with torch.no_grad():
            got = model(tokens).logits
        assert torch.equal(got, want)
        assert seen == [(2, 2, 16, 16)]  # [B, kv_heads, S, head_dim]: the key states reach the quantizer
        assert modeling_llama.eager_attention_forward is eager_fn
        assert modeling_llama.ALL_ATTENTION_FUNCTIONS["sdpa"] is sdpa_fn
        with pytest.raises(Exception):
            model(torch.full((1, 4), 10 ** 6))  # out-of-range token: the forward raises inside the model
        assert modeling_llama.eager_attention_forward is eager_fn
        assert modeling_llama.ALL_ATTENTION_FUNCTIONS["sdpa"] is sdpa_fn
        # set_quantizer_by_cfg reaches the new quantizers by wildcard; the KV preset enables k and v only
        mq = moa.model_quant
        mnn.replace_quant_module(model)
        cfg2 = mq.update_quant_cfg_with_kv_cache_quant(mq.FP8_DEFAULT_CFG, mq.FP8_KV_CFG["quant_cfg"])
        assert cfg2 is not mq.FP8_DEFAULT_CFG and "*[kv]_bmm_quantizer" not in mq.FP8_DEFAULT_CFG["quant_cfg"]
        mq.set_quantizer_by_cfg(model, cfg2["quant_cfg"])
        assert attn.k_bmm_quantizer.is_enabled and attn.v_bmm_quantizer.is_enabled
        assert not attn.q_bmm_quantizer.is_enabled and not attn.p_bmm_quantizer.is_enabled
        assert tuple(attn.k_bmm_quantizer._num_bits) == (4, 3)
        assert moa.export.get_kv_cache_format(model) == "FP8"
    assert hf_attention.register_hf_attentions_on_the_fly(torch.nn.Linear(2, 2)) == 0


def test_fused_experts_registration_and_name_matching():
    """hf_experts (plugins/huggingface.py:976-1142, conversion.py:317-341 mirror) without any kernel: per-expert
    quantizer layout, wildcard matching through the ModuleList index, F.linear restored after the forward, outputs
    unchanged while every quantizer is disabled."""
    import torch.nn.functional as F
    from transformers import MixtralConfig, MixtralForCausalLM

    from model_optimizer_amd import hf_experts, nn as mnn
    cfg = MixtralConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=4,
                        num_key_value_heads=2, vocab_size=64, max_position_embeddings=32, num_local_experts=4,
                        num_experts_per_tok=2)
    torch.manual_seed(0)
    model = MixtralForCausalLM(cfg).eval()
    tokens = torch.randint(0, 64, (2, 16), generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        want = model(tokens).logits
    linear_fn = F.linear
    mnn.replace_quant_module(model)
    ex = model.model.layers[0].mlp.experts
    assert hf_experts.is_quant_fused_experts(ex) and type(ex).__name__ == "QuantMixtralExperts"
    assert len(ex.gate_up_proj_weight_quantizers) == 4 and len(ex.down_proj_weight_quantizers) == 4
    mq = moa.model_quant
    assert mq._normalize_fused_experts_quantizer_name("a.experts.gate_up_proj_weight_quantizers.3") == \
        "a.experts.gate_up_proj_weight_quantizer"
    assert mq._normalize_fused_experts_quantizer_name("a.weight_quantizer.0") == "a.weight_quantizer"
    assert mq._normalize_fused_experts_quantizer_name("a.input_quantizer") == "a.input_quantizer"
    mq.set_quantizer_by_cfg(model, {"*weight_quantizer": {"num_bits": (4, 3), "axis": None},
                                    "*input_quantizer": {"num_bits": (4, 3), "axis": None},
                                    "*experts.down_proj_weight_quantizers.2": {"enable": False}})
    assert tuple(ex.gate_up_proj_weight_quantizers[3]._num_bits) == (4, 3)
    assert tuple(ex.gate_up_proj_input_quantizer._num_bits) == (4, 3)
    assert ex.down_proj_weight_quantizers[1].is_enabled and not ex.down_proj_weight_quantizers[2].is_enabled
    for q in [m for m in model.modules() if isinstance(m, TensorQuantizer)]:
        q.disable()
    seen = []
    for i, q in enumerate(ex.gate_up_proj_weight_quantizers):
        q.register_forward_hook(lambda m, inp, out, i=i: seen.append((i, inp[0].data_ptr() == ex.gate_up_proj[i].data_ptr())))
    with torch.no_grad():
        got = model(tokens).logits
    assert torch.equal(got, want) and F.linear is linear_fn
    assert seen and all(ok for _, ok in seen)  # expert i's slice reached expert i's quantizer
    pairs = list(ex.iter_weights_for_calibration())
    assert len(pairs) == 8 and pairs[5][0].data_ptr() == ex.down_proj[1].data_ptr() and pairs[5][1] is ex.down_proj_weight_quantizers[1]
    # checkpoint key names of the exported per-expert tensors
    ren = moa.export.rename_to_checkpoint_keys({"model.layers.0.mlp.experts.2.up_proj.weight": 1,
                                                "model.layers.0.mlp.gate.weight": 2,
                                                "model.layers.0.self_attn.q_proj.weight": 3}, model)
    assert set(ren) == {"model.layers.0.block_sparse_moe.experts.2.w3.weight", "model.layers.0.block_sparse_moe.gate.weight",
                        "model.layers.0.self_attn.q_proj.weight"}


def test_grouped_quantizer_container():
    """GroupedQuantizer (tensor_quantizer.py:1865-1893): members act on different tensors, broadcasts return lists,
    property reads come from the first member; set_quantizer_by_cfg reaches the members through the index."""
    from model_optimizer_amd.tensor_quantizer import GroupedQuantizer, SequentialQuantizer
    qs = [TensorQuantizer(QuantizerAttributeConfig(num_bits=8, axis=None)) for _ in range(3)]
    g = GroupedQuantizer(*qs)
    assert len(g) == 3 and g[1] is qs[1] and g.is_enabled and g.amax is None
    assert g.disable() == [None, None, None] and not any(q.is_enabled for q in qs)
    g.enable()
    qs[0].amax = torch.tensor(2.0)
    assert g.amax.item() == 2.0 and qs[1].amax is None
    g.reset_amax()
    assert g.amax is None
    with pytest.raises(AssertionError):
        GroupedQuantizer(torch.nn.Identity())
    m = torch.nn.Module()
    m.weight_quantizer = g
    model_quant.set_quantizer_by_cfg(m, {"*weight_quantizer": {"num_bits": (4, 3), "axis": None}})
    assert all(tuple(q._num_bits) == (4, 3) for q in g) and isinstance(m.weight_quantizer, GroupedQuantizer)
    assert isinstance(GroupedQuantizer(SequentialQuantizer(TensorQuantizer(), TensorQuantizer()))[0], SequentialQuantizer)


def test_library_ops_schema_and_fake_implementations():
    """moquant::quantize_op / moquant::dynamic_block_quantize_op (S2 mirror): defined once, schemas as the reference's
    tensorrt:: operators, fake implementations give shape / dtype without touching a kernel (meta tensors)."""
    from model_optimizer_amd import library_ops as lo
    assert lo.define() and lo.define()
    s1 = torch.ops.moquant.quantize_op.default._schema
    assert [a.name for a in s1.arguments] == ["input", "amax", "num_bits", "exponent_bits", "unsigned", "narrow_range"]
    s2 = torch.ops.moquant.dynamic_block_quantize_op.default._schema
    assert [a.name for a in s2.arguments] == ["input", "block_size", "amax", "num_bits", "exponent_bits", "scale_num_bits",
                                              "scale_exponent_bits"]
    x = torch.empty(4, 64, dtype=torch.bfloat16, device="meta")
    y = torch.ops.moquant.quantize_op(x, torch.empty((), device="meta"), 8, 4, False, False)
    assert y.shape == x.shape and y.dtype == x.dtype and y.device.type == "meta"
    y = torch.ops.moquant.dynamic_block_quantize_op(x, 32, None, 4, 2, 9, 8)
    assert y.shape == x.shape and y.dtype == x.dtype
    assert lo._formats(4, 2, 9, 8) == ((2, 1), (8, 0)) and lo._formats(4, 2, 8, 4) == ((2, 1), (4, 3))  # E + M + 1 bits
    with pytest.raises(NotImplementedError):
        torch.ops.moquant.quantize_op(torch.zeros(4), torch.ones(()), 8, 0, False, True)  # no CPU implementation


def test_host_helpers_agree_with_the_reference_functions():
    """Pure-host helpers (no kernel) against the reference's own functions, called directly on CPU (build container only):
    quantizer-name normalisation, block padding, reduce-axis conversion, percentile / entropy threshold searches, the
    KV-cache key post-processing of the checkpoint export."""
    sys.path.insert(0, GOLDEN)
    import ref_shim

    if not ref_shim.reference_available():
        pytest.skip("reference checkout not present (GPU box)")
    ref_shim.install()
    from modelopt.torch.export import quant_utils as ref_export
    from modelopt.torch.quantization import conversion as ref_conv
    from modelopt.torch.quantization.calib import histogram as ref_hist
    from modelopt.torch.quantization.utils import core_utils as ref_core

    from model_optimizer_amd import export as our_export
    from model_optimizer_amd import ops as our_ops

    names = ["model.layers.0.mlp.experts.gate_up_proj_weight_quantizers.3", "a.b.down_proj_input_quantizer",
             "x.weight_quantizer.0", "x.weight_quantizer.12.inner", "y.input_quantizers.7", "z.weight_quantizers",
             "w.k_bmm_quantizer", "experts.10.w1.weight_quantizer", "q.weight_quantizers.0.1"]
    for n in names:
        assert model_quant._normalize_fused_experts_quantizer_name(n) == ref_conv._normalize_fused_experts_quantizer_name(n), n
    gen = torch.Generator().manual_seed(0)
    for shape, blocks in [((5, 7), {-1: 4, -2: 2}), ((3, 10, 33), {-1: 16}), ((8, 8), {-1: 4, -2: 4}), ((2, 3, 5, 9), {1: 2, -1: 4})]:
        x = torch.randn(*shape, generator=gen)
        assert torch.equal(our_ops.reduce_block_padding(x, blocks), ref_core.reduce_block_padding(x, blocks))
        assert torch.equal(our_ops.reduce_block_padding(x, blocks, 1.5), ref_core.reduce_block_padding(x, blocks, 1.5))
    for nd in (1, 2, 4):
        x = torch.zeros(*([2] * nd))
        for axis in [None, 0, -1, (0,), tuple(range(nd))]:
            got = calib.convert_quantization_axis_to_reduce_axis(x, axis)
            assert got == ref_core.convert_quantization_axis_to_reduce_axis(x, axis), (nd, axis)
    rng = np.random.default_rng(5)
    for nb in (512, 2048):
        hist = (rng.exponential(1.0, nb) * 1e5 * np.exp(-np.arange(nb) / (nb / 5))).astype(np.int64)
        hist[rng.integers(0, nb, nb // 20)] = 0
        edges = np.linspace(0, 3.0, nb + 1, dtype=np.float32)
        for pct in (99.0, 99.99, 50.0):
            assert torch.equal(calib._compute_amax_percentile(hist, edges, pct), ref_hist._compute_amax_percentile(hist, edges, pct))
        assert torch.equal(calib._compute_amax_entropy(hist, edges, 8, False, 4, 128),
                           ref_hist._compute_amax_entropy(hist, edges, 8, False, 4, 128))
        assert torch.equal(calib._compute_amax_entropy(hist, edges, 8, True, 16, 64),
                           ref_hist._compute_amax_entropy(hist, edges, 8, True, 16, 64))
    keys = {"model.layers.0.self_attn.k_bmm_quantizer._amax": torch.tensor(3.5),
            "model.layers.0.self_attn.v_bmm_quantizer._amax": torch.tensor(100.0),
            "model.layers.0.self_attn.q_bmm_quantizer._amax": torch.tensor(1.0),
            "model.layers.0.self_attn.o_proj.output_quantizer._amax": torch.tensor(1.0),
            "model.layers.0.input_layernorm.weight": torch.ones(4),
            "model.layers.0.mlp.gate.weight": torch.ones(2, 2)}
    for k, v in keys.items():
        ours = our_export._postprocess_kv_key(k, v, "FP8")
        ref = ref_export._postprocess_single_tensor(k, v, 448.0, "FP8")
        assert ours[0] == ref[0], k
        if ref[0] is not None:
            assert torch.equal(ours[1], ref[1]), k


def test_export_refuses_formats_it_does_not_pack():
    """get_quantization_format: INT4 -> FP8 sequential weight quantizers are W4A8_AWQ, any other chain raises instead of
    exporting its first stage only; disabled weight quantizers export the plain weight; NVFP4-layout blocks are refused."""
    from model_optimizer_amd import export, nn as mnn
    m = torch.nn.Sequential(torch.nn.Linear(128, 64, bias=False))
    mnn.replace_quant_module(m)
    model_quant.set_quantizer_by_cfg(m, model_quant.W4A8_MAX_CFG["quant_cfg"])
    assert export.get_quantization_format(m[0]) == "w4a8_awq"
    model_quant.set_quantizer_by_cfg(m, {"*weight_quantizer": [{"num_bits": 8, "axis": 0}, {"num_bits": (4, 3), "axis": None}]})
    with pytest.raises(NotImplementedError, match="SequentialQuantizer"):
        export.get_quantization_format(m[0])
    model_quant.set_quantizer_by_cfg(m, {"*weight_quantizer": {"enable": False}})
    assert export.get_quantization_format(m[0]) is export.QUANTIZATION_NONE
    m2 = torch.nn.Sequential(torch.nn.Linear(128, 64, bias=False))
    mnn.replace_quant_module(m2)
    model_quant.set_quantizer_by_cfg(m2, model_quant.NVFP4_DEFAULT_CFG["quant_cfg"])
    with pytest.raises(NotImplementedError):
        export.get_quantization_format(m2[0])
    for cfg, want in [(model_quant.INT8_DEFAULT_CFG, "int8_sq"), (model_quant.FP8_DEFAULT_CFG, "fp8"),
                      (model_quant.INT4_AWQ_CFG, "int4_awq"), (model_quant.MXFP4_DEFAULT_CFG, "mxfp4"),
                      (model_quant.FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG, "fp8_pb_wo")]:
        m3 = torch.nn.Sequential(torch.nn.Linear(128, 64, bias=False))
        mnn.replace_quant_module(m3)
        model_quant.set_quantizer_by_cfg(m3, cfg["quant_cfg"])
        assert export.get_quantization_format(m3[0]) == want, want


def test_quant_cfg_list_form_default_key_and_parent_class():
    """set_quantizer_by_cfg takes the reference's canonical ordered-list quant_cfg (config.py:1447-1569,
    conversion.py:245-314): entries with cfg replace the attributes and enable, enable-only entries toggle, the legacy
    "default" key means "*", parent_class restricts an entry to quantizers of that module class."""
    moa = _moa_import.load()
    mq = moa.model_quant

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = torch.nn.Linear(16, 16)
            self.norm = torch.nn.LayerNorm(16)
            self.lm_head = torch.nn.Linear(16, 8)

    def build(cfg):
        net = Net()
        moa.nn.replace_quant_module(net)
        mq.set_quantizer_by_cfg(net, cfg)
        return net

    listed = build([{"quantizer_name": "*", "enable": False},
                    {"quantizer_name": "*weight_quantizer", "cfg": {"num_bits": 4, "block_sizes": {-1: 8, "type": "static"}}},
                    {"quantizer_name": "*input_quantizer", "cfg": {"num_bits": (4, 3), "axis": None}},
                    {"parent_class": "nn.LayerNorm", "quantizer_name": "*", "enable": False},
                    {"quantizer_name": "*lm_head*", "enable": False}])
    assert listed.fc.weight_quantizer.is_enabled and listed.fc.weight_quantizer.num_bits == 4
    assert listed.fc.input_quantizer.is_enabled and tuple(listed.fc.input_quantizer.num_bits) == (4, 3)
    assert not listed.fc.output_quantizer.is_enabled
    assert not listed.norm.input_quantizer.is_enabled                      # parent_class entry
    assert tuple(listed.norm.input_quantizer.num_bits) == (4, 3)           # ... which only toggled `enable`
    assert not listed.lm_head.weight_quantizer.is_enabled and not listed.lm_head.input_quantizer.is_enabled
    legacy = build({"default": {"enable": False}, "*weight_quantizer": {"num_bits": 8, "axis": 0},
                    "nn.LayerNorm": {"*input_quantizer": {"num_bits": 8, "axis": None}}})
    assert legacy.fc.weight_quantizer.is_enabled and not legacy.fc.input_quantizer.is_enabled  # "default" == "*"
    assert legacy.norm.input_quantizer.is_enabled and not legacy.lm_head.input_quantizer.is_enabled
    with pytest.raises(ValueError):
        mq.normalize_quant_cfg_list([{"quantizer_name": "*"}])              # neither cfg nor enable
    with pytest.raises(ValueError):
        build([{"parent_class": "nn.NoSuchLayer", "quantizer_name": "*", "enable": False}])
    # every preset ends with the reference's default-disabled patterns (units/default_disabled_quantizers.yaml)
    for preset in (mq.FP8_DEFAULT_CFG, mq.INT4_AWQ_CFG, mq.INT8_SMOOTHQUANT_CFG, mq.MXFP4_DEFAULT_CFG, mq.W4A8_MAX_CFG):
        tail = list(preset["quant_cfg"].items())[-len(mq.DEFAULT_DISABLED_QUANTIZERS):]
        assert [k for k, _ in tail] == list(mq.DEFAULT_DISABLED_QUANTIZERS) and all(v == {"enable": False} for _, v in tail)


def test_presets_keep_moe_gates_and_routers_in_high_precision():
    """A Qwen2-MoE style block: `mlp.gate` (the router), `mlp.shared_expert_gate` and `lm_head` are nn.Linear modules the
    reference's presets leave unquantized (default_disabled_quantizers.yaml)."""
    moa = _moa_import.load()

    class MLP(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.gate = torch.nn.Linear(16, 4, bias=False)
            self.shared_expert_gate = torch.nn.Linear(16, 1, bias=False)
            self.up_proj = torch.nn.Linear(16, 32, bias=False)

    class Layer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.mlp = MLP()

    class Net(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = torch.nn.ModuleList([Layer()])
            self.router = torch.nn.Linear(16, 4)
            self.lm_head = torch.nn.Linear(16, 8)

    net = Net()
    moa.nn.replace_quant_module(net)
    moa.model_quant.set_quantizer_by_cfg(net, moa.model_quant.FP8_DEFAULT_CFG["quant_cfg"])
    mlp = net.layers[0].mlp
    assert mlp.up_proj.weight_quantizer.is_enabled and mlp.up_proj.input_quantizer.is_enabled
    for lin in (mlp.gate, mlp.shared_expert_gate, net.router, net.lm_head):
        assert not lin.weight_quantizer.is_enabled and not lin.input_quantizer.is_enabled


def test_quantize_does_not_import_transformers():
    """quantize() of a plain nn.Module must not import transformers: the cold import pages in for 17-30 s on a fresh box and
    sat inside the timed INT4-AWQ call of round 3's driver run (profiles/r04_awq_unstaged.md).  Only a process that already
    imported transformers.modeling_utils can hold a PreTrainedModel, so the check reads sys.modules."""
    import subprocess
    import sys

    code = ("import sys, torch; sys.path.insert(0, %r); import _moa_import; moa = _moa_import.load(); "
            "from model_optimizer_amd import nn as qnn, hf_attention; "
            "m = torch.nn.Sequential(torch.nn.Linear(8, 8), torch.nn.ReLU(), torch.nn.Linear(8, 4)); "
            "qnn.replace_quant_module(m); "
            "assert isinstance(m[0], qnn.QuantLinear) and not hf_attention._is_supported_hf_model(m); "
            "assert 'transformers' not in sys.modules, 'convert imported transformers'; "
            "import transformers; "
            "cfg = transformers.LlamaConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, "
            "num_key_value_heads=2, vocab_size=50); hm = transformers.LlamaForCausalLM(cfg); "
            "assert hf_attention._is_supported_hf_model(hm) and not hf_attention._is_supported_hf_model(m); print('ok')"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]


def test_entropy_device_search_breaks_ties_with_the_reference_arithmetic():
    """calib._pick_entropy_candidate: the device's divergences decide alone when one candidate is the clear minimum; every
    candidate within 1e-9 of the minimum is re-scored by the numpy restatement of the reference's loop and the LAST minimum
    of those exact values wins (histogram.py:277-279) -- so a device value that is off in its last bits cannot move the
    choice."""
    rng = np.random.default_rng(3)
    nb, start, stride = 600, 32, 1
    hist = (rng.exponential(1.0, nb) * 1e5 * np.exp(-np.arange(nb) / 90)).astype(np.int64)
    exact = []
    calib._compute_amax_entropy(hist, np.linspace(0, 1, nb + 1, dtype=np.float32), 8, False, stride, start, divergences_out=exact)
    exact = np.array(exact)
    want = len(exact) - 1 - int(np.argmin(exact[::-1]))
    calls = []

    def hist_fn():
        calls.append(1)
        return hist

    assert calib._pick_entropy_candidate(exact.copy(), hist_fn, 8, False, stride, start) == want and not calls
    # the device is off by a few ulp and makes two OTHER candidates look minimal: the exact re-score restores the choice
    noisy = exact.copy()
    others = [want - 3, want + 2]
    noisy[others] = exact[want] * (1 - 1e-12)
    noisy[want] = exact[want] * (1 + 1e-12)
    assert calib._pick_entropy_candidate(noisy, hist_fn, 8, False, stride, start) == want and len(calls) == 1
    # a true tie: the LAST of the tied candidates (two identical divergences cannot be told apart by value)
    tied = exact.copy()
    tied[:] = 5.0
    assert calib._pick_entropy_candidate(tied, lambda: np.full(nb, 7, dtype=np.int64), 8, False, stride, start) == \
        len(tied) - 1 - int(np.argmin(np.array(_all_divergences(np.full(nb, 7, dtype=np.int64), stride, start))[::-1]))
    nan = exact.copy()
    nan[5] = np.nan
    assert calib._pick_entropy_candidate(nan, hist_fn, 8, False, stride, start) == 5  # np.argmin: a NaN is the minimum


def _all_divergences(hist, stride, start):
    out = []
    calib._compute_amax_entropy(hist, np.linspace(0, 1, len(hist) + 1, dtype=np.float32), 8, False, stride, start,
                                divergences_out=out)
    return out


def test_small_quantizer_helpers_and_the_post_calibration_warning(monkeypatch):
    """step_size / is_fp8 / is_mxfp / disable_pre_quant_scale / validate_attr (tensor_quantizer.py:396-405, :553-604,
    :753-773, :1387-1395) and the check mtq.calibrate ends with (model_quant.py:119-122): a warning per quantizer buffer
    holding a negative / inf / NaN entry -- same text as the reference's -- found by ONE flattened test per device."""
    import warnings

    import hostmem_backend

    hostmem_backend.install(monkeypatch, moa)
    TQ, Cfg = moa.TensorQuantizer, moa.QuantizerAttributeConfig
    q = TQ(Cfg(num_bits=8, axis=None))
    with pytest.warns(UserWarning, match="undefined under dynamic amax"):
        assert q.step_size is None
    q.amax = torch.tensor(12.7)
    assert torch.equal(q.step_size, torch.tensor(12.7) / 127.0)
    assert TQ(Cfg(num_bits=(4, 3), axis=None)).is_fp8 and not TQ(Cfg(num_bits=(4, 3), axis=0)).is_fp8
    mx = TQ(Cfg(num_bits=(2, 1), block_sizes={-1: 32, "type": "dynamic", "scale_bits": (8, 0)}))
    assert mx.is_mxfp(4) and not mx.is_mxfp(8) and not q.is_mxfp(4)
    with pytest.raises(NotImplementedError):
        mx.is_mxfp(5)
    q.pre_quant_scale = torch.ones(4) * 2
    with q.disable_pre_quant_scale():
        assert q.pre_quant_scale is None
    assert torch.equal(q.pre_quant_scale, torch.ones(4) * 2)
    assert q.validate_attr() and q.validate_attr(attr_name="_pre_quant_scale")
    with pytest.raises(ValueError, match="contains invalid values"):
        q.validate_attr(attr_value=torch.tensor([1.0, float("inf")]), raise_error=True)

    net = torch.nn.Sequential(torch.nn.Linear(16, 16), torch.nn.ReLU(), torch.nn.Linear(16, 8))
    with warnings.catch_warnings():
        warnings.simplefilter("error")  # a healthy model: no warning
        moa.quantize(net, moa.model_quant.INT8_DEFAULT_CFG, lambda m: m(torch.randn(4, 16)))
    net[0].input_quantizer._amax.fill_(float("nan"))
    net[2].weight_quantizer._amax.view(-1)[0] = -1.0
    with pytest.warns(UserWarning) as rec:
        moa.model_quant.calibrate(net, None)
    texts = [str(w.message) for w in rec]
    assert any(t.startswith("0.input_quantizer._amax contains invalid values: ") for t in texts), texts
    assert any(t.startswith("2.weight_quantizer._amax contains invalid values: ") for t in texts), texts
    assert len([t for t in texts if "contains invalid values" in t]) == 2


def test_kv_cache_format_names_and_the_int8_refusal(monkeypatch):
    """get_kv_cache_format names what _compute_kv_cache_dtype names (export/quant_utils.py:440-461); an INT8 KV cache stops
    the checkpoint writer with the reference's assertion (quant_utils.py:1038-1040: only FP8 / NVFP4 amax become scales)."""
    import hostmem_backend
    import transformers as tf

    hostmem_backend.install(monkeypatch, moa)
    cfg = tf.LlamaConfig(hidden_size=32, intermediate_size=64, num_hidden_layers=1, num_attention_heads=2, num_key_value_heads=2,
                         vocab_size=50, architectures=["LlamaForCausalLM"])
    torch.manual_seed(0)
    model = tf.LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    mq = moa.model_quant
    qcfg = mq.update_quant_cfg_with_kv_cache_quant(mq.FP8_DEFAULT_CFG, {"*[kv]_bmm_quantizer": {"num_bits": 8, "axis": None, "enable": True}})
    moa.quantize(model, qcfg, lambda m: m(torch.randint(0, 50, (2, 6))))
    assert moa.export.get_kv_cache_format(model) == "INT8"
    assert moa.export.hf_quant_config(model)["quantization"]["kv_cache_quant_algo"] == "INT8"
    with pytest.raises(AssertionError, match="Invalid KV cache quantization format"):
        moa.export.export_state_dict(model, torch.bfloat16, lambda: model(torch.ones([1, 2], dtype=torch.long)))


def test_deferred_layer_statistics_bookkeeping():
    """calib.DeferredAmax on host tensors (the sweep itself -- moq_mt_amax_running -- is the GPU tier's): a tensor noted by
    several calibrators is held once, every buffer that asked gets the maximum, buffers keep their running value, a tensor
    written in place before the flush is an error, the byte limit flushes by itself."""
    from model_optimizer_amd import calib

    class OnHost(calib.DeferredAmax):
        def _run(self, ents):
            for x, _, bufs in ents:
                for b in bufs:
                    b.copy_(torch.maximum(b, x.abs().max().float().reshape(1)))

    d = OnHost("cpu", limit_bytes=1 << 20)
    a, b = torch.randn(64, 32), torch.randn(16, 8) * 3
    qkv = [torch.zeros(1) for _ in range(3)]
    other = torch.full((1,), 100.0)
    for buf in qkv:
        assert d.add(a, 0, buf)
    assert d.add(b, 0, other) and d.add(b, 0, qkv[0])
    assert len(d.entries) == 2 and d.bytes == a.numel() * 4 + b.numel() * 4 and d.stats["requests"] == 5
    d.flush()
    assert not d.entries and d.bytes == 0 and d.stats["flushes"] == 1 and d.stats["tensors"] == 2
    assert qkv[1].item() == qkv[2].item() == a.abs().max().item()
    assert qkv[0].item() == max(a.abs().max().item(), b.abs().max().item()) and other.item() == 100.0
    d.flush()  # nothing noted: no launch
    assert d.stats["flushes"] == 1
    # a write between the note and the flush
    c = torch.randn(8, 8)
    d.add(c, 0, qkv[0])
    c.mul_(2.0)
    with pytest.raises(RuntimeError, match="written in place"):
        d.flush()
    assert not d.entries
    # the byte limit
    big = torch.randn(1 << 18)  # 1 MiB
    d.add(big, 0, qkv[1])
    assert not d.entries and d.stats["flushes"] == 2 and qkv[1].item() >= big.abs().max().item()
    # another device / dtype is not taken
    assert not calib.DeferredAmax("cuda:0").add(a, 0, qkv[0])
