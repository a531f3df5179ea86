"""The whole Python host on CPU, through a host-memory stand-in for the C-ABI (tests/hostmem_backend.py: every
`moq_*` call is served by the oracle's C restatement on host pointers): calibration flows and checkpoint exports on
the tiny Llama / Mixtral of the fixtures, compared with what the reference produced -- on this same kind of CPU, so
activations agree bit for bit and the comparisons are exact where the GPU tests need a tolerance.

What this tier adds: host-side bugs (shapes, scale layouts, key names, promotion rules) are caught without a GPU.
What it does not claim: anything about the HIP kernels -- those are the `-m gpu` tests."""

import numpy as np
import pytest
import torch

import _moa_import
import hostmem_backend
from conftest import pinned_or_live, from_bits

moa = _moa_import.load()

_TD = {"torch.bfloat16": torch.bfloat16, "torch.float32": torch.float32, "torch.uint8": torch.uint8,
       "torch.float16": torch.float16}


@pytest.fixture
def hostmem(monkeypatch):
    return hostmem_backend.install(monkeypatch, moa)


def _llama(g, cases, dtype, impl=None):
    from transformers import LlamaConfig, LlamaForCausalLM

    cfg = LlamaConfig(architectures=["LlamaForCausalLM"], **cases["config"])
    if impl:
        cfg._attn_implementation = impl
    model = LlamaForCausalLM(cfg).to(dtype)
    sd = {k[len("orig/"):]: from_bits(g.raw(k), dtype) for k in g.z.files if k.startswith("orig/")}
    assert not model.load_state_dict(sd, strict=False).unexpected_keys
    return model.eval()


def _compare_state(state, g, cases, exact=True):
    assert sorted(state) == sorted(cases["dtypes"]), set(state) ^ set(cases["dtypes"])
    for key, dts in cases["dtypes"].items():
        if f"exp/{key}" not in g.z.files:
            continue
        got = state[key].detach().cpu().contiguous()
        raw = g.raw(f"exp/{key}")
        if dts in ("torch.float8_e4m3fn", "torch.int8", "torch.uint8"):
            assert np.array_equal(got.view(torch.uint8).numpy().reshape(raw.shape), raw), f"{key}: bytes differ"
        else:
            want = from_bits(raw, _TD[dts])
            assert got.dtype == want.dtype and tuple(got.shape) == tuple(want.shape), f"{key}: {got.dtype} {tuple(got.shape)}"
            assert torch.equal(got.reshape(-1).view(torch.uint8), want.contiguous().reshape(-1).view(torch.uint8)), key


def test_fp8_calibration_and_export_equal_reference(golden, hostmem):
    """FP8 W + A max calibration of the tiny bf16 Llama from the ORIGINAL weights and tokens, then the checkpoint:
    every amax and every exported tensor equals the reference run."""
    g = golden("export_llama_fp8")
    cases = g.cases
    model = _llama(g, cases, torch.bfloat16)
    batches = [torch.from_numpy(g.raw(f"tokens{i}")) for i in range(cases["n_batches"])]
    with torch.no_grad():
        moa.quantize(model, moa.model_quant.FP8_DEFAULT_CFG, lambda m: [m(b) for b in batches])
    for name in cases["linears"]:
        lin = model.get_submodule(name)
        assert torch.equal(lin.weight_quantizer._amax.float().reshape(-1), from_bits(g.raw(f"pre/{name}.w_amax"), torch.float32).reshape(-1)), name
        assert torch.equal(lin.input_quantizer._amax.float().reshape(-1), from_bits(g.raw(f"pre/{name}.in_amax"), torch.float32).reshape(-1)), name
    state = moa.export.export_state_dict(model, torch.bfloat16, lambda: model(torch.ones([1, 2], dtype=torch.long)))
    _compare_state(state, g, cases)


def test_int8_smoothquant_flow_and_export_equal_reference(golden, hostmem):
    pinned_or_live(golden, ["export_llama_int8_sq"], _int8_smoothquant_flow_and_export)


def _int8_smoothquant_flow_and_export(golden):
    g = golden("export_llama_int8_sq")
    cases = g.cases
    model = _llama(g, cases, torch.bfloat16)
    batches = [torch.from_numpy(g.raw(f"tokens{i}")) for i in range(cases["n_batches"])]
    with torch.no_grad():
        moa.quantize(model, moa.model_quant.INT8_SMOOTHQUANT_CFG, lambda m: [m(b) for b in batches])
    for name in cases["linears"]:
        lin = model.get_submodule(name)
        assert torch.equal(lin.input_quantizer._pre_quant_scale, from_bits(g.raw(f"pre/{name}.pre_quant_scale"), torch.bfloat16)), name
        assert torch.equal(lin.weight, from_bits(g.raw(f"pre/{name}.weight"), torch.bfloat16)), name
        assert torch.equal(lin.weight_quantizer._amax.float().reshape(-1), from_bits(g.raw(f"pre/{name}.w_amax"), torch.float32).reshape(-1)), name
        assert torch.equal(lin.input_quantizer._amax.float().reshape(-1), from_bits(g.raw(f"pre/{name}.in_amax"), torch.float32).reshape(-1)), name
    state = moa.export.export_state_dict(model, torch.bfloat16, lambda: model(torch.ones([1, 2], dtype=torch.long)))
    _compare_state(state, g, cases)


def test_fp8_2d_blockwise_and_mxfp4_exports_equal_reference(golden, hostmem):
    """Presets that need no calibration data: FP8 128 x 128 tiles, MXFP4 everywhere, MXFP4 weights under FP8 inputs
    (w4a8_mxfp4_fp8), MXFP4 on the MLP projections only (a partially quantized model) -- every exported tensor byte for
    byte and the whole `quantization` table of hf_quant_config.json (algorithm, group size, exclude_modules with their
    prefix wildcards)."""
    for fixture, cfg in (("export_llama_fp8_2d", moa.model_quant.FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG),
                         ("export_llama_mxfp4", moa.model_quant.MXFP4_DEFAULT_CFG),
                         ("export_llama_w4a8_mxfp4_fp8", moa.model_quant.W4A8_MXFP4_FP8_CFG),
                         ("export_llama_mxfp4_mlp", moa.model_quant.MXFP4_MLP_WEIGHT_ONLY_CFG)):
        g = golden(fixture)
        cases = g.cases
        model = _llama(g, cases, torch.bfloat16)
        moa.quantize(model, cfg, None)
        _compare_state(moa.export.export_state_dict(model, torch.bfloat16), g, cases)
        assert moa.export.hf_quant_config(model)["quantization"] == cases["hf_quant_config"]["quantization"], fixture


def test_summarized_exclude_modules_are_the_shortest_safe_prefix_wildcards():
    """_prefix_wildcard_summarize_exclude_modules (export/quant_utils.py:607-677) on hand-made layer lists."""
    f = moa.export._summarize_excluded
    assert f(["lm_head", "model.embed_tokens"], ["model.layers.0.mlp.up_proj"]) == {"lm_head", "model.embed_tokens"}
    quant = [f"model.layers.{i}.mlp.{p}" for i in range(2) for p in ("up_proj", "down_proj")]
    unq = [f"model.layers.{i}.self_attn.{p}" for i in range(2) for p in ("q_proj", "k_proj")] + ["lm_head"]
    assert f(unq, quant) == {"lm_head", "model.layers.0.self_attn*", "model.layers.1.self_attn*"}
    # `a*` would swallow the quantized `ab.c`: the pair {a, a.*} is tried next and emitted as `a.*`
    assert f(["a.x"], ["ab.c"]) == {"a.*"}
    assert f(["a.x", "a.y"], []) == {"a*"}


@pytest.mark.parametrize("impl", ["sdpa", "eager"])
def test_fp8_kv_cache_flow_equals_reference(golden, hostmem, impl):
    g = golden("export_llama_fp8_kv")
    cases = g.cases
    mq = moa.model_quant
    model = _llama(g, cases, torch.float32, impl)
    batches = [torch.from_numpy(g.raw(f"tokens{i}")) for i in range(cases["n_batches"])]
    cfg = mq.update_quant_cfg_with_kv_cache_quant(mq.FP8_DEFAULT_CFG, mq.FP8_KV_CFG["quant_cfg"])
    with torch.no_grad():
        mq.quantize(model, cfg, lambda m: [m(b) for b in batches])
        logits = model(batches[0]).logits
    for n in cases["attentions"]:
        m = model.get_submodule(n)
        for which in "kv":
            want = from_bits(g.raw(f"{impl}/{n}.{which}_amax"), torch.float32)
            assert torch.equal(getattr(m, f"{which}_bmm_quantizer")._amax.float().reshape(()), want.reshape(())), (n, which)
    assert torch.equal(logits, from_bits(g.raw(f"{impl}/logits"), torch.float32))
    if impl == "sdpa":
        state = moa.export.export_state_dict(model, torch.float32)
        for key in cases["dtypes"]:
            assert torch.equal(state[key].cpu(), from_bits(g.raw(f"exp/{key}"), torch.float32)), key
        assert sorted(state) == cases["exported_keys"]


def test_mixtral_fused_experts_flow_and_export_equal_reference(golden, hostmem):
    from transformers import MixtralConfig, MixtralForCausalLM

    g = golden("moe_fp8")
    cases = g.cases
    model = MixtralForCausalLM(MixtralConfig(architectures=["MixtralForCausalLM"], **cases["config"])).to(torch.float32)
    sd = {k[len("orig/"):]: from_bits(g.raw(k), torch.float32) for k in g.z.files if k.startswith("orig/")}
    assert not model.load_state_dict(sd, strict=False).unexpected_keys
    model.eval()
    batches = [torch.from_numpy(g.raw(f"tokens{i}")) for i in range(cases["n_batches"])]
    with torch.no_grad():
        moa.quantize(model, moa.model_quant.FP8_DEFAULT_CFG, lambda m: [m(b) for b in batches])
        logits = model(batches[0]).logits
    ours = {n: m for n, m in model.named_modules() if isinstance(m, moa.TensorQuantizer)}
    for key in g.z.files:
        if key.startswith("amax/"):
            n = key[len("amax/"):]
            assert torch.equal(ours[n]._amax.float().reshape(-1), from_bits(g.raw(key), torch.float32).reshape(-1)), n
    assert torch.equal(logits, from_bits(g.raw("logits"), torch.float32))
    _compare_state(moa.export.export_state_dict(model, torch.float32), g, cases)


# ------------------------------------------------------------------------------------------------------------------
# mtq.quantize() flows of the reference on the tiny MLP (tests/golden/model_flows.npz), here exact where the GPU tests
# (tests/test_gpu_host.py) need tolerances for the GEMM summation order
import copy

from conftest import DT, assert_bits_equal


class _TinyMLP(torch.nn.Module):
    def __init__(self, w1, w2, b2):
        super().__init__()
        self.fc1 = torch.nn.Linear(w1.shape[1], w1.shape[0], bias=False)
        self.fc2 = torch.nn.Linear(w2.shape[1], w2.shape[0], bias=True)
        self.to(w1.dtype)
        with torch.no_grad():
            self.fc1.weight.copy_(w1); self.fc2.weight.copy_(w2); self.fc2.bias.copy_(b2)

    def forward(self, x):
        return self.fc2(torch.nn.functional.gelu(self.fc1(x)))


def _mlp_flow(golden, name, cfg):
    g = golden("model_flows")
    c = g.cases[name]
    dn = c["dtype"]
    dt = DT[dn]
    model = _TinyMLP(g.t(f"{dn}_w1", dt), g.t(f"{dn}_w2", dt), g.t(f"{dn}_b2", dt))
    batches = [g.t(f"{dn}_x{i}", dt) for i in range(c["n_batches"])]
    q = moa.quantize(model, copy.deepcopy(cfg), lambda m: [m(b) for b in batches])
    return g, c, q, batches, dt


@pytest.mark.parametrize("name", ["int8_max", "fp8_max"])
def test_mlp_max_calibration_equals_reference(golden, hostmem, name):
    pinned_or_live(golden, ["model_flows"], lambda get: _mlp_max_calibration(get, name))


def _mlp_max_calibration(golden, name):
    cfg = moa.model_quant.INT8_DEFAULT_CFG if name == "int8_max" else moa.model_quant.FP8_DEFAULT_CFG
    g, c, q, batches, dt = _mlp_flow(golden, name, cfg)
    for tname, tdtype, tshape in c["tensors"]:
        lname, rest = tname.split("_", 1)
        qn, attr = rest.rsplit("_", 1)
        t = getattr(getattr(getattr(q, lname), qn), "_" + attr)
        assert list(t.shape) == tshape and str(t.dtype) == tdtype, f"{name} {tname}: {t.dtype} {tuple(t.shape)}"
        want = g.t(f"{name}_{tname}")
        assert_bits_equal(t.float().reshape(want.shape), want, f"{name} {tname}")
    assert_bits_equal(q(batches[0]), g.t(f"{name}_y", dt), f"{name} forward with fake quant")


def test_mlp_smoothquant_equals_reference(golden, hostmem):
    g, c, q, batches, dt = _mlp_flow(golden, "int8_sq", moa.model_quant.INT8_SMOOTHQUANT_CFG)
    for lname in ("fc1", "fc2"):
        lin = getattr(q, lname)
        for attr, key in (("input_quantizer._pre_quant_scale", "input_quantizer_pre_quant_scale"),
                          ("weight_quantizer._amax", "weight_quantizer_amax"), ("input_quantizer._amax", "input_quantizer_amax")):
            obj = lin
            for part in attr.split("."):
                obj = getattr(obj, part)
            want = g.t(f"int8_sq_{lname}_{key}")
            assert_bits_equal(obj.float().reshape(want.shape), want, f"{lname} {key}")
        assert_bits_equal(lin.weight.float(), g.t(f"int8_sq_{lname}_wfinal").float().reshape(lin.weight.shape), f"{lname} smoothed weight")
    assert_bits_equal(q(batches[0]), g.t("int8_sq_y", dt), "smoothquant forward")


@pytest.mark.parametrize("name,search", [("int4_awq", "gram"), ("int4_awq", "gemm"), ("int4_awq_bf16", "gemm")])
def test_mlp_awq_lite_equals_reference(golden, hostmem, name, search):
    """awq_lite on CPU with both engines (bf16: the error-GEMM engine through the oracle's contraction): same alpha as the
    reference on every linear, statistics / losses / folded weights within the summation-order tolerances of the GPU
    test (tests/test_gpu_host.py)."""
    cfg = copy.deepcopy(moa.model_quant.INT4_AWQ_CFG)
    cfg["algorithm"]["search"] = search
    g, c, q, batches, dt = _mlp_flow(golden, name, cfg)
    def close(x, want, rtol, what):
        x, want = x.float().reshape(-1), want.float().reshape(-1)
        assert x.shape == want.shape, what
        err = ((x - want).abs() / want.abs().clamp_min(1e-12)).max().item()
        assert err <= rtol, f"{what}: max rel err {err:.3e} > {rtol}"

    f32 = dt == torch.float32
    for lname in ("fc1", "fc2"):
        lin = getattr(q, lname)
        h = lin.awq_lite
        # mean |x| / mean |w| ratios: the reference reduces with torch's fp32 order, the oracle with a wide sum
        close(h.act_scale, g.t(f"{name}_{lname}_act_scale"), 1e-6 if f32 else 2 ** -7, f"{lname} act_scale")
        close(h.weight_scale, g.t(f"{name}_{lname}_weight_scale"), 1e-6 if f32 else 2 ** -7, f"{lname} weight_scale")
        ref_loss = c[f"{lname}_loss"]
        for a, v in h.loss.items():
            rv = ref_loss[str(a)]
            assert abs(float(v) - rv) <= (1e-4 if f32 else 3e-2) * max(rv, 1e-9), f"{name} {lname} loss[alpha={a}] {float(v)} vs {rv}"
        assert h.best_alpha == c[f"{lname}_best_alpha"], f"{name} {lname}: alpha {h.best_alpha}"
        tol = 1e-5 if f32 else 2e-2
        close(h.best_scale, g.t(f"{name}_{lname}_best_scale"), tol, f"{lname} best_scale")
        close(lin.input_quantizer._pre_quant_scale, g.t(f"{name}_{lname}_input_quantizer_pre_quant_scale"), tol, f"{lname} pqs")
        close(lin.weight_quantizer._amax, g.t(f"{name}_{lname}_weight_quantizer_amax"), tol, f"{lname} weight amax")
        close(lin.weight, g.t(f"{name}_{lname}_wfinal", dt), tol * 10, f"{lname} folded weight")


@pytest.mark.parametrize("preset,fmt", [("MXFP8_DEFAULT_CFG", "E4M3"), ("MXFP6_DEFAULT_CFG", "E3M2"),
                                        ("MXINT8_DEFAULT_CFG", "INT8"), ("MXFP4_DEFAULT_CFG", "E2M1")])
def test_mx_presets_configure_dynamic_e8m0_block_quantizers(hostmem, preset, fmt):
    """presets/model/mx{fp8,fp6,int8,fp4}.yaml: weights and inputs in blocks of 32 with E8M0 scales, no calibration;
    the weight QDQ is the fused block kernel's result for that element format."""
    from oracle import oracle

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(64, 32)).to(torch.bfloat16)
    moa.quantize(model, getattr(moa.model_quant, preset), None)
    lin = model[0]
    for q in (lin.weight_quantizer, lin.input_quantizer):
        assert q.is_enabled and q._block_dynamic and q.block_sizes[-1] == 32 and not hasattr(q, "_amax")
    w = lin.weight.detach()
    assert_bits_equal(lin.weight_quantizer(w), oracle.mx_fused_amax_convert(w, 32, fmt, "E8M0", None), preset)
    assert model(torch.randn(3, 64).to(torch.bfloat16)).shape == (3, 32)


class _FlatStack(torch.nn.Module):
    """Linears with plain Gaussian weights: with nearly uniform activation channels the 11 AWQ candidates score almost
    alike, so the roundings the Gram formulation leaves out decide the order -- the near-tie case."""

    def __init__(self, dims, dtype, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.linears = torch.nn.ModuleList()
        for co, ci in dims:
            lin = torch.nn.Linear(ci, co, bias=False)
            with torch.no_grad():
                lin.weight.copy_(torch.randn(co, ci, generator=g) * 0.02)
            self.linears.append(lin)
        self.to(dtype)

    def forward(self, xs):
        return [lin(x) for lin, x in zip(self.linears, xs)]


def _flat_batches(dims, dtype, n, tokens, seed, spread):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        xs = []
        for _, ci in dims:
            ch = torch.exp(spread * torch.randn(ci, generator=g))
            xs.append((torch.randn(tokens, ci, generator=g) * ch).to(dtype))
        out.append(xs)
    return out


def test_awq_lite_near_tie_is_rescored_like_the_reference_structure(hostmem, monkeypatch):
    """search="auto": candidates whose Gram scores lie within the margin are re-scored by the error-GEMM engine (the
    reference's arithmetic, model_calib.py:1489-1495, :1548-1556) and the first minimum of THOSE scores wins -- the
    selection of search="gemm".  Seed 4 is a case where the plain Gram search picks another alpha."""
    from model_optimizer_amd import model_calib

    monkeypatch.setattr(model_calib._WeightCacheBudget, "host_bytes", 1 << 30)
    dims, dt = [(64, 256), (128, 128)], torch.bfloat16

    def run(search, **kw):
        model = _FlatStack(dims, dt, 4)
        cfg = copy.deepcopy(moa.model_quant.INT4_AWQ_CFG)
        cfg["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "search": search, **kw}
        b = _flat_batches(dims, dt, 2, 48, 104, 0.02)
        q = moa.quantize(model, cfg, lambda m: [m(x) for x in b])
        return [lin.awq_lite for lin in q.linears], [lin.weight.detach().clone() for lin in q.linears]

    gemm, w_gemm = run("gemm")
    gram, _ = run("gram")
    auto, w_auto = run("auto")
    assert [h.best_alpha for h in gram] != [h.best_alpha for h in gemm], "fixture no longer holds a flipped near-tie"
    for hg, ha, wg, wa in zip(gemm, auto, w_gemm, w_auto):
        assert ha.use_gram and ha.contenders is not None and len(ha.contenders) > 1
        assert ha.best_alpha == hg.best_alpha
        assert_bits_equal(wa, wg, "folded weight after the re-scored search")
        # the re-scored candidates carry the error-GEMM engine's loss, bit for bit; the others keep the Gram score
        for j, i in enumerate(ha.contenders):
            assert float(ha.exact_buf[j]) == float(hg.loss_buf[i]) == float(ha.loss_buf[i])
        assert ha.num_exact_steps == 2 and ha.num_search_steps == 2
    # margin 0: nothing is re-scored (a single best candidate), the Gram scores decide
    zero, _ = run("auto", tie_margin=0.0)
    assert [h.best_alpha for h in zero] == [h.best_alpha for h in gram] and all(h.contenders is None for h in zero)
    # infinite margin: every candidate is re-scored -> the full table of the error-GEMM engine
    inf, _ = run("auto", tie_margin=float("inf"))
    for hg, hi in zip(gemm, inf):
        assert hi.contenders == list(range(11)) and torch.equal(hi.loss_buf, hg.loss_buf)
        assert hi.gram_loss is not None and len(hi.gram_loss) == 11


def test_awq_lite_with_quantized_inputs(hostmem):
    """W4A8-style AWQ (model_calib.py:1432-1444, :1534-1538, :1642-1653, :1257-1265): the input quantizer is bypassed
    during the search (same alphas as the weight-only run), max-calibrated per channel in the cache pass, and ends up
    enabled with the per-tensor amax of the SMOOTHED activation; a channel axis other than the last one switches the
    search off for that linear (neutral pre_quant_scale)."""
    shapes = [(64, 128), (96, 256)]
    b = _flat_batches(shapes, torch.float32, 2, 16, 1, 0.1)

    def run(input_cfg):
        model = _FlatStack(shapes, torch.float32, 0)
        cfg = copy.deepcopy(moa.model_quant.INT4_AWQ_CFG)
        if input_cfg is not None:
            cfg["quant_cfg"]["*input_quantizer"] = input_cfg
        moa.quantize(model, cfg, lambda m: [m(x) for x in b])
        return model

    plain = run(None)
    quant = run({"num_bits": 8, "axis": None, "enable": True})
    lin_p = [m for m in plain.modules() if hasattr(m, "awq_lite")]
    lin_q = [m for m in quant.modules() if hasattr(m, "awq_lite")]
    assert len(lin_q) == len(shapes)
    for mp, mq_ in zip(lin_p, lin_q):
        assert mq_.awq_lite.is_input_quantized and not mp.awq_lite.is_input_quantized
        assert mq_.awq_lite.best_alpha == mp.awq_lite.best_alpha
        assert torch.equal(mq_.weight, mp.weight) and torch.equal(mq_.weight_quantizer.amax, mp.weight_quantizer.amax)
        iq = mq_.input_quantizer
        assert iq.is_enabled and iq.axis is None and iq.amax.numel() == 1
        per_channel = iq._amax_for_smoothing.reshape(-1)
        assert per_channel.numel() == mq_.weight.shape[1]
        assert torch.equal(iq.amax.reshape(()), (per_channel * iq.pre_quant_scale.reshape(-1)).amax())
    for i, m in enumerate(lin_q):  # per-channel amax = column-wise abs-max over all batches of that linear's input
        want = torch.stack([xs[i].reshape(-1, xs[i].shape[-1]).abs().amax(0) for xs in b]).amax(0)
        assert torch.equal(m.input_quantizer._amax_for_smoothing.reshape(-1), want)
    with pytest.warns(UserWarning, match="Forcing pre_quant_scale=1"):
        off = run({"num_bits": 8, "axis": 0, "enable": True})
    for m in (m for m in off.modules() if hasattr(m, "awq_lite")):
        assert not m.awq_lite.is_enabled and m.awq_lite.best_alpha is None
        assert torch.all(m.input_quantizer.pre_quant_scale == 1) and m.input_quantizer.is_enabled


def test_int4_awq_checkpoint_with_replayed_inputs_is_byte_identical(golden, hostmem):
    """The whole INT4-AWQ flow (awq_lite search -> fold -> per-group amax -> export) from the ORIGINAL weights, with the
    reference run's per-linear inputs replayed: every statistic of the search and every exported byte equal the
    reference's (tests/replay_common.py)."""
    import replay_common

    g, r = golden("export_llama"), golden("export_llama_replay")
    model = _llama(g, g.cases, torch.bfloat16)
    with torch.no_grad():
        q = moa.quantize(model, moa.model_quant.INT4_AWQ_CFG, replay_common.replay_loop(r, "cpu"))
    report = replay_common.stage_report(q, g, r)
    assert not any(report.values()), {k: v for k, v in report.items() if v}
    state = moa.export.export_state_dict(q, torch.bfloat16, lambda: q(torch.ones([1, 2], dtype=torch.long)))
    _compare_state(state, g, g.cases)


def test_w4a8_awq_checkpoint_with_replayed_inputs_is_byte_identical(golden, hostmem):
    """W4A8_AWQ_BETA_CFG (INT4 blocks -> FP8 weights, FP8 inputs) end to end with the reference run's per-linear inputs
    replayed (the input quantizers are bypassed in both passes, so the INT4 run's replay data is this run's too -- asserted
    by gen_golden.gen_export_w4a8): per-channel input amax, its collapse to the smoothed per-tensor amax, the FP8 stage's
    amax and every byte of the exported checkpoint (weight_scale_2, input_scale included) equal the reference's."""
    import replay_common

    base, r, g = golden("export_llama"), golden("export_llama_replay"), golden("export_llama_w4a8")
    model = _llama(base, base.cases, torch.bfloat16)
    with torch.no_grad():
        q = moa.quantize(model, moa.model_quant.W4A8_AWQ_BETA_CFG, replay_common.replay_loop(r, "cpu"))
    replay_common.check_w4a8_state(q, g)
    state = moa.export.export_state_dict(q, torch.bfloat16, lambda: q(torch.ones([1, 2], dtype=torch.long)))
    _compare_state(state, g, g.cases)
    assert any(k.endswith("weight_scale_2") for k in state) and any(k.endswith("input_scale") for k in state)
    assert moa.export.hf_quant_config(q)["quantization"] == g.cases["hf_quant_config"]["quantization"]




def test_awq_lite_ragged_input_width_equals_the_reference_run(golden, hostmem):
    """Cin not a multiple of the INT4 block: zero-padded last block like the reference (get_weight_scale,
    model_calib.py:1453-1469): alpha, scales, folded weight, per-block amax and the fake-quantized output bit for bit."""
    import replay_common

    replay_common.awq_ragged_check(moa, golden, "cpu")


def test_tensor_quantizer_fused_input_pass_equals_unfused_chain(hostmem):
    """Per-tensor input quantizer with a pre_quant_scale: one moq_input_quant call per forward (calibrating, quantizing, or
    both); outputs, running amax and the calibrated amax equal the stage-by-stage path."""
    TQ, Cfg = moa.TensorQuantizer, moa.QuantizerAttributeConfig
    g = torch.Generator().manual_seed(0)
    x1 = (torch.randn(24, 64, generator=g) * torch.exp(torch.randn(64, generator=g))).to(torch.bfloat16)
    x2 = (torch.randn(10, 64, generator=g) * 3).to(torch.bfloat16)
    pqs = torch.exp(torch.randn(64, generator=g) * 0.3).to(torch.bfloat16)
    calls = []
    real = hostmem.moq_input_quant
    hostmem.moq_input_quant = lambda *a: (calls.append(1), real(*a))[1]
    for nb in (8, (4, 3)):
        q = TQ(Cfg(num_bits=nb, axis=None))
        q.pre_quant_scale = pqs
        q.disable_quant()
        q.enable_calib()
        n0 = len(calls)
        assert_bits_equal(q(x1), x1 * pqs, "calibration hands x * s on")
        q(x2)
        q._if_quant = True  # calibrating and quantizing at once needs an amax first
        q.amax = torch.tensor(2.5, dtype=torch.bfloat16)
        y_both = q(x1)
        assert len(calls) - n0 == 3
        q.disable_calib()
        amax = q._calibrator.compute_amax()
        assert amax.item() == max((x1 * pqs).abs().max().item(), (x2 * pqs).abs().max().item())
        v = moa.ops.scale_cols(x1, pqs)
        want = moa.ops.scaled_e4m3(v, q.amax) if nb == (4, 3) else moa.ops.fake_tensor_quant(v, q.amax, 8, False, False)
        assert_bits_equal(y_both, want, f"num_bits={nb}")
        assert_bits_equal(q(x1), want, f"quant only, num_bits={nb}")


def test_smoothquant_composed_with_mxfp4_equals_reference_halves(golden, hostmem):
    pinned_or_live(golden, ["export_llama_int8_sq", "sq_mxfp4"], _smoothquant_composed_with_mxfp4)


def _smoothquant_composed_with_mxfp4(golden):
    """BASELINE configs[4]: MXFP4_SMOOTHQUANT_CFG from the ORIGINAL weights and tokens.  The per-channel scales and the
    folded weights equal the reference's INT8 SmoothQuant run bit for bit (the scale math does not depend on the format);
    the exported packed E2M1 nibbles / E8M0 scales equal the reference's MXFP4QTensor.quantize of those weights; the MX
    quantizers keep no amax; the fake-quantized forward equals the oracle's MX QDQ of (x * s) and the folded weight."""
    from oracle import oracle

    g, mx = golden("export_llama_int8_sq"), golden("sq_mxfp4")
    cases = g.cases
    model = _llama(g, cases, torch.bfloat16)
    batches = [torch.from_numpy(g.raw(f"tokens{i}")) for i in range(cases["n_batches"])]
    cfg = copy.deepcopy(moa.model_quant.MXFP4_SMOOTHQUANT_CFG)
    cfg["algorithm"]["alpha"] = mx.cases["alpha"]
    with torch.no_grad():
        moa.quantize(model, cfg, lambda m: [m(b) for b in batches])
    for name in cases["linears"]:
        lin = model.get_submodule(name)
        assert torch.equal(lin.input_quantizer._pre_quant_scale, from_bits(g.raw(f"pre/{name}.pre_quant_scale"), torch.bfloat16)), name
        assert torch.equal(lin.weight, from_bits(g.raw(f"pre/{name}.weight"), torch.bfloat16)), name
        assert lin.input_quantizer.amax is None and lin.input_quantizer.axis is None and lin.input_quantizer.is_mx_format
    state = moa.export.export_state_dict(model, torch.bfloat16, lambda: model(torch.ones([1, 2], dtype=torch.long)))
    for name in cases["linears"]:
        for key in ("weight", "weight_scale"):
            got = state[f"{name}.{key}"].contiguous().view(torch.uint8).numpy()
            want = mx.raw(f"mx/{name}.{key}")
            assert np.array_equal(got.reshape(want.shape), want), f"{name}.{key}"
        assert torch.equal(state[f"{name}.pre_quant_scale"], from_bits(g.raw(f"pre/{name}.pre_quant_scale"), torch.bfloat16))
    assert moa.export.hf_quant_config(model)["quantization"]["quant_algo"] == "mxfp4"
    # one linear's fake-quantized forward: MX QDQ of the scaled input times MX QDQ of the folded weight
    lin = model.get_submodule(cases["linears"][0])
    x = (torch.randn(8, lin.weight.shape[1], generator=torch.Generator().manual_seed(0)) * 2).to(torch.bfloat16)
    xq = oracle.mx_fused_amax_convert(x * lin.input_quantizer._pre_quant_scale, 32, "E2M1")
    wq = oracle.mx_fused_amax_convert(lin.weight.detach(), 32, "E2M1")
    assert_bits_equal(lin(x), torch.nn.functional.linear(xq, wq), "MXFP4 forward of a smoothed linear")
    # the reference's own behaviour stays the default: formats="int8" skips MX linears
    model2 = _llama(g, cases, torch.bfloat16)
    cfg2 = copy.deepcopy(moa.model_quant.MXFP4_SMOOTHQUANT_CFG)
    cfg2["algorithm"] = {"method": "smoothquant", "alpha": 1.0}
    with torch.no_grad(), pytest.warns(UserWarning, match="Only int8 smoothing"):
        moa.quantize(model2, cfg2, lambda m: [m(b) for b in batches[:1]])
    assert model2.get_submodule(cases["linears"][0]).input_quantizer.pre_quant_scale is None


def test_weight_statistics_are_collected_once_per_max_calibration(monkeypatch):
    """The reference re-collects every weight's abs-max on every calibration forward (model_calib.py:351-362: the weight
    quantizers stay in calibration mode through forward_loop) -- 0.4 GB of reads per decoder layer and batch for
    Llama-3-8B.  A running abs-max of an unchanged tensor is idempotent, so here weight_only_quantize's one pass is the
    only one: the forward loop must not reduce a weight again, and the result is what the redundant collection gives."""
    import copy

    import hostmem_backend

    hostmem_backend.install(monkeypatch, moa)
    from model_optimizer_amd import model_calib, ops

    def build():
        torch.manual_seed(0)
        return torch.nn.Sequential(torch.nn.Linear(64, 128, bias=False), torch.nn.ReLU(), torch.nn.Linear(128, 32))

    batches = [torch.randn(8, 64) for _ in range(4)]
    reduced = []
    orig = ops.reduce_amax
    monkeypatch.setattr(ops, "reduce_amax", lambda x, *a, **k: (reduced.append(tuple(x.shape)), orig(x, *a, **k))[1])
    for preset in ("FP8_DEFAULT_CFG", "INT8_DEFAULT_CFG"):
        reduced.clear()
        model = moa.quantize(build(), copy.deepcopy(getattr(moa.model_quant, preset)), lambda m: [m(b) for b in batches])
        weight_shapes = {(128, 64), (32, 128)}
        assert sum(s in weight_shapes for s in reduced) == 2, reduced  # once per weight, not once per weight and batch
        # the same amax as with the re-collection forced on every forward
        redo = build()
        moa.nn.replace_quant_module(redo)
        moa.model_quant.set_quantizer_by_cfg(redo, getattr(moa.model_quant, preset)["quant_cfg"])
        monkeypatch.setattr(moa.TensorQuantizer, "mark_weight_stats_done", lambda self, w: None)
        model_calib.max_calibrate(redo, lambda m: [m(b) for b in batches])
        monkeypatch.undo()
        hostmem_backend.install(monkeypatch, moa)
        monkeypatch.setattr(ops, "reduce_amax", lambda x, *a, **k: (reduced.append(tuple(x.shape)), orig(x, *a, **k))[1])
        for (n, q), (_, r) in zip(model.named_modules(), redo.named_modules()):
            if hasattr(q, "_amax"):
                assert torch.equal(q._amax, r._amax), n


# ------------------------------------------------------------------------------------------------ layer-local awq_lite
class _Block(torch.nn.Module):
    def __init__(self, d, g):
        super().__init__()
        self.q = torch.nn.Linear(d, d, bias=False)
        self.k = torch.nn.Linear(d, d // 2, bias=False)
        self.up = torch.nn.Linear(d, 2 * d, bias=False)
        self.down = torch.nn.Linear(2 * d, d, bias=False)
        with torch.no_grad():
            for lin in (self.q, self.k, self.up, self.down):
                lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * 0.05)

    def forward(self, h, scale=1.0):
        a = self.q(h) + torch.nn.functional.pad(self.k(h), (0, h.shape[-1] // 2))  # q and k read the SAME tensor
        m = h + scale * a
        return m + self.down(torch.nn.functional.gelu(self.up(m)))


class _Stack(torch.nn.Module):
    def __init__(self, d=128, n=3, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.embed = torch.nn.Linear(d, d, bias=False)
        self.layers = torch.nn.ModuleList([_Block(d, g) for _ in range(n)])

    def forward(self, x):
        h = self.embed(x)
        for layer in self.layers:
            h = layer(h, scale=0.5)
        return h


def _stack_batches(d=128, n=3):
    g = torch.Generator().manual_seed(7)
    ch = torch.exp(torch.randn(d, generator=g))
    ch[:2] *= 30
    return [(torch.randn(32, d, generator=g) * ch).to(torch.bfloat16) for _ in range(n)]


@pytest.mark.parametrize("tie_margin", [None, float("inf"), 0.05])
def test_awq_lite_layer_local_equals_the_whole_model_flow(hostmem, monkeypatch, tie_margin):
    """awq_lite walking the decoder stack one layer at a time -- ONE forward through every layer, the near-ties re-scored from
    the stored activations (no second pass) -- picks the alphas, folds the weights and leaves the scales of the whole-model
    two-pass flow, bit for bit; with tie_margin = inf every candidate of every linear goes through the replayed exact pass."""
    import copy

    from model_optimizer_amd import model_calib, model_quant

    monkeypatch.setattr(model_calib._WeightCacheBudget, "host_bytes", 1 << 30)
    base = _Stack().to(torch.bfloat16)
    batches = _stack_batches()
    cfg = copy.deepcopy(model_quant.INT4_AWQ_CFG)
    cfg["quant_cfg"]["*embed*"] = {"enable": False}
    cfg["quant_cfg"]["*weight_quantizer"] = {"num_bits": 4, "block_sizes": {-1: 32, "type": "static"}, "enable": True}

    def run(layer_local, store="auto"):
        c = copy.deepcopy(cfg)
        c["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "search": "auto", "layer_local": layer_local,
                          "store_activations": store, **({"tie_margin": tie_margin} if tie_margin is not None else {})}
        calls = {"n": 0}

        def loop(m):
            calls["n"] += 1
            for b in batches:
                m(b)

        with torch.no_grad():
            q = moa.quantize(copy.deepcopy(base), c, loop)
        return q, dict(model_calib.AWQ_LITE_STATS), calls["n"]

    whole, ws, wn = run(False, store=False)  # the two-pass structure: a second forward for the exact pass
    local, ls, ln = run(True)
    kept, ks, kn = run(False)  # whole model, inputs kept by tensor identity, out_actual recomputed in the replay
    assert ls.get("layer_local") and ls["passes"] == 1 and ls["layers"] == 3 and ln == 1
    if tie_margin is not None:
        assert ls["replayed_passes"] >= 3 and ws["passes"] >= 2 and wn >= 2  # the whole-model flow needed a second forward
        assert ls["rescored_candidates"] == ws["rescored_candidates"] > 0
        assert ks["passes"] == 1 and kn == 1 and ks["replayed_passes"] >= 1 and ks["stored_input_bytes"] > 0, ks
        assert ks["rescored_candidates"] == ws["rescored_candidates"]
    for (name, a), (_, b) in zip(whole.named_modules(), kept.named_modules()):
        if hasattr(a, "awq_lite"):
            assert a.awq_lite.best_alpha == b.awq_lite.best_alpha and a.awq_lite.contenders == b.awq_lite.contenders, name
            assert torch.equal(a.awq_lite.loss_buf, b.awq_lite.loss_buf), name
            assert torch.equal(a.weight, b.weight), name
    n = 0
    for (name, a), (_, b) in zip(whole.named_modules(), local.named_modules()):
        if hasattr(a, "awq_lite"):
            n += 1
            assert a.awq_lite.best_alpha == b.awq_lite.best_alpha, name
            assert a.awq_lite.contenders == b.awq_lite.contenders, name
            assert torch.equal(a.awq_lite.loss_buf, b.awq_lite.loss_buf), name
            assert torch.equal(a.weight, b.weight), name
            assert torch.equal(a.input_quantizer.pre_quant_scale, b.input_quantizer.pre_quant_scale), name
            assert torch.equal(a.weight_quantizer.amax, b.weight_quantizer.amax), name
    assert n == 12


def test_awq_lite_layer_local_is_the_default_on_a_hugging_face_stack(golden, hostmem, monkeypatch):
    """search="auto" on a transformers decoder stack driven by its own forward: the layer-local single pass is chosen by
    itself (KV-cache objects among the layer arguments are dropped) and equals the whole-model flow; a forward loop that
    feeds the linears directly (the replay fixtures) never reaches the first decoder layer and falls back."""
    import copy

    from model_optimizer_amd import model_calib, model_quant

    monkeypatch.setattr(model_calib._WeightCacheBudget, "host_bytes", 1 << 30)
    g = golden("export_llama")
    base = _llama(g, g.cases, torch.bfloat16)
    vocab = base.config.vocab_size
    toks = [torch.randint(0, vocab, (2, 24), generator=torch.Generator().manual_seed(i)) for i in range(3)]

    def run(layer_local):
        c = copy.deepcopy(model_quant.INT4_AWQ_CFG)
        c["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "search": "auto", "tie_margin": 0.02,
                          **({} if layer_local is None else {"layer_local": layer_local})}
        with torch.no_grad():
            q = moa.quantize(copy.deepcopy(base), c, lambda m: [m(t) for t in toks])
        return q, dict(model_calib.AWQ_LITE_STATS)

    whole, ws = run(False)
    auto, st = run(None)
    assert st.get("layer_local") and st["passes"] == 1 and not ws.get("layer_local")
    n = 0
    for (name, a), (_, b) in zip(whole.named_modules(), auto.named_modules()):
        if hasattr(a, "awq_lite"):
            n += 1
            assert a.awq_lite.best_alpha == b.awq_lite.best_alpha, name
            assert torch.equal(a.weight, b.weight), name
            assert torch.equal(a.weight_quantizer.amax, b.weight_quantizer.amax), name
    assert n == 7 * base.config.num_hidden_layers


def test_fp8_per_channel_per_token_flow_and_export_equal_reference(golden, hostmem):
    """FP8_PER_CHANNEL_PER_TOKEN_CFG (presets/model/fp8_per_channel_per_token.yaml): per-output-channel E4M3 weights, inputs
    with a dynamic abs-max per token.  From the ORIGINAL weights and tokens (export_llama_fp8's): the per-channel weight amax
    and every byte of the fp8_pc_pt checkpoint -- E4M3 weights from the fp32-promoted quotient, fp32 [Cout] weight_scale, no
    input_scale -- equal the reference run (export/quant_utils.py:545-547, :879-909)."""
    g, base = golden("export_llama_fp8_pc_pt"), golden("export_llama_fp8")
    cases = g.cases
    model = _llama(base, cases, torch.bfloat16)
    batches = [torch.from_numpy(base.raw(f"tokens{i}")) for i in range(cases["n_batches"])]
    with torch.no_grad():
        moa.quantize(model, moa.model_quant.FP8_PER_CHANNEL_PER_TOKEN_CFG, lambda m: [m(b) for b in batches])
    for name in cases["linears"]:
        lin = model.get_submodule(name)
        assert torch.equal(lin.weight_quantizer._amax.float().reshape(-1),
                           from_bits(g.raw(f"pre/{name}.w_amax"), torch.float32).reshape(-1)), name
        assert getattr(lin.input_quantizer, "_amax", None) is None, name
        assert moa.export.get_quantization_format(lin) == moa.export.QUANTIZATION_FP8_PC_PT
    state = moa.export.export_state_dict(model, torch.bfloat16, lambda: model(torch.ones([1, 2], dtype=torch.long)))
    _compare_state(state, g, cases)
    assert not any(k.endswith("input_scale") for k in state)
    assert moa.export.hf_quant_config(model)["quantization"] == cases["hf_quant_config"]["quantization"]
    assert cases["hf_quant_config"]["quantization"]["quant_algo"] == "FP8_PER_CHANNEL_PER_TOKEN"


def test_awq_lite_layer_local_falls_back_when_the_stores_do_not_fit_or_the_loop_bypasses_the_stack(hostmem, monkeypatch):
    """Two ways out of the single-pass flow, both back to the same result: (1) the stored activations do not fit the HBM
    budget -- the stores are dropped in the middle of the cache pass and the exact pass of that layer is a real second
    forward through the layer (`passes` == 2, nothing replayed); (2) the forward loop never calls the first decoder layer
    (it feeds the linears directly, like the replay fixtures) -- the whole-model flow runs."""
    import copy

    from model_optimizer_amd import model_calib, model_quant

    base = _Stack().to(torch.bfloat16)
    batches = _stack_batches()
    cfg = copy.deepcopy(model_quant.INT4_AWQ_CFG)
    cfg["quant_cfg"]["*embed*"] = {"enable": False}
    cfg["quant_cfg"]["*weight_quantizer"] = {"num_bits": 4, "block_sizes": {-1: 32, "type": "static"}, "enable": True}

    def run(layer_local, budget, loop=None):
        monkeypatch.setattr(model_calib._WeightCacheBudget, "host_bytes", budget)
        c = copy.deepcopy(cfg)
        c["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "search": "auto", "layer_local": layer_local, "tie_margin": 0.05}
        with torch.no_grad():
            q = moa.quantize(copy.deepcopy(base), c, loop or (lambda m: [m(b) for b in batches]))
        return q, dict(model_calib.AWQ_LITE_STATS)

    def same(a, b):
        for (name, x), (_, y) in zip(a.named_modules(), b.named_modules()):
            if hasattr(x, "awq_lite"):
                assert x.awq_lite.best_alpha == y.awq_lite.best_alpha and torch.equal(x.weight, y.weight), name

    whole, _ = run(False, 1 << 30)
    roomy, st = run(True, 1 << 30)
    assert st["passes"] == 1 and st["replayed_passes"] >= 1
    # room for the Gram matrices (3 x 128^2 + 256^2 floats per layer) and a few batches of stores, not for all of them
    tight, st = run(True, 600_000)
    assert st["layer_local"] and st["passes"] == 2 and st["rescored_candidates"] > 0, st
    same(whole, roomy)
    same(whole, tight)

    # (2) a loop that drives the linears itself: every linear gets a tensor, the decoder layers are never called
    def direct(m):
        for b in batches:
            h = b
            for layer in m.layers:
                layer.q(h), layer.k(h), layer.up(h), layer.down(torch.cat([h, h], dim=-1))

    a, st = run(None, 1 << 30, direct)  # (None = auto: not a Hugging Face model -> whole-model flow anyway)
    assert not st.get("layer_local")
    b, st = run(True, 1 << 30, direct)   # asked for explicitly, but the stack is never entered: falls back
    assert not st.get("layer_local")
    same(a, b)


def test_small_model_quant_api(hostmem, tmp_path, capsys):
    """mtq.calibrate / postprocess_amax / disable_quantizer / enable_quantizer / print_quant_summary
    (quantization/model_quant.py:64-144, :698-725): quantizers put in place without an algorithm, then calibrated by the
    separate call (a forward loop without an argument is accepted with the deprecation warning); amax post-processing by
    wildcard; toggling by wildcard and by filter function; the summary printed and written."""
    import warnings

    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(64, 64), torch.nn.ReLU(), torch.nn.Linear(64, 32))
    x = torch.randn(8, 64)
    cfg = copy.deepcopy(moa.model_quant.INT8_DEFAULT_CFG)
    cfg["algorithm"] = None
    moa.quantize(model, cfg, None)
    assert not hasattr(model[0].weight_quantizer, "_amax")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        moa.calibrate(model, "max", lambda: model(x))
    assert any(issubclass(m.category, DeprecationWarning) for m in w)
    a0 = model[0].input_quantizer.amax.clone()
    assert torch.equal(a0.reshape(()), x.abs().max())
    moa.postprocess_amax(model, "0.input_quantizer", lambda a: a * 2)
    assert torch.equal(model[0].input_quantizer.amax, a0 * 2) and torch.equal(model[2].weight_quantizer.amax.reshape(-1),
                                                                               model[2].weight.abs().amax(dim=1))
    moa.disable_quantizer(model, "*input_quantizer")
    assert not model[0].input_quantizer.is_enabled and not model[2].input_quantizer.is_enabled and model[0].weight_quantizer.is_enabled
    moa.enable_quantizer(model, lambda name: name.startswith("2."))
    assert model[2].input_quantizer.is_enabled and not model[0].input_quantizer.is_enabled
    moa.print_quant_summary(model)
    out = capsys.readouterr().out
    assert "0.input_quantizer" in out and "TensorQuantizers found in model" in out
    moa.print_quant_summary(model, str(tmp_path))
    text = open(tmp_path / ".quant_summary.txt").read()
    assert text.rstrip().endswith("TensorQuantizers found in model") and "2.weight_quantizer" in text
    training = model.training
    moa.calibrate(model, {"method": "mse"}, lambda m: m(x))
    assert model.training == training

