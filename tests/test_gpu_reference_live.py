"""The REAL reference on the MI355X, with device tensors (SURVEY.md 8b seams S1 / S2 / S3 / S5 / S6).

The reference's Python package reaches the GPU box as the gitignored archive `tools/stage_reference.sh` packs under
oracle/_ref/ (tests/golden/ref_shim.py unpacks it into a temp dir; in the build container the checkout itself is used, but
there is no GPU there).  Test infrastructure only -- the product never imports it.  Skipped ONLY when neither is present.

  A. modelopt_plugin.install() + the reference's own mtq.quantize(model.cuda(), <preset>, loop): every state_dict entry and
     the fake-quantized logits equal the UN-INSTALLED run of the same reference in the same process (its eager torch ops on
     the same device) bit for bit -- the bar of the reference's own GPU tests, atol = 0 against eager
     (tests/gpu/torch/quantization/test_tensor_quant_cuda.py:55-119).  The seam counters prove the reference's unmodified
     call sites (tensor_quant.py:83-91, :103-111, :184-191; calib/max.py:63-64) reached our adapters, without fallbacks.
     MX formats have no eager implementation: the reference through the seams must equal this package's own quantize(), and
     the literal vectors of the reference's MX tests must come out of the reference's TensorQuantizer.
  B. this package's quantize() + export on the device against the reference's eager run on the same device: amax of every
     quantizer, logits, and every checkpoint tensor byte for byte (the CPU tier's live differential, with device tensors).
  C. the reference's OWN GPU test files for this path, unmodified, in a pytest subprocess with the seams installed.
  D. mts.sparsify (magnitude 2:4) on the device through S5 == the un-installed run.
"""

import contextlib
import copy
import json
import os
import re
import subprocess
import sys

import pytest
import torch

import _moa_import
from conftest import GOLDEN, ROOT, note

moa = _moa_import.load()
sys.path.insert(0, GOLDEN)
import ref_shim  # noqa: E402

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not ref_shim.reference_available(),
                                 reason="no reference: oracle/_ref/reference_modelopt.tgz absent (tools/stage_reference.sh)")]
DEV = "cuda"


@pytest.fixture(scope="module")
def ref():
    ref_shim.install()
    import modelopt.torch.quantization as mtq

    from model_optimizer_amd import modelopt_plugin

    modelopt_plugin.uninstall()
    yield mtq
    modelopt_plugin.uninstall()


@contextlib.contextmanager
def installed(**kw):
    from model_optimizer_amd import modelopt_plugin

    got = modelopt_plugin.install(**kw)
    modelopt_plugin.STATS.clear()
    try:
        yield modelopt_plugin, got
    finally:
        modelopt_plugin.uninstall()


def _tiny(dtype=torch.bfloat16):
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, vocab_size=64, max_position_embeddings=64, architectures=["LlamaForCausalLM"])
    return LlamaForCausalLM(cfg).to(dtype).eval().to(DEV)


def _batches():
    return [torch.randint(0, 64, (4, 32), generator=torch.Generator().manual_seed(i)).to(DEV) for i in range(3)]


def _reference_quantize(mtq, preset, dtype, extra=None):
    m = _tiny(dtype)
    cfg = copy.deepcopy(getattr(mtq, preset))
    if extra:
        cfg["quant_cfg"] = list(cfg["quant_cfg"]) + copy.deepcopy(extra)
    batches = _batches()
    with torch.no_grad():
        q = mtq.quantize(m, cfg, (lambda mm: [mm(b) for b in batches]) if cfg.get("algorithm") else None)
        logits = q(batches[0]).logits.clone()
    return {n: t.detach().clone() for n, t in q.state_dict().items()}, logits


# ------------------------------------------------------------------------------------------------------------- A
SEAM_CASES = [
    ("FP8_DEFAULT_CFG", torch.bfloat16, ["S1:fake_e4m3fy", "S6:reduce_amax"]),
    ("FP8_DEFAULT_CFG", torch.float16, ["S1:fake_e4m3fy", "S6:reduce_amax"]),
    ("INT8_DEFAULT_CFG", torch.bfloat16, ["S1:fake_tensor_quant", "S1:fake_tensor_quant_with_axis", "S6:reduce_amax"]),
    ("INT8_SMOOTHQUANT_CFG", torch.float32, ["S1:fake_tensor_quant_with_axis", "S6:reduce_amax"]),
    ("INT4_AWQ_CFG", torch.bfloat16, ["S1:fake_tensor_quant_with_axis", "S6:reduce_amax"]),
    ("INT4_BLOCKWISE_WEIGHT_ONLY_CFG", torch.float16, ["S1:fake_tensor_quant_with_axis", "S6:reduce_amax"]),
    ("W4A8_AWQ_BETA_CFG", torch.bfloat16, ["S1:fake_tensor_quant_with_axis", "S1:fake_e4m3fy", "S6:reduce_amax"]),
    ("FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG", torch.bfloat16, ["S6:reduce_amax"]),
    ("FP8_PER_CHANNEL_PER_TOKEN_CFG", torch.bfloat16, ["S6:reduce_amax"]),
]


@pytest.mark.parametrize("preset,dtype,expected", SEAM_CASES)
def test_reference_quantize_on_the_device_through_installed_seams(ref, preset, dtype, expected):
    base_state, base_logits = _reference_quantize(ref, preset, dtype)
    with installed() as (plugin, got):
        assert "S1:extensions" in got and "S6:reduce_amax" in got
        our_state, our_logits = _reference_quantize(ref, preset, dtype)
        stats = dict(plugin.STATS)
    for key in expected:
        assert stats.get(key, 0) > 0, f"{preset}: the reference never reached {key}: {stats}"
    assert not [k for k in stats if "fallback" in k], f"{preset}: seams handed calls back to the reference: {stats}"
    assert set(base_state) == set(our_state)
    for n in base_state:
        assert torch.equal(base_state[n], our_state[n]), f"{preset}: {n} differs from the un-installed reference run"
    assert torch.equal(base_logits, our_logits), f"{preset}: logits differ from the un-installed reference run"
    note(f"reference on the device through the seams, {preset} {str(dtype)[6:]}: {len(base_state)} state entries + logits "
         f"== un-installed eager run; seam calls {sum(stats.values())} ({', '.join(f'{k}={v}' for k, v in sorted(stats.items()))})")


def test_reference_quantize_with_the_library_op_seam(ref):
    """S2: tensor_quant.quantize_op / dynamic_block_quantize_op re-pointed at the moquant:: torch.library operators."""
    base_state, base_logits = _reference_quantize(ref, "FP8_DEFAULT_CFG", torch.bfloat16)
    with installed(library_ops=True) as (plugin, got):
        assert "S2:library_ops" in got
        our_state, our_logits = _reference_quantize(ref, "FP8_DEFAULT_CFG", torch.bfloat16)
    for n in base_state:
        assert torch.equal(base_state[n], our_state[n]), n
    assert torch.equal(base_logits, our_logits)


@pytest.mark.parametrize("preset", ["MXFP4_DEFAULT_CFG", "MXFP8_DEFAULT_CFG", "W4A8_MXFP4_FP8_CFG"])
def test_reference_mx_presets_through_the_seams_equal_this_package(ref, preset):
    """The reference's MX fake quantization exists only as its CUDA extension: un-installed there is nothing to run.  Under
    install() its TensorQuantizer / QuantLinear code drives our MX kernel through S1; this package's quantize() of the same
    model reaches the same kernel from its own host code -- identical logits."""
    with installed() as (plugin, _):
        _, ref_logits = _reference_quantize(ref, preset, torch.bfloat16)
        stats = dict(plugin.STATS)
    assert stats.get("S1:fused_amax_convert", 0) > 0, stats
    assert not [k for k in stats if "fallback" in k], stats
    m = _tiny(torch.bfloat16)
    batches = _batches()
    cfg = copy.deepcopy(getattr(moa.model_quant, preset))
    with torch.no_grad():
        moa.quantize(m, cfg, (lambda mm: [mm(b) for b in batches]) if cfg.get("algorithm") else None)
        mine = m(batches[0]).logits
    assert torch.equal(mine, ref_logits), f"{preset}: reference through the seams vs this package"


def test_reference_tensor_quantizer_reproduces_the_reference_mx_vectors(ref):
    """The literal vectors of the reference's MX tests (tests/gpu/torch/quantization/test_quantize_mxformats_cuda.py:59-179,
    held as tests/golden/mx_vectors.json) out of the reference's OWN TensorQuantizer, whose dynamic-block branch
    (tensor_quant.py:157-195) calls get_cuda_ext_mx().fused_amax_convert == our adapter."""
    from modelopt.torch.quantization.config import QuantizerAttributeConfig
    from modelopt.torch.quantization.nn import TensorQuantizer

    bits = {"E2M1": (2, 1), "E3M2": (3, 2), "E2M3": (2, 3), "E4M3": (4, 3), "E5M2": (5, 2), "INT8": 8}
    cases = json.load(open(os.path.join(GOLDEN, "mx_vectors.json")))
    ran = 0
    with installed() as (plugin, _):
        for c in cases:
            if c["fmt"] not in bits:
                continue
            for bs in ([c["block_size"]] if c["block_size"] else [8, 16, 32]):
                for dt in ([torch.float32] if c["dtype"] else [torch.float32, torch.float16, torch.bfloat16]):
                    rep = max(bs // c["in_size"], 1)
                    tin = torch.tensor(c["test_in"], dtype=dt).repeat(1, rep).to(DEV)
                    tout = torch.tensor(c["test_out"], dtype=dt).repeat(1, rep)
                    q = TensorQuantizer(QuantizerAttributeConfig(
                        num_bits=bits[c["fmt"]], block_sizes={-1: bs, "type": "dynamic", "scale_bits": (8, 0)})).to(DEV)
                    got = q(tin).cpu()
                    assert torch.allclose(got.float(), tout.float(), rtol=1e-5, atol=c["atol"]), f"{c['fn']} {c['fmt']} bs={bs} {dt}"
                    ran += 1
        assert plugin.STATS.get("S1:fused_amax_convert", 0) == ran
    assert ran >= 10


def test_reference_quant_backend_seam_on_the_device(ref):
    """S3: a reference TensorQuantizer configured with backend="mi355x" (tensor_quantizer.py:87-129, :892-896) == the same
    quantizer on the reference's eager path."""
    from modelopt.torch.quantization.config import QuantizerAttributeConfig
    from modelopt.torch.quantization.nn import TensorQuantizer

    x = (torch.randn(64, 512, generator=torch.Generator().manual_seed(3)) * 0.05).to(torch.bfloat16).to(DEV)
    for cfg in (dict(num_bits=8, axis=None), dict(num_bits=(4, 3), axis=None), dict(num_bits=8, axis=0)):
        plain = TensorQuantizer(QuantizerAttributeConfig(**cfg)).to(DEV)
        want = plain(x)
        with installed() as (plugin, got):
            assert "S3:backend=mi355x" in got
            q = TensorQuantizer(QuantizerAttributeConfig(backend="mi355x", **cfg)).to(DEV)
            out = q(x)
            assert plugin.STATS.get("S3:mi355x_backend", 0) == 1
        assert torch.equal(out, want), cfg


# ------------------------------------------------------------------------------------------------------------- B
import test_differential_cpu as diff  # noqa: E402  (helpers only; its tests are not gpu-marked)

DEVICE_DIFF_CASES = [
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "llama"), ("FP8_DEFAULT_CFG", torch.float16, "cast", "llama"),
    ("FP8_DEFAULT_CFG", torch.bfloat16, "affine", "llama"),
    ("INT8_SMOOTHQUANT_CFG", torch.bfloat16, False, "llama"), ("INT8_DEFAULT_CFG", torch.float32, False, "opt"),
    ("INT4_BLOCKWISE_WEIGHT_ONLY_CFG", torch.bfloat16, False, "llama"),
    ("FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG", torch.bfloat16, False, "llama"),
    ("FP8_PER_CHANNEL_PER_TOKEN_CFG", torch.bfloat16, False, "llama"),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "mixtral"), ("INT8_WEIGHT_ONLY_CFG", torch.bfloat16, False, "qwen3_moe"),
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "gemma2"), ("FP8_DEFAULT_CFG", torch.float32, False, "gpt2"),
]


@pytest.mark.parametrize("preset,dtype,with_kv,arch", DEVICE_DIFF_CASES)
def test_this_package_on_the_device_equals_the_reference_eager_run_on_the_device(ref, preset, dtype, with_kv, arch):
    """Same tiny model, same batches, both sides on cuda:0: the reference's eager torch ops vs the HIP library through the
    C-ABI.  The model's own GEMMs are the same library kernels on both sides, every quantize-dequantize in between is
    bit-exact, so activations -- and with them every activation amax -- must agree exactly, not within a tolerance.
    The small-vector scale math runs in numerics mode "device" here: the reference writes a different checkpoint from a GPU
    run than from a CPU run (torch's GPU `tensor / scalar` multiplies by a reciprocal: every `amax / maxbound` scale can
    move by an ulp -- first seen as 1 ulp in down_proj.input_scale of this very test), and the default "host" mode equals its
    CPU run (the committed fixtures); "device" must equal the run on THIS device byte for byte."""
    ref_amax, ref_state = diff._reference_run(preset, dtype, with_kv, arch, None, device=DEV)
    with moa.numerics.scale_math("device"):
        our_amax, our_state = diff._our_run(preset, dtype, with_kv, arch, None, device=DEV)
    for n, a in ref_amax.items():
        assert n in our_amax, f"{preset}: quantizer {n} has no amax here"
        assert torch.equal(our_amax[n].reshape(-1), a.reshape(-1)), f"{preset}: amax of {n} differs"
    ref_json, our_json = ref_state.pop("__quant_json__", None), our_state.pop("__quant_json__", None)
    ref_logits, our_logits = ref_state.pop("__logits__"), our_state.pop("__logits__")
    assert torch.equal(our_logits, ref_logits), f"{preset}: logits of the fake-quantized model differ"
    assert sorted(our_state) == sorted(ref_state), set(our_state) ^ set(ref_state)
    if ref_json is not None and ref_json[0] is not None:
        diff._assert_same_quant_json(our_json, ref_json, f"{preset} {arch}")
    for k, want in ref_state.items():
        got = our_state[k].detach().cpu()
        assert got.dtype == want.dtype and tuple(got.shape) == tuple(want.shape), k
        assert torch.equal(got.contiguous().reshape(-1).view(torch.uint8), want.contiguous().reshape(-1).view(torch.uint8)), f"{preset}: {k} differs"
    note(f"this package vs reference eager, both on the device, {preset} {arch} {str(dtype)[6:]}: {len(ref_amax)} amax, logits, "
         f"{len(ref_state)} checkpoint tensors byte-identical")


@pytest.mark.parametrize("preset,dtype,arch", [("INT8_DEFAULT_CFG", torch.bfloat16, "llama"), ("INT8_DEFAULT_CFG", torch.float16, "opt"),
                                               ("FP8_DEFAULT_CFG", torch.bfloat16, "llama"),
                                               ("INT4_BLOCKWISE_WEIGHT_ONLY_CFG", torch.bfloat16, "llama"),
                                               ("INT8_DEFAULT_CFG", torch.float32, "llama")])
def test_mse_calibration_on_the_device_equals_the_reference_eager_run_on_the_device(ref, preset, dtype, arch):
    """algorithm "mse" (calib/mse.py:83-121, model_calib.py:732-826), both sides on cuda:0.  The candidate amax values are
    `initial_amax * multiplier` with a 0-dim fp32 multiplier: for a 16-bit amax a GPU casts the multiplier to that dtype BEFORE
    the product (a CPU multiplies in fp32 and rounds once), and the device's linspace differs from the host's in the last
    ulp -- the candidate grid of the reference's GPU run is not that of its CPU run.  A candidate amax one 16-bit step away
    moves a 72-element row's loss by 2-30 % (tools/diag/mse_device_diff.py: 23 of 384 channels of a flow_fuzz case picked
    another candidate before the fused candidate table followed the numerics mode).  Every picked amax, the logits and every
    checkpoint byte must equal the reference's run on this device."""
    ref_amax, ref_state = diff._reference_run(preset, dtype, False, arch, "mse", device=DEV)
    with moa.numerics.scale_math("device"):
        our_amax, our_state = diff._our_run(preset, dtype, False, arch, "mse", device=DEV)
    bad = [n for n, a in ref_amax.items() if n not in our_amax or not torch.equal(our_amax[n].reshape(-1), a.reshape(-1))]
    assert not bad, f"{preset} mse: {len(bad)} of {len(ref_amax)} amax differ, first {bad[:4]}"
    ref_state.pop("__quant_json__", None), our_state.pop("__quant_json__", None)
    assert torch.equal(our_state.pop("__logits__"), ref_state.pop("__logits__")), f"{preset} mse: logits differ"
    assert sorted(our_state) == sorted(ref_state)
    for k, want in ref_state.items():
        got = our_state[k].detach().cpu()
        assert got.dtype == want.dtype and torch.equal(got.contiguous().reshape(-1).view(torch.uint8),
                                                       want.contiguous().reshape(-1).view(torch.uint8)), f"{preset} mse: {k} differs"
    note(f"mse calibration, this package vs reference eager, both on the device, {preset} {arch} {str(dtype)[6:]}: "
         f"{len(ref_amax)} amax, logits, {len(ref_state)} checkpoint tensors byte-identical")


@pytest.mark.parametrize("arch,dtype", [("llama", torch.bfloat16), ("qwen2", torch.bfloat16), ("opt", torch.float16)])
def test_int4_awq_on_the_device_against_the_reference_eager_search(ref, arch, dtype):
    """INT4-AWQ end to end.  The reference scores its 11 candidates per linear with the library's GEMM, this package with its
    Gram screen + own MFMA error GEMM: the two sum the same products in different orders, so candidates closer than the
    GEMMs' rounding can swap (random-init tiny models are the adversarial case: every candidate within a fraction of a
    percent).  Stated tolerance, evaluated on EVERY searched linear (the search tables of both sides are kept: `debug`) -- in
    OPT the q / k / v and fc1 scales are folded into the LayerNorms at export and leave no `pre_quant_scale` in the checkpoint,
    so the checkpoint alone says nothing about two thirds of its linears: at least 90 % of the
    linears pick the reference's alpha; every other pick must be a tie in the REFERENCE's own loss table (relative gap of the
    two candidates below 2e-3, the summation-order noise of an fp16 / bf16 GEMM at these sizes); and every checkpoint tensor
    that differs must belong to (or be fused with: LayerNorm, shared-input group) a linear with such a pick -- or, with equal
    alphas, differ by the activation mean's summation order only (one 16-bit step in the scale).  Nothing is left unexplained:
    the test fails on a differing tensor it cannot attribute."""
    algo = {"method": "awq_lite", "alpha_step": 0.1, "debug": True}
    tables = {}

    def keep(side):
        def inspect(model):
            tables[side] = {n: (round(float(m.awq_lite.best_alpha), 2),
                                {round(float(a), 2): float(v) for a, v in m.awq_lite.loss.items()},
                                None if m.awq_lite.act_scale is None else m.awq_lite.act_scale.detach().float().cpu().clone())
                            for n, m in model.named_modules() if hasattr(m, "awq_lite")}
        return inspect

    ref_amax, ref_state = diff._reference_run("INT4_AWQ_CFG", dtype, False, arch, algo, device=DEV, inspect=keep("ref"))
    with moa.numerics.scale_math("device"):
        our_amax, our_state = diff._our_run("INT4_AWQ_CFG", dtype, False, arch, algo, device=DEV, inspect=keep("ours"))
    ref_logits, our_logits = ref_state.pop("__logits__"), our_state.pop("__logits__")
    ref_state.pop("__quant_json__", None), our_state.pop("__quant_json__", None)
    assert sorted(our_state) == sorted(ref_state)
    rt, ot = tables["ref"], tables["ours"]
    assert sorted(rt) == sorted(ot) and len(rt) > 0, (sorted(rt), sorted(ot))
    same_alpha = [n for n in rt if rt[n][0] == ot[n][0]]
    other = {n: (rt[n][0], ot[n][0], (rt[n][1][ot[n][0]] - rt[n][1][rt[n][0]]) / max(rt[n][1][rt[n][0]], 1e-30)) for n in rt if n not in same_alpha}
    act_same = [n for n in rt if rt[n][2] is not None and ot[n][2] is not None and torch.equal(rt[n][2], ot[n][2])]
    moved = {n.replace("model.", ""): (int((rt[n][2] != ot[n][2]).sum()), float(((rt[n][2] - ot[n][2]).abs() / rt[n][2].abs().clamp_min(1e-30)).max()))
             for n in rt if n not in act_same and rt[n][2] is not None and ot[n][2] is not None}
    assert all(rel <= 2.0 ** -7 for _, rel in moved.values()), moved  # one step of the 16-bit mean, summed over the batches
    differing = [k for k in ref_state
                 if not torch.equal(our_state[k].cpu().reshape(-1).view(torch.uint8), ref_state[k].reshape(-1).view(torch.uint8))]
    # attribute every differing tensor: to a linear with another pick, to a linear whose activation mean moved a bit, or to a
    # tensor fused with such a linear (the LayerNorm in front of it / the sibling projections sharing its input)
    suspects = set(other) | {n for n in rt if n not in act_same}
    unexplained = []
    for k in differing:
        owner = k.rsplit(".", 1)[0]
        layer = owner.rsplit(".", 2)[0] if "layers." in owner else owner
        if not any(s == owner or s.startswith(owner + ".") or (layer and s.startswith(layer + ".")) for s in suspects):
            unexplained.append(k)
    note(f"INT4-AWQ on the device vs the reference's eager search ({arch} {str(dtype)[6:]}): {len(same_alpha)} / {len(rt)} searched linears "
         f"pick the reference's alpha; other picks (reference alpha, ours, relative gap of the two in the REFERENCE's loss table): "
         f"{ {n.replace('model.', ''): (a, b, float(f'{g:.2e}')) for n, (a, b, g) in other.items()} }; activation means bit-identical for "
         f"{len(act_same)} / {len(rt)} linears (others: channels moved, largest relative step: {moved}); {len(ref_state) - len(differing)} / {len(ref_state)} checkpoint tensors byte-identical, "
         f"{len(differing)} differing, of which {len(unexplained)} not attributable to a tie or a moved activation mean")
    assert len(same_alpha) >= 0.9 * len(rt), f"only {len(same_alpha)} of {len(rt)} linears pick the reference's alpha: {other}"
    assert all(abs(g) < 2e-3 for _, _, g in other.values()), f"a pick that is no tie in the reference's own table: {other}"
    assert not unexplained, unexplained
    span = (ref_logits.float().max() - ref_logits.float().min()).item()
    assert (our_logits.float() - ref_logits.float()).abs().max().item() <= 2e-2 * span


@pytest.mark.parametrize("arch,dtype", [("opt", torch.float16), ("llama", torch.bfloat16), ("qwen2", torch.bfloat16)])
def test_int4_awq_checkpoint_on_the_device_is_the_references_byte_for_byte(ref, arch, dtype):
    """VERDICT round 5, weak #1 closed.  OPT fp16 on the device used to end with 41 of 54 checkpoint tensors identical although
    every linear picked the reference's alpha.  tools/diag/opt_awq_device_diff.py took it apart:
      * the 13 tensors were q / k / v and the LayerNorm their shared scale is folded into, in both layers -- produced at EXPORT,
        from a quantized model whose state was the reference's: the resmooth step averaged the three pre_quant_scale vectors on
        the HOST whatever the numerics mode, and torch's GPU `mean` (fp32 sum times a rounded 1/3) and its CPU `mean` (sum
        divided by 3) round an fp16 mean differently in 3-6 of 128 channels (export.preprocess_linear_fusion now evaluates the
        reference's expression where numerics.mode() says; bf16 Llama never showed it);
      * separately, AWQ's per-channel mean |x| was summed by this library's kernel in another order than torch's reduction:
        one channel of fc2 a 16-bit step apart in each layer -- invisible in OPT's packed nibbles, 23 of 56 scale / amax
        vectors at Llama-3-8B width (numerics "device" now takes that one statistic by torch's own expression,
        model_calib.awq_lite).
    With both: every checkpoint tensor and the logits are the reference's eager run on the same device, byte for byte."""
    ref_amax, ref_state = diff._reference_run("INT4_AWQ_CFG", dtype, False, arch, None, device=DEV)
    with moa.numerics.scale_math("device"):
        our_amax, our_state = diff._our_run("INT4_AWQ_CFG", dtype, False, arch, None, device=DEV)
    ref_logits, our_logits = ref_state.pop("__logits__"), our_state.pop("__logits__")
    ref_json, our_json = ref_state.pop("__quant_json__", None), our_state.pop("__quant_json__", None)
    assert sorted(our_state) == sorted(ref_state)
    differing = [k for k in ref_state
                 if not torch.equal(our_state[k].cpu().reshape(-1).view(torch.uint8), ref_state[k].reshape(-1).view(torch.uint8))]
    note(f"INT4-AWQ on the device ({arch} {str(dtype)[6:]}): {len(ref_state) - len(differing)} / {len(ref_state)} checkpoint tensors "
         f"byte-identical to the reference's eager run")
    assert not differing, differing
    assert torch.equal(our_logits, ref_logits)
    diff._assert_same_quant_json(our_json, ref_json, f"INT4-AWQ {arch}")


def test_a_second_format_on_part_of_the_model_on_the_device_equals_the_reference(ref):
    """Per-layer overrides under max calibration (FP8 preset, INT8 per-channel MLP): both sides on cuda:0, byte for byte -- the
    per-layer table of the mixed-precision checkpoint included."""
    edit = diff._override(diff._INT8_MLP)
    ref_amax, ref_state = diff._reference_run("FP8_DEFAULT_CFG", torch.bfloat16, True, "llama", None, device=DEV, edit=edit)
    with moa.numerics.scale_math("device"):
        our_amax, our_state = diff._our_run("FP8_DEFAULT_CFG", torch.bfloat16, True, "llama", None, device=DEV, edit=edit)
    assert sorted(ref_amax) == sorted(n for n in our_amax if n in ref_amax)
    for n, a in ref_amax.items():
        assert torch.equal(our_amax[n].reshape(-1), a.reshape(-1)), f"amax of {n} differs"
    ref_json, our_json = ref_state.pop("__quant_json__", None), our_state.pop("__quant_json__", None)
    assert torch.equal(our_state.pop("__logits__"), ref_state.pop("__logits__"))
    assert sorted(our_state) == sorted(ref_state), set(our_state) ^ set(ref_state)
    diff._assert_same_quant_json(our_json, ref_json, "FP8 + INT8 MLP")
    for k, want in ref_state.items():
        got = our_state[k].detach().cpu()
        assert torch.equal(got.contiguous().reshape(-1).view(torch.uint8), want.contiguous().reshape(-1).view(torch.uint8)), k
    note(f"FP8 preset with INT8 per-channel MLP linears, both on the device: {len(ref_amax)} amax, logits, {len(ref_state)} "
         "checkpoint tensors byte-identical")


def test_awq_over_a_model_with_a_second_format_on_the_device_against_the_reference(ref):
    """INT4-AWQ with per-tensor FP8 attention linears: awq_lite smooths and searches those too (the generic route: scaled
    weight through the quantizer itself, error-GEMM engine).  Same stated tolerance as the plain INT4-AWQ test: the
    candidates are scored by GEMMs that sum in different orders, so near-ties of a random-init model may swap."""
    edit = diff._override(diff._FP8_ATTENTION)
    ref_amax, ref_state = diff._reference_run("INT4_AWQ_CFG", torch.bfloat16, False, "llama", None, device=DEV, edit=edit)
    with moa.numerics.scale_math("device"):
        our_amax, our_state = diff._our_run("INT4_AWQ_CFG", torch.bfloat16, False, "llama", None, device=DEV, edit=edit)
    ref_logits, our_logits = ref_state.pop("__logits__"), our_state.pop("__logits__")
    ref_json, our_json = ref_state.pop("__quant_json__", None), our_state.pop("__quant_json__", None)
    assert sorted(our_state) == sorted(ref_state), set(our_state) ^ set(ref_state)
    diff._assert_same_quant_json(our_json, ref_json, "INT4-AWQ + FP8 attention")
    pqs = [k for k in ref_state if k.endswith("pre_quant_scale")]
    attn = [k for k in pqs if "self_attn" in k]
    assert attn, "the FP8 attention linears carry no smoothing scale in the reference's checkpoint?"
    same = [k for k in pqs if torch.equal(our_state[k].cpu(), ref_state[k])]
    note(f"INT4-AWQ + FP8 attention on the device vs the reference: {len(same)} / {len(pqs)} pre_quant_scale vectors identical "
         f"({sum(k in same for k in attn)} / {len(attn)} of the FP8 linears')")
    # the INT4 linears under the plain INT4-AWQ test's bound; the FP8 linears' candidates differ by FP8's small rounding error
    # only, far inside the two GEMMs' summation-order noise on a random-init model: their count is reported, not bounded (on
    # the CPU tier, where both sides sum alike, every one of them equals the reference: tests/test_differential_cpu.py)
    int4 = [k for k in pqs if k not in attn]
    assert sum(k in same for k in int4) >= 0.9 * len(int4), f"only {sum(k in same for k in int4)} of {len(int4)} INT4 scale vectors equal"
    span = (ref_logits.float().max() - ref_logits.float().min()).item()
    assert (our_logits.float() - ref_logits.float()).abs().max().item() <= 2e-2 * span


def _wide_llama(outliers: bool):
    """Two decoder layers of Llama-3-8B's width (hidden 4096, MLP 14336, 32 / 8 heads), random init, small vocabulary.
    outliers: a few embedding channels and norm gains are scaled up, so activations carry massive channels like a trained
    model's (then AWQ's candidates stand apart); without them every candidate of a linear lies within a fraction of a
    percent -- the adversarial case for a search that sums the same products in another order."""
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(11)
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=2, num_attention_heads=32,
                      num_key_value_heads=8, vocab_size=1024, max_position_embeddings=1024, architectures=["LlamaForCausalLM"])
    m = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    if outliers:
        with torch.no_grad():
            g = torch.Generator().manual_seed(5)
            hot = torch.randperm(4096, generator=g)[:24]
            m.model.embed_tokens.weight[:, hot] *= 30.0
            for layer in m.model.layers:
                layer.input_layernorm.weight[hot] *= 8.0
                layer.post_attention_layernorm.weight[hot] *= 8.0
    return m.to(DEV)


@pytest.mark.parametrize("outliers", [True, False])
def test_int4_awq_at_llama_3_8b_width_picks_the_reference_alphas(ref, outliers):
    """The screen + exact re-score (search="auto") against the REFERENCE's own awq_lite at full layer width, both on the
    device (until now only this package's exhaustive engine had been the witness at Cin = 4096 /
    14336).  The reference scores with the library GEMM, so near-ties may fall the other way; with activation outliers every
    linear must pick the reference's alpha, without them at least 12 of 14."""
    import modelopt.torch.quantization as mtq

    batches = [torch.randint(0, 1024, (4, 512), generator=torch.Generator().manual_seed(90 + i)).to(DEV) for i in range(8)]

    def loop(m):
        with torch.no_grad():
            for b in batches:
                m(b)

    def scales(model, quantizer_type):
        return {n: mod.input_quantizer._pre_quant_scale.detach().float().cpu().clone() for n, mod in model.named_modules()
                if hasattr(mod, "input_quantizer") and getattr(mod.input_quantizer, "_pre_quant_scale", None) is not None}

    rcfg = copy.deepcopy(mtq.INT4_AWQ_CFG)
    rcfg["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "debug": True}  # debug: the search tables stay on the modules
    r = mtq.quantize(_wide_llama(outliers), rcfg, loop)
    want = scales(r, None)
    ref_alpha = {n: float(mod.awq_lite.best_alpha) for n, mod in r.named_modules() if hasattr(mod, "awq_lite")}
    ref_loss = {n: {round(float(a), 1): float(v) for a, v in mod.awq_lite.loss.items()} for n, mod in r.named_modules()
                if hasattr(mod, "awq_lite")}
    del r
    torch.cuda.empty_cache()
    ours = _wide_llama(outliers)
    with moa.numerics.scale_math("device"), torch.no_grad():
        moa.quantize(ours, copy.deepcopy(moa.model_quant.INT4_AWQ_CFG), loop)
    got = scales(ours, None)
    alphas = {n: float(mod.awq_lite.best_alpha) for n, mod in ours.named_modules() if hasattr(mod, "awq_lite")}
    assert sorted(got) == sorted(want) and len(want) == 14 and sorted(alphas) == sorted(ref_alpha)
    same_alpha = [n for n in ref_alpha if round(alphas[n], 1) == round(ref_alpha[n], 1)]
    same_vec = [n for n in want if torch.equal(got[n], want[n])]
    # a linear that picked another alpha: how far apart are the two candidates in the REFERENCE's own loss table?
    gaps = {}
    for n in ref_alpha:
        if n not in same_alpha:
            t = ref_loss[n]
            gaps[n.replace("model.layers.", "L")] = (round(ref_alpha[n], 1), round(alphas[n], 1),
                                                      (t[round(alphas[n], 1)] - t[round(ref_alpha[n], 1)]) / t[round(ref_alpha[n], 1)])
    # same alpha but another bit in the scale vector: the per-channel mean |x| behind it is summed in this library's own
    # (defined) order, torch's GPU reduction in another -- stated tolerance: one step of the 16-bit scale
    # same alpha => the same scale vector, bit for bit: numerics "device" takes the per-channel mean |x| behind it by torch's own
    # reduction, like the reference's run on this device (rounds 4-5: this library's kernel, another summation order, 2-3 of
    # 14 vectors one 16-bit step apart)
    worst = max([((got[n] - want[n]).abs() / want[n].abs()).max().item() for n in same_alpha if n not in same_vec] or [0.0])
    assert worst == 0.0 and len(same_vec) == len(same_alpha), (worst, len(same_vec), len(same_alpha))
    st = moa.model_calib.AWQ_LITE_STATS
    note(f"INT4-AWQ at Llama-3-8B width on the device vs the reference's eager search ({'outlier channels' if outliers else 'plain random init'}): "
         f"{len(same_alpha)} / 14 linears pick the reference's alpha ({len(same_vec)} scale vectors bit-identical, the others within {worst:.1e} relative: the activation mean's summation order); other picks "
         f"(reference alpha, ours, relative gap of the two in the reference's loss table): {gaps}; "
         f"re-scored {st.get('rescored_candidates')} candidates of {st.get('rescored_linears')} linears")
    # another pick is accepted only where the reference's own table calls the two candidates a tie to its GEMM's rounding
    assert all(abs(g[2]) < 2e-4 for g in gaps.values()), gaps
    assert len(same_alpha) >= (14 if outliers else 11), gaps


def test_random_quantizer_configurations_equal_the_references_quantizer_on_the_device(ref):
    """tools/quantizer_fuzz.py: 200 seeded random TensorQuantizer configurations (INT4 / 6 / 8 and FP8; per tensor, per axis,
    static last-axis blocks of 16-128 incl. ragged widths, dynamic blocks; signed / unsigned, narrow range; rank 2-3, three
    dtypes, magnitudes 0.02-30, zeros and outliers planted): two calibration batches, then the fake-quantized output --
    amax and output equal the REFERENCE's own TensorQuantizer on the same device bit for bit, and whatever the reference
    refuses this package refuses too."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import quantizer_fuzz

    st = quantizer_fuzz.main(200, 2025, verbose=False)
    note(f"quantizer fuzz vs the reference on the device: {st['equal']} of {st['cases']} configurations bit-equal (amax + output), "
         f"{st['both_refused']} refused by both, {st['reference_refused']} refused by the reference only, "
         f"{len(st['ours_refused'])} by this package only, {len(st['different'])} different")
    assert not st["different"], st["different"][:3]
    assert not st["ours_refused"], st["ours_refused"][:3]
    assert st["reference_refused"] == 0, st.get("reference_refusals")
    assert st["equal"] >= 150


def test_random_calls_of_the_paths_functions_equal_the_references_eager_functions_on_the_device(ref):
    """tools/ops_fuzz.py, 100 seeded random calls per family: reduce_amax over any axis subset (keepdims on / off),
    reduce_block_padding + reduce_block_amax on rank 2-4, fake_tensor_quant / scaled_e4m3 with scalar, per-axis, leading-prefix
    and apart amax, create_asp_mask on rank 1-4 with planted ties, FP8QTensor (per tensor / axis / block) and MXFP4QTensor
    quantize + dequantize -- against the reference's eager implementations on the same device tensors, bit for bit (scale
    math in numerics mode "device": the reference runs on this GPU too)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import ops_fuzz

    out = ops_fuzz.main(100, 2025, verbose=False)
    note("functions fuzz vs the reference's eager functions on the device: " +
         ", ".join(f"{fam} {st['equal']}/{st['cases']}" for fam, st in out.items()))
    for fam, st in out.items():
        assert not st["different"], (fam, st["different"][:3])
        assert not st["ours_refused"], (fam, st["ours_refused"][:3])
        assert not st["reference_refused"], (fam, st["reference_refused"])
        assert st["equal"] >= 90, (fam, st["equal"])


# ------------------------------------------------------------------------------------------------------------- A'
# S7, the algorithm seam (modelopt_algorithms.py): install(algorithms=True) + the reference's own, unmodified mtq.quantize and
# export_hf_checkpoint on device tensors.  What runs underneath is this package's fused flow (multi-tensor weight pass, one
# statistics launch per decoder layer, Gram screen + MFMA error GEMM, fused packers) on the REFERENCE's model objects; what
# the reference holds and exports afterwards must equal its own eager run on the same device, under section B's tolerances.
def _quantize_then_export(mtq, preset, dtype, with_kv, arch, stats_after_quantize=None, algorithm=None):
    """diff._reference_run on the device, with the seam counters read right after mtq.quantize returns (the logits forward
    and the export after it go through the kernel seams too, and are not the search)."""
    import tempfile

    from modelopt.torch.export import export_hf_checkpoint
    from safetensors import safe_open

    model = diff._model(dtype, arch).to(DEV)
    cfg = copy.deepcopy(getattr(mtq, preset))
    if algorithm is not None:
        cfg["algorithm"] = copy.deepcopy(algorithm)
    if with_kv:
        cfg = mtq.update_quant_cfg_with_kv_cache_quant(cfg, copy.deepcopy(mtq.FP8_KV_CFG["quant_cfg"]))
    batches = [b.to(DEV) for b in diff._batches()]
    with torch.no_grad():
        q = mtq.quantize(model, cfg, (lambda m: [m(b) for b in batches]) if cfg.get("algorithm") else None)
        if stats_after_quantize is not None:
            stats_after_quantize()
        state = {"state_dict/" + n: t.detach().cpu().clone() for n, t in q.state_dict().items()}
        logits = None if "MXFP" in preset and stats_after_quantize is None else q(batches[0]).logits.cpu().clone()
    with tempfile.TemporaryDirectory() as d:
        export_hf_checkpoint(q, export_dir=d)
        with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
            for k in f.keys():
                state[k] = f.get_tensor(k)
        state["__json__"] = open(os.path.join(d, "hf_quant_config.json")).read()
    return state, logits


ALGORITHM_SEAM_CASES = [
    ("FP8_DEFAULT_CFG", torch.bfloat16, True, "llama", "S7:max_calibrate"),
    ("FP8_DEFAULT_CFG", torch.float16, True, "mixtral", "S7:max_calibrate"),
    ("INT8_DEFAULT_CFG", torch.float32, False, "opt", "S7:max_calibrate"),
    ("INT8_SMOOTHQUANT_CFG", torch.bfloat16, False, "llama", "S7:smoothquant"),
    ("INT4_BLOCKWISE_WEIGHT_ONLY_CFG", torch.bfloat16, False, "llama", "S7:max_calibrate"),
    ("FP8_PER_CHANNEL_PER_TOKEN_CFG", torch.bfloat16, False, "llama", "S7:max_calibrate"),
]


@pytest.mark.parametrize("preset,dtype,with_kv,arch,entry", ALGORITHM_SEAM_CASES)
def test_reference_quantize_and_export_through_the_algorithm_seam_on_the_device(ref, preset, dtype, with_kv, arch, entry):
    base_state, base_logits = _quantize_then_export(ref, preset, dtype, with_kv, arch)
    seen = {}
    with installed(algorithms=True) as (plugin, got):
        assert any(g.startswith("S7:_calib_func") for g in got), got
        our_state, our_logits = _quantize_then_export(ref, preset, dtype, with_kv, arch, lambda: seen.update(plugin.STATS))
        total = dict(plugin.STATS)
    assert seen.get(entry, 0) >= 1, f"{preset}: the reference's quantize() never reached {entry}: {seen}"
    assert not [k for k in total if "fallback" in k], f"{preset}: handed back to the reference: {total}"
    assert sorted(base_state) == sorted(our_state), set(base_state) ^ set(our_state)
    for k, want in base_state.items():
        got_t = our_state[k]
        if k == "__json__":
            assert got_t == want, f"{preset}: hf_quant_config.json differs"
            continue
        assert got_t.dtype == want.dtype and got_t.shape == want.shape, k
        assert torch.equal(got_t.contiguous().reshape(-1).view(torch.uint8), want.contiguous().reshape(-1).view(torch.uint8)), f"{preset}: {k} differs"
    assert torch.equal(base_logits, our_logits), f"{preset}: logits differ"
    packers = {k: v for k, v in total.items() if k.startswith("S7:to_quantized_weight") or k.startswith("S7:pack_int4")}
    note(f"reference through the ALGORITHM seam on the device, {preset} {arch} {str(dtype)[6:]}: {len(base_state) - 1} state / checkpoint "
         f"tensors + logits + hf_quant_config.json == its eager run; calibration served by {entry} with "
         f"{sum(v for k, v in seen.items() if k.startswith('S1:') or k.startswith('S6:'))} per-tensor kernel-seam calls inside it; "
         f"export packers {packers}")


@pytest.mark.parametrize("preset,dtype,arch", [("INT8_DEFAULT_CFG", torch.bfloat16, "llama"), ("FP8_DEFAULT_CFG", torch.float16, "llama")])
def test_reference_mse_calibration_through_the_algorithm_seam_on_the_device(ref, preset, dtype, arch):
    """The reference's own `mtq.quantize(..., algorithm "mse")` with S7 installed == its eager run on the device (state, logits,
    checkpoint, JSON): the fused 39-candidate sweep evaluates the candidate grid of the reference's run on THIS device."""
    base_state, base_logits = _quantize_then_export(ref, preset, dtype, False, arch, algorithm="mse")
    seen = {}
    with installed(algorithms=True) as (plugin, got):
        our_state, our_logits = _quantize_then_export(ref, preset, dtype, False, arch, lambda: seen.update(plugin.STATS), algorithm="mse")
        total = dict(plugin.STATS)
    assert seen.get("S7:mse_calibrate", 0) >= 1, seen
    assert not [k for k in total if "fallback" in k], total
    assert sorted(base_state) == sorted(our_state)
    for k, want in base_state.items():
        got_t = our_state[k]
        if k == "__json__":
            assert got_t == want
            continue
        assert got_t.dtype == want.dtype and torch.equal(got_t.contiguous().reshape(-1).view(torch.uint8),
                                                         want.contiguous().reshape(-1).view(torch.uint8)), f"{preset} mse: {k} differs"
    assert torch.equal(base_logits, our_logits)
    note(f"reference mse calibration through the ALGORITHM seam on the device, {preset} {arch} {str(dtype)[6:]}: "
         f"{len(base_state) - 1} state / checkpoint tensors + logits == its eager run")


@pytest.mark.parametrize("preset,arch,dtype", [("INT4_AWQ_CFG", "llama", torch.bfloat16), ("INT4_AWQ_CFG", "qwen2", torch.bfloat16),
                                               ("W4A8_AWQ_BETA_CFG", "llama", torch.bfloat16)])
def test_reference_int4_awq_through_the_algorithm_seam_on_the_device(ref, preset, arch, dtype):
    """The reference's own mtq.quantize(INT4_AWQ_CFG) + export with the search served by this package's awq_lite.  Section
    B's stated tolerance (the two sides score with GEMMs that sum in different orders): >= 90 % of the linears end with the
    reference's scale vector -- those linears' packed weights and group scales are then byte-identical -- and the logits agree to
    2e-2 of their range.  And the point of the seam: NO per-tensor fake_tensor_quant_with_axis call inside the search."""
    base_state, base_logits = _quantize_then_export(ref, preset, dtype, False, arch)
    seen = {}
    with installed(algorithms=True) as (plugin, got):
        our_state, our_logits = _quantize_then_export(ref, preset, dtype, False, arch, lambda: seen.update(plugin.STATS))
        total = dict(plugin.STATS)
    assert seen.get("S7:awq", 0) == 1, seen
    assert seen.get("S1:fake_tensor_quant_with_axis", 0) == 0 and seen.get("S1:fake_tensor_quant", 0) == 0, \
        f"per-tensor kernel-seam calls inside the AWQ search: {seen}"
    assert not [k for k in total if "fallback" in k], total
    assert total.get("S7:pack_int4_in_uint8", 0) > 0, total
    assert sorted(base_state) == sorted(our_state), set(base_state) ^ set(our_state)
    assert base_state["__json__"] == our_state["__json__"]
    pqs = [k for k in base_state if k.endswith("pre_quant_scale") and not k.startswith("state_dict/")]
    same = [k for k in pqs if torch.equal(our_state[k], base_state[k])]
    tensors = [k for k in base_state if k != "__json__" and not k.startswith("state_dict/")]
    identical = sum(torch.equal(our_state[k].reshape(-1).view(torch.uint8), base_state[k].reshape(-1).view(torch.uint8)) for k in tensors)
    note(f"reference {preset} through the ALGORITHM seam on the device ({arch}): {len(same)} / {len(pqs)} pre_quant_scale vectors "
         f"identical to its eager search, {identical} / {len(tensors)} checkpoint tensors byte-identical; kernel-seam calls inside "
         f"quantize(): {sorted((k, v) for k, v in seen.items() if not k.startswith('S7'))}")
    assert len(same) >= 0.9 * len(pqs), f"only {len(same)} of {len(pqs)} scale vectors equal the reference's"
    if len(same) == len(pqs):
        assert identical == len(tensors), [k for k in tensors if not torch.equal(our_state[k].reshape(-1).view(torch.uint8),
                                                                                 base_state[k].reshape(-1).view(torch.uint8))][:5]
    span = (base_logits.float().max() - base_logits.float().min()).item()
    assert (our_logits.float() - base_logits.float()).abs().max().item() <= 2e-2 * span


def test_reference_mxfp4_through_the_algorithm_seam_equals_the_kernel_seams(ref):
    """The reference has no eager MX implementation: its MXFP4 run exists only through the kernel seams (S1).  MXFP4_DEFAULT_CFG
    has no calibration algorithm (E8M0 block scales come from every input), so what the algorithm seam adds here is the export
    packer: the reference's torch expression of MXFP4QTensor.quantize (kernel-seam run) against this package's pack kernel
    (algorithm-seam run) -- the checkpoint must not change by a byte."""
    with installed() as (plugin, _):
        base_state, base_logits = _quantize_then_export(ref, "MXFP4_DEFAULT_CFG", torch.bfloat16, False, "llama", lambda: None)
    with installed(algorithms=True) as (plugin, _):
        our_state, our_logits = _quantize_then_export(ref, "MXFP4_DEFAULT_CFG", torch.bfloat16, False, "llama", lambda: None)
        total = dict(plugin.STATS)
    assert total.get("S7:to_quantized_weight:mxfp4", 0) == 14 and not [k for k in total if "fallback" in k], total
    assert sorted(base_state) == sorted(our_state)
    for k, want in base_state.items():
        if k == "__json__":
            assert our_state[k] == want
        else:
            assert torch.equal(our_state[k].reshape(-1).view(torch.uint8), want.reshape(-1).view(torch.uint8)), k
    assert torch.equal(base_logits, our_logits)


# ------------------------------------------------------------------------------------------------------------- D
def test_reference_sparsify_on_the_device_through_the_mask_seam(ref):
    import modelopt.torch.sparsity as mts

    def run():
        m = _tiny(torch.bfloat16)
        s = mts.sparsify(m, "sparse_magnitude")
        masks = {n: mod._weight_mask.clone() for n, mod in s.named_modules() if getattr(mod, "_weight_mask", None) is not None}
        with torch.no_grad():
            return masks, s(_batches()[0]).logits.clone()

    base_masks, base_logits = run()
    with installed() as (plugin, got):
        assert "S5:create_asp_mask" in got
        our_masks, our_logits = run()
        calls = plugin.STATS.get("S5:create_asp_mask", 0)
    assert calls > 0 and len(base_masks) == calls
    assert set(base_masks) == set(our_masks)
    for n in base_masks:
        assert our_masks[n].dtype == torch.bool and torch.equal(base_masks[n], our_masks[n]), n
    assert torch.equal(base_logits, our_logits)


# ------------------------------------------------------------------------------------------------------------- C
# The reference's own GPU tests, unmodified, THREE ways in pytest subprocesses: plain (its eager / extension-less path on this
# chip), with the kernel seams installed BEFORE collection (their modules call get_cuda_ext*() at import time), and with the
# algorithm seam on top.  The staged archive holds tests/{conftest.py, _test_utils, gpu/conftest.py, gpu/torch/quantization,
# gpu/torch/export}.  Files left out need packages or topologies the box does not have (FSDP / DeepSpeed / tensor parallel ranks,
# ONNX, diffusers, vLLM, the example scripts).
REFERENCE_TEST_DIRS = ("quantization", "quantization/plugins", "export")
REFERENCE_TEST_SKIP = ("gpt_oss", "fsdp", "deepspeed", "onnx", "tp.py", "diffusers", "vllm", "torch_export", "unified_hf_export_and_check")


def reference_test_files():
    base = os.path.join(ref_shim.reference_root(), "tests", "gpu", "torch")
    return [f"{d}/{f}" for d in REFERENCE_TEST_DIRS if os.path.isdir(os.path.join(base, d)) for f in sorted(os.listdir(os.path.join(base, d)))
            if f.startswith("test_") and f.endswith(".py") and not any(s in f for s in REFERENCE_TEST_SKIP)]


def run_reference_tests(files, seams=True, timeout=1500, extra_args=(), algorithms=False):
    """pytest subprocess over the reference's own test files (paths relative to tests/gpu/torch; a bare file name means its
    quantization directory).  Returns (summary dict, per-test outcomes, raw output)."""
    root = ref_shim.reference_root()
    shim = ref_shim.install()
    tdir = os.path.join(root, "tests", "gpu", "torch", "quantization")
    if not os.path.isdir(tdir):
        pytest.skip("the staged archive holds no reference tests (re-run tools/stage_reference.sh)")
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(ROOT, "tests"), ROOT, shim, root, os.path.join(root, "tests")])
    env["MOQ_INSTALL_SEAMS"] = "1" if seams else "0"
    env["MOQ_INSTALL_ALGORITHMS"] = "1" if algorithms else "0"
    env["MOQ_REPO_ROOT"] = ROOT
    cmd = [sys.executable, "-m", "pytest", "-p", "ref_seams_plugin", "-q", "-rA", "--no-header", "--continue-on-collection-errors",
           "-p", "no:cacheprovider", "--rootdir", os.path.join(root, "tests"), "-c", os.path.join(shim, "pytest.ini"),
           *extra_args, *[os.path.join(os.path.dirname(tdir), f) if "/" in f else os.path.join(tdir, f) for f in files]]
    with open(os.path.join(shim, "pytest.ini"), "w") as f:
        f.write("[pytest]\n")
    p = subprocess.run(cmd, env=env, cwd=os.path.join(root, "tests"), capture_output=True, text=True, timeout=timeout)
    out = p.stdout + "\n" + p.stderr
    outcomes = {}
    for m in re.finditer(r"^(PASSED|FAILED|ERROR|SKIPPED|XFAIL|XPASS)\s+(?:\[\d+\]\s+)?(\S+)", out, re.M):
        outcomes[m.group(2)] = m.group(1)
    tail = out.strip().splitlines()[-1] if out.strip() else ""
    counts = {k: int(v) for v, k in re.findall(r"(\d+) (passed|failed|skipped|errors?|xfailed|xpassed)", tail)}
    return counts, outcomes, out


_REFERENCE_RUNS = {}


def reference_runs():
    """{mode: (counts, outcomes, output)} for plain / seams / s7 over every file, once per session."""
    if not _REFERENCE_RUNS:
        files = reference_test_files()
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        for mode, kw in (("plain", {"seams": False}), ("seams", {"seams": True}), ("s7", {"seams": True, "algorithms": True})):
            _REFERENCE_RUNS[mode] = run_reference_tests(files, timeout=2400, **kw)
            with open(os.path.join(ROOT, "gpurun_out", f"reference_own_gpu_tests_{mode}.txt"), "w") as f:
                f.write(_REFERENCE_RUNS[mode][2])
        _REFERENCE_RUNS["files"] = files
    return _REFERENCE_RUNS


def _still_failing(test_ids, **mode):
    """Some of the reference's tests draw unseeded inputs: a test that fails in one mode only is run once more in that mode, alone,
    before it counts."""
    if not test_ids:
        return []
    paths = [t.split("::")[0].replace("gpu/torch/", "", 1) for t in test_ids]
    _, again, _ = run_reference_tests(sorted(set(paths)), timeout=1500, **mode)
    return sorted(t for t in test_ids if again.get(t) != "PASSED")


# Reasons a reference test may fail on this box that have nothing to do with the seams (matched in the test's own failure
# text); every other failure that the plain run does not share fails this suite.
REFERENCE_TEST_FAILURE_REASONS = {
    "__nv_isnanf": "the reference's Triton NVFP4 kernels (kernels/quantization/gemm/fp4_kernel.py) call CUDA libdevice "
                   "functions the ROCm Triton backend refuses -- its own code, never reaches a seam",
    "No module named": "a package the box does not have",
}


def _failure_sections(out):
    """{test id as pytest prints it in a failure header: that failure's text}."""
    heads = [(m.start(), m.group(1)) for m in re.finditer(r"^_+ (\S.*?) _+$", out, re.M)]
    return {name: out[pos:(heads[i + 1][0] if i + 1 < len(heads) else len(out))] for i, (pos, name) in enumerate(heads)}


def test_the_references_own_gpu_tests_pass_with_the_seams_installed(ref):
    """Every test file of the reference's tests/gpu/torch/{quantization, quantization/plugins, export} that this box can collect
    (22 files), as they lie in the reference, collected AFTER modelopt_plugin.install(): `get_cuda_ext*()` hands them our
    adapters, so their CUDA-extension assertions (extension == eager with atol = 0, the literal MX vectors, INT4 pack / unpack,
    calibrators on device tensors, real quantization, export) run on this library.  Nothing that passes WITHOUT the seams may
    fail with them, and every remaining failure must be shared with the plain run or carry a known reason."""
    runs = reference_runs()
    (pc, plain, _), (counts, outcomes, out) = runs["plain"], runs["seams"]
    regressed = _still_failing(sorted(t for t, v in plain.items() if v == "PASSED" and outcomes.get(t) != "PASSED"), seams=True)
    sections = _failure_sections(out)
    by_reason, unexpected = {}, []
    for tid in sorted(k for k, v in outcomes.items() if v in ("FAILED", "ERROR")):
        text = sections.get(".".join(tid.split("::")[1:]), "")
        why = next((r for r in REFERENCE_TEST_FAILURE_REASONS if r in text), None)
        if why is not None:
            by_reason[why] = by_reason.get(why, 0) + 1
        elif plain.get(tid) in ("FAILED", "ERROR"):
            by_reason["fails without the seams too"] = by_reason.get("fails without the seams too", 0) + 1
        elif tid in regressed or plain.get(tid) != "PASSED":
            unexpected.append(tid)
    gained = sum(1 for t, v in outcomes.items() if v == "PASSED" and plain.get(t) != "PASSED")
    seam_lines = [ln for ln in out.splitlines() if ln.startswith("[seams] S")]
    note(f"the reference's own GPU tests ({len(runs['files'])} files of tests/gpu/torch, unmodified) on the MI355X: plain {pc}; with the "
         f"seams installed {counts} ({gained} tests pass only with the seams, {len(regressed)} only without); failures by reason: "
         f"{by_reason or 'none'}; seam calls: {'; '.join(ln[8:] for ln in seam_lines if 'fallback' not in ln)}")
    assert counts.get("passed", 0) >= 700, out[-3000:]
    assert not regressed, f"{len(regressed)} reference tests pass without the seams and fail with them: {regressed[:20]}\n{out[-3000:]}"
    assert not unexpected, f"{len(unexpected)} reference tests fail with the seams installed: {unexpected[:20]}\n{out[-3000:]}"


def test_the_references_own_quantize_tests_pass_with_the_algorithm_seam_installed(ref):
    """The same files collected after install(algorithms=True) -- among them the reference's high-level tests: test_quantize_cuda.py
    (`mtq.quantize` over 22 configurations -- INT8 / FP8 / W4A8 / SmoothQuant / INT4 blockwise / AWQ lite, clip, full / NVFP4
    variants / SVDQuant / local Hessian / MX formats / KV rotation / 2-D blocks / MSE with and without the FP8 scale sweep -- x
    linear, conv and conv + linear models, save / restore), test_calib_cuda.py, test_real_quantize_cuda.py (quantize -> compress),
    test_layerwise_calibrate.py, test_gptq.py, plugins/test_accelerate_gpu.py (resident against offloaded runs), the export tests.
    Every test that passes with the kernel seams alone must pass with the algorithm seam on top: adoptable models calibrate
    through this package's flows, everything else (conv weights, NVFP4 blocks, rotation, SVDQuant, the FP8 scale sweep, offloaded
    weights) is handed back to the reference's own function, and the counters say which was which."""
    runs = reference_runs()
    (base_counts, base, _), (counts, outcomes, out) = runs["seams"], runs["s7"]
    regressed = _still_failing(sorted(t for t, v in base.items() if v == "PASSED" and outcomes.get(t) != "PASSED"),
                               seams=True, algorithms=True)
    seam_lines = [ln[8:] for ln in out.splitlines() if ln.startswith("[seams] S7")]
    served = [ln for ln in seam_lines if "fallback" not in ln]
    handed_back = [ln for ln in seam_lines if "fallback" in ln]
    note(f"the reference's own GPU tests ({len(runs['files'])} files) with the ALGORITHM seam installed: {counts} (kernel seams alone: "
         f"{base_counts}); served by S7: {'; '.join(served)}; handed back: {len(handed_back)} kinds, e.g. {'; '.join(handed_back[:4])}")
    assert not regressed, f"{len(regressed)} reference tests pass with the kernel seams and fail with the algorithm seam: {regressed[:10]}\n{out[-3000:]}"
    assert counts.get("passed", 0) >= 700 and abs(counts.get("passed", 0) - base_counts.get("passed", 0)) <= 3, (counts, base_counts)
    assert any(ln.startswith("S7:max_calibrate =") for ln in served) and any(ln.startswith("S7:awq =") for ln in served), seam_lines
