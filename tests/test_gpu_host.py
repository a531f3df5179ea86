"""GPU tests of the host-side mirror of the reference interface: TensorQuantizer, calibrators, quantize()
with max / smoothquant / awq_lite, sparsity and the modelopt seams -- against reference-generated golden
fixtures (tests/golden/*.npz) and the oracle."""

import copy
import json

import numpy as np
import pytest
import torch

import _moa_import
from conftest import DT, assert_bits_equal

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import QuantizerAttributeConfig, TensorQuantizer, calib, model_calib, model_quant, ops  # noqa: E402
from oracle import oracle  # noqa: E402

DEV = "cuda:0"


def test_tensor_quantizer_block_matches_reference(golden):
    g = golden("tq_block")
    for k, c in g.cases.items():
        if c.get("kind") not in ("dynamic", "static"):
            continue
        dt = DT[c["dtype"]]
        x, y = g.t(f"{k}_x", dt).to(DEV), g.t(f"{k}_y", dt)
        q = TensorQuantizer(QuantizerAttributeConfig(num_bits=4, block_sizes={-1: c["g"]}))
        if c["kind"] == "static":
            q.disable_quant(); q.enable_calib()
            q(x)
            q.load_calib_amax(); q.enable_quant(); q.disable_calib()
            assert str(q._amax.dtype) == c["amax_dtype"] and list(q._amax.shape) == c["amax_shape"], \
                f"{k}: amax {q._amax.dtype} {tuple(q._amax.shape)} vs {c['amax_dtype']} {c['amax_shape']}"
            assert_bits_equal(q._amax.float().cpu(), g.t(f"{k}_amax"), f"tq static amax {k}")
        got = q(x)
        assert got.shape == y.shape
        assert_bits_equal(got, y, f"TensorQuantizer {c['kind']} {k} {c}")


def test_max_calibrator_matches_reference(golden):
    g = golden("tq_block")
    for k, c in g.cases.items():
        if c.get("kind") != "maxcal":
            continue
        cal = calib.MaxCalibrator(8, c["axis"], False)
        for b in range(3):
            cal.collect(g.t(f"{k}_b{b}", torch.bfloat16).to(DEV))
        a = cal.compute_amax()
        assert str(a.dtype) == c["out_dtype"] and list(a.shape) == c["out_shape"]
        assert_bits_equal(a.float().cpu(), g.t(f"{k}_a"), f"MaxCalibrator {k}")
    cal = calib.MaxCalibrator(8, None, False)
    x = torch.randn(4, 64, device=DEV)
    x[1, 3] = float("nan")
    cal.collect(x)
    with pytest.raises(AssertionError, match="nan"):
        cal.compute_amax()
    cal.reset()
    cal.collect(torch.randn(4, 64, device=DEV))
    with pytest.raises(RuntimeError, match="shape changed"):
        cal._axis = 1
        cal.collect(torch.randn(4, 64, device=DEV))


def test_histogram_calibrator_matches_reference(golden):
    g = golden("hist")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        cal = calib.HistogramCalibrator(8, None, False, num_bins=c["num_bins"], skip_zeros=c["skip_zeros"])
        for b in range(3):
            cal.collect(g.t(f"{k}_b{b}", dt).to(DEV))
            assert torch.equal(cal._calib_hist.cpu().float(), g.t(f"{k}_h{b}")), f"hist {k} batch {b}"
            assert torch.equal(cal._calib_bin_edges.cpu(), g.t(f"{k}_e{b}")), f"edges {k} batch {b}"
        assert torch.equal(cal.compute_amax("percentile", percentile=99.9).reshape(1), g.t(f"{k}_pct"))
        assert torch.equal(cal.compute_amax("entropy", start_bin=64).reshape(1), g.t(f"{k}_ent"))
        # "mse": what the reference's call computes (bit width in the bias slot, see calib._compute_amax_mse), pinned
        assert torch.equal(cal.compute_amax("mse", start_bin=64).reshape(1).cpu(), g.t(f"{k}_mse"))
        hist, edges = cal._calib_hist, cal._calib_bin_edges
        for tag, args in (("mseu", (hist, edges, 8, True, 1, 64)), ("mse100", (hist, edges * 100, 8, False, 1, 16)),
                          ("mse4s3", (hist, edges * 37, 4, False, 3, 16))):
            got = calib._compute_amax_mse(*args)
            assert got.device == hist.device and torch.equal(got.reshape(1).cpu(), g.t(f"{k}_{tag}")), f"{tag} {k}"
        assert cal.compute_amax("mse_qdq", start_bin=64) > 0  # the documented-intent search (not the reference's result)


@pytest.mark.parametrize("num_bits,unsigned,nb,stride,start", [(8, False, 2048, 1, 128), (4, False, 300, 1, 16),
                                                                 ((4, 3), False, 512, 3, 64), (8, True, 2048, 7, 128)])
def test_histogram_mse_threshold_is_the_candidate_loop(num_bits, unsigned, nb, stride, start):
    """calib._compute_amax_mse_qdq (the documented-intent search: "mse_qdq", and "mse" for (4, 3)) evaluates every
    candidate in one per-row QDQ launch: same pick as the loop form it replaces (one QDQ of the bin centres +
    count-weighted mean per candidate, first strict minimum), whose per-candidate errors must agree to fp32
    reduction-order noise."""
    gen = torch.Generator().manual_seed(nb + start)
    x = (torch.randn(1 << 18, generator=gen) * torch.exp(0.7 * torch.randn(1 << 18, generator=gen))).abs()
    counts = torch.histc(x, bins=nb, min=0, max=float(x.max())).to(torch.int64).to(DEV)
    edges = torch.linspace(0, float(x.max()), nb + 1)
    got = calib._compute_amax_mse_qdq(counts, edges, num_bits, unsigned, stride, start)
    if not isinstance(num_bits, int):
        assert torch.equal(got, calib._compute_amax_mse(counts, edges, num_bits, unsigned, stride, start))  # (4, 3) routes here
    e = edges.float().to(DEV)
    centers = ((e[1:] + e[:-1]) / 2).contiguous()
    c = counts.float()
    errs = []
    for i in range(start, nb, stride):
        amax = centers[i:i + 1]
        q = ops.fake_tensor_quant(centers, amax, num_bits, unsigned) if isinstance(num_bits, int) else ops.scaled_e4m3(centers, amax)
        errs.append((((q - centers) ** 2) * c).mean().item())
    errs = torch.tensor(errs, dtype=torch.float64)
    picked = ((centers - got).abs().argmin().item() - start) // stride
    assert errs[picked] <= errs.min() * (1 + 1e-5)           # the pick is a minimum of the loop form's errors ...
    if (errs <= errs.min() * (1 + 1e-5)).sum() == 1:
        assert picked == int(errs.argmin())                    # ... and THE minimum when that is unambiguous
    for search in (calib._compute_amax_mse, calib._compute_amax_mse_qdq):
        with pytest.raises(ValueError, match="no candidate"):
            search(counts, edges, num_bits, unsigned, stride, nb)
        with pytest.raises(TypeError, match="Invalid num_bits"):
            search(counts, edges, (5, 2), unsigned, stride, start)


def test_awq_weight_scale_vs_oracle_and_reference(golden):
    g = golden("awq")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        w = g.t(f"{k}_w", dt)
        got = ops.awq_weight_scale(w.to(DEV), c["g"]).cpu()
        want = g.t(f"{k}_wscale")
        ulp = want.abs() * (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10)
        assert ((got - want).abs() <= ulp).all(), f"awq_weight_scale vs reference {k}"   # 1 dtype ulp
        assert ((got - oracle.awq_weight_scale(w, c["g"])).abs() <= ulp).all()


@pytest.mark.parametrize("dn", ["f32", "bf16", "f16"])
def test_awq_weight_scale_quotients_are_ieee_divisions(dn):
    """The |w| / (group amax + tiny) quotients share their denominator inside a packet: one refined reciprocal and a
    residual-corrected product per element where that is provably exact, an IEEE division elsewhere.  With at most 16 rows
    the kernel's sum is one row-ordered fp32 chain, so the result can be restated on the host bit for bit: group magnitudes
    from 1e-36 to 1e18 (denominators below, inside and above the window), numerators down to subnormals, zeros."""
    dt = DT[dn]
    g_ = torch.Generator().manual_seed(15)
    cols, g = 1 << 15, 128
    lo, hi = (-36.0, 18.0) if dn != "f16" else (-7.0, 4.0)
    tiny = torch.finfo(dt).tiny
    for rows in (1, 16):
        mag = torch.exp(torch.empty(rows, cols // g, 1).uniform_(lo, hi, generator=g_) * 2.302585)
        rel = torch.exp(torch.empty(rows, cols // g, g).uniform_(-60.0 if dn != "f16" else -8.0, 0.0, generator=g_))
        w = (torch.randn(rows, cols // g, g, generator=g_) * mag * rel)
        w[:, :, 0] = mag[:, :, 0]  # the group's abs-max itself
        w[:, ::7, 3] = 0.0
        w = w.reshape(rows, cols).to(dt)
        got = ops.awq_weight_scale(w.to(DEV), g).cpu()
        gmax = w.abs().reshape(rows, cols // g, g).amax(dim=-1, keepdim=True)
        den = gmax + tiny                                       # storage dtype, like the reference
        q = (w.abs().reshape(rows, cols // g, g) / den).reshape(rows, cols).float()
        acc = torch.zeros(cols, dtype=torch.float32)
        for r in range(rows):
            acc = acc + q[r]
        want = (acc / float(rows)).to(dt).float()
        assert_bits_equal(got, want, f"awq_weight_scale {dn} rows={rows}")


class TinyMLP(torch.nn.Module):
    def __init__(self, w1, w2, b2):
        super().__init__()
        self.fc1 = torch.nn.Linear(w1.shape[1], w1.shape[0], bias=False)
        self.fc2 = torch.nn.Linear(w2.shape[1], w2.shape[0], bias=True)
        self.to(w1.dtype)
        with torch.no_grad():
            self.fc1.weight.copy_(w1); self.fc2.weight.copy_(w2); self.fc2.bias.copy_(b2)

    def forward(self, x):
        return self.fc2(torch.nn.functional.gelu(self.fc1(x)))


def _flow(golden, name, cfg):
    g = golden("model_flows")
    c = g.cases[name]
    dn = c["dtype"]
    dt = DT[dn]
    model = TinyMLP(g.t(f"{dn}_w1", dt), g.t(f"{dn}_w2", dt), g.t(f"{dn}_b2", dt)).to(DEV)
    batches = [g.t(f"{dn}_x{i}", dt).to(DEV) for i in range(c["n_batches"])]

    def loop(m):
        for b in batches:
            m(b)

    q = moa.quantize(model, copy.deepcopy(cfg), loop)
    return g, c, q, batches, dt


def _close(a, b, rtol, what):
    a, b = a.float().cpu().reshape(-1), b.float().cpu().reshape(-1)
    assert a.shape == b.shape, f"{what}: shape"
    err = ((a - b).abs() / b.abs().clamp_min(1e-12)).max().item()
    assert err <= rtol, f"{what}: max rel err {err:.3e} > {rtol}"


@pytest.mark.parametrize("name,cfg", [("int8_max", model_quant.INT8_DEFAULT_CFG), ("fp8_max", model_quant.FP8_DEFAULT_CFG)])
def test_quantize_max_calibration_matches_reference(golden, name, cfg):
    g, c, q, batches, dt = _flow(golden, name, cfg)
    for tname, tdtype, tshape in c["tensors"]:
        lname, rest = tname.split("_", 1)
        qn, attr = rest.rsplit("_", 1)
        t = getattr(getattr(getattr(q, lname), qn), "_" + attr)
        assert list(t.shape) == tshape and str(t.dtype) == tdtype, f"{name} {tname}: {t.dtype} {tuple(t.shape)}"
        want = g.t(f"{name}_{tname}")
        if "weight" in qn or lname == "fc1":
            assert_bits_equal(t.float().cpu().reshape(want.shape), want, f"{name} {tname}")  # exact: max of inputs
        else:
            _close(t, want, 2e-2 if dt == torch.bfloat16 else 1e-5, f"{name} {tname}")  # depends on the fc1 GEMM
    y = q(batches[0])
    _close(y, g.t(f"{name}_y", dt), 0.25 if dt == torch.bfloat16 else 1e-2, f"{name} forward")


def test_quantize_smoothquant_matches_reference(golden):
    g, c, q, batches, dt = _flow(golden, "int8_sq", model_quant.INT8_SMOOTHQUANT_CFG)
    for lname in ("fc1", "fc2"):
        lin = getattr(q, lname)
        tol = 1e-6 if lname == "fc1" else 1e-4  # fc2's statistics sit behind the fc1 GEMM (fp32 sum order)
        _close(lin.input_quantizer._pre_quant_scale, g.t(f"int8_sq_{lname}_input_quantizer_pre_quant_scale"), tol,
               f"{lname} pre_quant_scale")
        _close(lin.weight, g.t(f"int8_sq_{lname}_wfinal"), tol, f"{lname} smoothed weight")
        _close(lin.weight_quantizer._amax, g.t(f"int8_sq_{lname}_weight_quantizer_amax"), tol, f"{lname} weight amax")
        _close(lin.input_quantizer._amax, g.t(f"int8_sq_{lname}_input_quantizer_amax"), tol, f"{lname} input amax")
    _close(q(batches[0]), g.t("int8_sq_y"), 5e-2, "smoothquant forward")


@pytest.mark.parametrize("name", ["int4_awq", "int4_awq_bf16"])
def test_quantize_awq_lite_matches_reference(golden, name):
    g, c, q, batches, dt = _flow(golden, name, model_quant.INT4_AWQ_CFG)
    for lname in ("fc1", "fc2"):
        lin = getattr(q, lname)
        h = lin.awq_lite
        tol = 1e-5 if dt == torch.float32 else 2e-2
        if lname == "fc1":
            _close(h.act_scale, g.t(f"{name}_{lname}_act_scale"), 1e-6 if dt == torch.float32 else 2 ** -7,
                   f"{lname} act_scale")
        _close(h.weight_scale, g.t(f"{name}_{lname}_weight_scale"), 1e-6 if dt == torch.float32 else 2 ** -7,
               f"{lname} weight_scale")
        ref_loss = c[f"{lname}_loss"]
        for a, v in h.loss.items():
            rv = ref_loss[str(a)]
            assert abs(float(v) - rv) <= (1e-3 if dt == torch.float32 else 3e-2) * max(rv, 1e-9), \
                f"{name} {lname} loss[alpha={a}] {float(v)} vs {rv}"
        assert h.best_alpha == c[f"{lname}_best_alpha"], f"{name} {lname}: alpha {h.best_alpha} vs {c[f'{lname}_best_alpha']}"
        if lname == "fc1":
            _close(h.best_scale, g.t(f"{name}_{lname}_best_scale"), tol, f"{lname} best_scale")
            _close(lin.input_quantizer._pre_quant_scale, g.t(f"{name}_{lname}_input_quantizer_pre_quant_scale"), tol,
                   f"{lname} pre_quant_scale")
            _close(lin.weight_quantizer._amax, g.t(f"{name}_{lname}_weight_quantizer_amax"), tol, f"{lname} weight amax")
            _close(lin.weight, g.t(f"{name}_{lname}_wfinal", dt), tol * 10, f"{lname} folded weight")


def test_create_asp_mask_shapes_vs_oracle():
    sp = moa.sparsity
    w2 = torch.randn(64, 128).to(torch.bfloat16)
    assert torch.equal(sp.create_asp_mask(w2.to(DEV)).cpu(), oracle.mask_2to4(w2))
    w4 = torch.randn(16, 32, 3, 3)
    m = sp.create_asp_mask(w4.to(DEV)).cpu()
    t = w4.permute(2, 3, 0, 1).contiguous().view(-1, 32)
    want = oracle.mask_2to4(t).view(3, 3, 16, 32).permute(2, 3, 0, 1)
    assert torch.equal(m, want) and m.dtype == torch.bool and m.shape == w4.shape
    assert not sp.check_weight_size(torch.empty(12, 32)) and sp.check_weight_size(torch.empty(16, 32))
    with pytest.raises(NotImplementedError):
        sp.create_asp_mask(w2.to(DEV), "1:2 sparsity")


def test_multi_tensor_weight_calibration_in_quantize():
    """weight_only_quantize batches all per-tensor weight quantizers into one launch: same amax as one by one."""
    torch.manual_seed(0)
    model = torch.nn.Sequential(*[torch.nn.Linear(256, 256, bias=False) for _ in range(6)]).to(DEV).to(torch.bfloat16)
    q = moa.quantize(model, model_quant.FP8_DEFAULT_CFG, lambda m: m(torch.randn(8, 256, device=DEV).to(torch.bfloat16)))
    for lin in q:
        assert lin.weight_quantizer._amax.item() == lin.weight.abs().max().item()


@pytest.mark.parametrize("name", ["w4a8_f32", "w4a8_bf16"])
def test_sequential_quantizer_w4a8_matches_reference(golden, name):
    """SequentialQuantizer (INT4 g128 blocks, then FP8 per tensor) on the weights + FP8 inputs, max calibration:
    per-stage amax and the chained fake-quantized weight equal the reference's (weights: bit-exact)."""
    g = golden("w4a8")
    c = g.cases[name]
    dt = getattr(torch, c["dtype"])
    model = TinyMLP(g.t(f"{name}_w1", dt), g.t(f"{name}_w2", dt), g.t(f"{name}_b2", dt)).to(DEV)
    batches = [g.t(f"{name}_x{i}", dt).to(DEV) for i in range(c["n_batches"])]
    q = moa.quantize(model, copy.deepcopy(model_quant.W4A8_MAX_CFG), lambda m: [m(b) for b in batches])
    for lname in ("fc1", "fc2"):
        lin = getattr(q, lname)
        wq = lin.weight_quantizer
        assert isinstance(wq, moa.tensor_quantizer.SequentialQuantizer) and len(wq) == c[f"{lname}_n_stages"]
        for i, st in enumerate(wq):
            want = g.t(f"{name}_{lname}_w{i}_amax")
            assert list(st._amax.shape) == c[f"{lname}_w{i}_amax_shape"]
            assert_bits_equal(st._amax.float().cpu().reshape(want.shape), want, f"{name} {lname} stage {i} amax")
        assert_bits_equal(wq(lin.weight).cpu(), g.t(f"{name}_{lname}_wq", dt), f"{name} {lname} chained QDQ")
        tol = 1e-6 if lname == "fc1" else (1e-4 if dt == torch.float32 else 2e-2)
        _close(lin.input_quantizer._amax, g.t(f"{name}_{lname}_in_amax"), tol, f"{name} {lname} input amax")
    _close(q(batches[0]), g.t(f"{name}_y", dt), 0.3 if dt == torch.bfloat16 else 2e-2, f"{name} forward")


def test_tensor_quantizer_2d_blocks_match_reference(golden):
    """block_sizes on both axes (FP8 128 x 128 tiles / INT8 64 x 32 tiles): calibrated amax (shape, dtype, values) and
    the fake-quant output equal the reference's run; kernel modes agree with the oracle."""
    g = golden("block2d")
    for k, c in g.cases.items():
        if "lead" in c:
            continue  # tiles of tensors of rank > 2: test_tensor_quantizer_tiles_on_the_last_two_axes_of_any_rank
        dt = DT[c["dtype"]]
        x = g.t(f"{k}_x", dt).to(DEV)
        nb = tuple(c["num_bits"]) if isinstance(c["num_bits"], list) else c["num_bits"]
        q = TensorQuantizer(QuantizerAttributeConfig(num_bits=nb, block_sizes={-1: c["bc"], -2: c["br"]}))
        q.disable_quant(); q.enable_calib()
        q(x)
        q.load_calib_amax()
        q.enable_quant(); q.disable_calib()
        y = q(x)
        assert list(q._amax.shape) == c["amax_shape"] and str(q._amax.dtype).split(".")[-1] == c["amax_dtype"]
        assert_bits_equal(q._amax.float().cpu(), g.t(f"{k}_amax").reshape(q._amax.shape), f"{k} amax")
        assert_bits_equal(y.cpu(), g.t(f"{k}_y", dt), f"{k} 2-D block fake quant")
        # fused mode (amax + QDQ in one read) and running-max accumulation against the oracle
        fp8 = isinstance(nb, tuple)
        x4 = x.reshape(x.shape[0] // c["br"], c["br"], x.shape[1] // c["bc"], c["bc"])
        y2, am2 = ops.block2d(x4, 2, fp8=fp8, num_bits=8 if fp8 else nb)
        wy, wam = oracle.block2d(x.cpu(), c["br"], c["bc"], 2, fp8=fp8, num_bits=8 if fp8 else nb)
        assert torch.equal(am2.cpu().reshape(-1), wam.reshape(-1))
        assert_bits_equal(y2.reshape(x.shape).cpu(), wy, f"{k} fused block2d")
        am3 = ops.block2d((x4 * 0.5).contiguous(), 0)
        ops.block2d(x4, 0, amax=am3, accumulate=True)
        assert torch.equal(am3.cpu().reshape(-1), wam.reshape(-1))


def test_two_level_block_format_flow_and_dynamic_type():
    """NVFP4-style quantizers (E2M1 blocks of 16, E4M3 block scales): the block scales are dynamic but the tensor-wide
    amax is max-calibrated and becomes the global amax of the two-level scale (tensor_quantizer.py:890-920).
    A top-level `type: dynamic` quantizer is never calibrated and takes the amax of each input."""
    torch.manual_seed(2)
    model = torch.nn.Sequential(torch.nn.Linear(64, 96, bias=False), torch.nn.Linear(96, 32, bias=False)).to(torch.bfloat16).to(DEV)
    batches = [torch.randn(40, 64, device=DEV, dtype=torch.bfloat16) * (1 + i) for i in range(3)]
    ref_w = [m.weight.detach().clone() for m in model]
    moa.quantize(model, moa.model_quant.NVFP4_DEFAULT_CFG, lambda m: [m(b) for b in batches])
    lin0 = model[0]
    assert lin0.weight_quantizer._block_dynamic and not lin0.weight_quantizer._dynamic and not lin0.weight_quantizer.is_mx_format
    assert lin0.weight_quantizer.amax.float().item() == ref_w[0].abs().max().float().item()
    assert lin0.input_quantizer.amax.float().item() == max(b.abs().max().float().item() for b in batches)
    x = batches[0]
    got = lin0.input_quantizer(x)
    want = oracle.mx_fused_amax_convert(x.cpu(), 16, "E2M1", "E4M3", lin0.input_quantizer.amax.float().cpu())
    assert_bits_equal(got, want, "input fake-quant with the calibrated global amax")
    gw = lin0.weight_quantizer(lin0.weight)
    assert_bits_equal(gw, oracle.mx_fused_amax_convert(ref_w[0].cpu(), 16, "E2M1", "E4M3", ref_w[0].abs().max().float().cpu()),
                      "weight fake-quant")
    # top-level dynamic: no calibration state, amax from the input at hand
    q = moa.TensorQuantizer(moa.QuantizerAttributeConfig(num_bits=(4, 3), axis=None, type="dynamic"))
    q.enable_calib()
    assert not q._if_calib
    for b in batches:
        assert_bits_equal(q(b), ops.scaled_e4m3(b, ops.reduce_amax(b)), "dynamic per-tensor FP8")
    assert q.amax is None


def test_library_ops_on_gpu_and_under_fake_tensor_tracing():
    from model_optimizer_amd import library_ops as lo
    x = (torch.randn(8, 128, device=DEV) * 3).to(torch.bfloat16)
    amax = x.abs().amax().float()
    assert_bits_equal(lo.quantize_op(x, amax, 8, 4, False, False), ops.scaled_e4m3(x, amax), "fp8 through the library op")
    assert_bits_equal(lo.quantize_op(x, amax, 4, 0, False, False), ops.fake_tensor_quant(x, amax, 4, False, False), "int4")
    assert_bits_equal(lo.dynamic_block_quantize_op(x, 32, None, 4, 2, 9, 8), ops.fused_amax_convert(x, 32, "E2M1"), "mxfp4")
    assert_bits_equal(lo.dynamic_block_quantize_op(x, 16, amax, 4, 2, 8, 4),
                      ops.fused_amax_convert(x, 16, "E2M1", "E4M3", amax), "two-level fp4")
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode(allow_non_fake_inputs=True) as mode:
        fx = mode.from_tensor(x)
        fy = torch.ops.moquant.quantize_op(fx, mode.from_tensor(amax), 8, 4, False, False)
        assert fy.shape == x.shape and fy.dtype == x.dtype


@pytest.mark.parametrize("dense", [False, True], ids=["with_empty_bins", "no_empty_bins"])
@pytest.mark.parametrize("num_bits,unsigned,nb,stride,start", [(8, False, 2048, 1, 128), (8, True, 2048, 3, 128),
                                                                 (4, False, 777, 1, 16), (6, False, 1500, 7, 100),
                                                                 (8, False, 4096, 1, 128), (8, False, 128, 1, 128)])
def test_entropy_search_on_the_device_equals_the_reference_loop(num_bits, unsigned, nb, stride, start, dense):
    """ops.hist_entropy_divergences (one workgroup per clipping candidate): every divergence agrees with the numpy
    restatement of the reference's loop to 1e-11 relative, the chosen amax is that loop's bit for bit -- zeros, gaps wider
    than a bucket, a heavy tail, grown ranges (4096 bins) and the single-candidate edge included."""
    rng = np.random.default_rng(nb + num_bits + int(dense))
    hist = (rng.exponential(1.0, nb) * 1e6 * np.exp(-np.arange(nb) / (nb / 6))).astype(np.int64)
    if dense:
        hist += 1
    else:
        hist[rng.integers(0, nb, nb // 10)] = 0
        hist[nb // 2: nb // 2 + 40] = 0
    hist[-1] = 12345
    edges = np.linspace(0, 3.0, nb + 1, dtype=np.float32)
    want_div = []
    want = calib._compute_amax_entropy(hist, edges, num_bits, unsigned, stride, start, divergences_out=want_div)
    want_div = np.array(want_div)
    got_div = ops.hist_entropy_divergences(torch.from_numpy(hist).to(DEV), 1 << (num_bits - 1 + int(unsigned)), start,
                                           stride).cpu().numpy()
    assert got_div.shape == want_div.shape
    fin = np.isfinite(want_div)
    assert np.array_equal(fin, np.isfinite(got_div))
    assert np.allclose(got_div[fin], want_div[fin], rtol=1e-11, atol=1e-14)
    cal = calib.HistogramCalibrator(num_bits, None, unsigned, num_bins=nb)
    cal._calib_hist = torch.from_numpy(hist).to(DEV)
    cal._calib_bin_edges = torch.from_numpy(edges)  # (host edges, device counts: what collect() leaves)
    ticket = cal.begin_amax("entropy", stride=stride, start_bin=start)
    assert ticket[0] == "entropy", "the search must run on the device"
    got = cal.finish_amax(ticket)
    assert got.dtype == torch.float32 and got.dim() == 0 and got.item() == want.item()
    pct = cal.begin_amax("percentile", percentile=99.9)
    assert pct[0] == "percentile"
    assert cal.finish_amax(pct).item() == calib._compute_amax_percentile(hist, edges, 99.9).item()


def test_percentile_search_on_the_device_is_numpy_bit_for_bit():
    """ops.hist_percentile_index = np.searchsorted(np.cumsum(h / h.sum()), q) per row (sequential fp64 running sums):
    int32 and int64 counts, ragged row counts (not a multiple of the 64-row workgroup), bins that are not a multiple of the
    tile, the extremes q = 0 and q = 1, an all-zero row."""
    rng = np.random.default_rng(5)
    for rows, bins, dtype in [(1, 2048, np.int64), (130, 512, np.int32), (77, 1000, np.int32), (64, 2048, np.int64)]:
        h = (rng.exponential(1.0, (rows, bins)) * 1e4 * np.exp(-np.arange(bins) / (bins / 5))).astype(dtype)
        h[:, rng.integers(0, bins, bins // 7)] = 0
        if rows > 2:
            h[2] = 0
        for pct in (99.99, 99.0, 50.0, 0.0, 100.0):
            q = pct / 100
            with np.errstate(invalid="ignore", divide="ignore"):
                cdf = np.cumsum(h.astype(np.int64) / h.astype(np.int64).sum(axis=1, keepdims=True), axis=1)
            want = np.array([np.searchsorted(cdf[r], q) for r in range(rows)])
            got = ops.hist_percentile_index(torch.from_numpy(h).to(DEV), q).cpu().numpy()
            assert np.array_equal(got, want), (rows, bins, dtype, pct, np.flatnonzero(got != want)[:5])


@pytest.mark.parametrize("name", ["lh_f32", "lh_bf16"])
def test_local_hessian_calibrate_on_the_gpu_equals_the_reference_run(golden, name):
    """local_hessian_calibrate (model_calib.py:1005-1127): per-block input Hessians from a forward with the weight quantizers
    off, then the amax multiplier search with the Hessian-weighted error -- the refined amax against the reference run."""
    g = golden("local_hessian")
    dt = {"float32": torch.float32, "bfloat16": torch.bfloat16}[g.cases[name]["dtype"]]

    class TinyMLP(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = torch.nn.Linear(128, 128, bias=False)
            self.fc2 = torch.nn.Linear(128, 128, bias=True)

        def forward(self, x):
            return self.fc2(torch.nn.functional.gelu(self.fc1(x)))

    model = TinyMLP().to(dt)
    model.fc1.weight.data.copy_(g.t(f"{name}_w1", dt))
    model.fc2.weight.data.copy_(g.t(f"{name}_w2", dt))
    model.fc2.bias.data.copy_(g.t(f"{name}_b2", dt))
    model = model.to(DEV)
    batches = [g.t(f"{name}_x{i}", dt).to(DEV) for i in range(g.cases[name]["n_batches"])]
    cfg = copy.deepcopy(model_quant.INT4_BLOCKWISE_WEIGHT_ONLY_CFG)
    cfg["quant_cfg"]["*weight_quantizer"] = {"num_bits": 4, "block_sizes": {-1: 16, "type": "static"}, "enable": True}
    cfg["algorithm"] = {"method": "local_hessian", "fp8_scale_sweep": False, "block_size": 16}
    model_quant.quantize(model, cfg, lambda m: [m(b) for b in batches])
    for lname in ("fc1", "fc2"):
        got = getattr(model, lname).weight_quantizer._amax.float().reshape(-1).cpu()
        want = g.t(f"{name}_local_hessian_{lname}_amax").reshape(-1)
        same = (got == want).float().mean().item()
        assert same >= 0.97, f"{lname}: {same:.4f} of the amax entries equal the reference's"



def test_affine_offset_of_the_kv_cache_presets_matches_the_reference_run(golden):
    """TensorQuantizer(bias=...) (tensor_quantizer.py:389-503, calib/bias.py) against tests/golden/affine_bias.npz: static
    and dynamic offsets, mean / max_min, per tensor / per head and channel, FP8 and INT8.  The offset is a torch reduction
    (its last bit follows the device's summation order: compared to a few ulps on the GPU, bit for bit on the host tier);
    FROM the reference's offset and amax the fake-quantized tensor is the reference's bit for bit."""
    g = golden("affine_bias")
    exact = not str(DEV).startswith("cuda")
    for key, c in g.cases.items():
        dt = DT[c["dtype"]]
        xs = [g.t(f"{c['dtype']}_x{i}", dt).to(DEV) for i in range(3)]
        nb = tuple(c["num_bits"]) if isinstance(c["num_bits"], list) else c["num_bits"]
        bias = {(int(k) if k.lstrip("-").isdigit() else k): v for k, v in c["bias"].items()}
        q = TensorQuantizer(QuantizerAttributeConfig(num_bits=nb, axis=None, bias=bias))
        q.disable_quant(); q.enable_calib()
        for x in xs:
            out = q(x)  # statistics only: the tensor passes through
            assert out is x or torch.equal(out, x)
        q.load_calib_amax()
        if c["static"]:
            q.load_calib_bias()
            assert list(q._bias_value.shape) == c["bias_shape"] and q._bias_value.dtype == dt, key
        q.enable_quant(); q.disable_calib()
        want_amax, want_y = g.t(f"{key}_amax"), g.t(f"{key}_y0", dt)
        ulp = 2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -22
        if c["static"]:
            want_bias = g.t(f"{key}_bias", dt)
            if exact:
                assert_bits_equal(q._bias_value, want_bias, f"{key} offset")
                assert_bits_equal(q._amax.float().cpu(), want_amax, f"{key} amax")
            else:
                scale = float(want_bias.float().abs().max())
                assert torch.allclose(q._bias_value.float().cpu(), want_bias.float(), rtol=0, atol=4 * ulp * scale), key
                assert torch.allclose(q._amax.float().cpu(), want_amax, rtol=8 * ulp, atol=0), key
            q.bias_value = want_bias.to(DEV)  # from the reference's statistics the output is the reference's
            q.amax = want_amax.to(q._amax.dtype).to(DEV)
            assert_bits_equal(q(xs[0]), want_y, f"{key} output")
            assert "_bias_value" in q.state_dict()
        else:
            assert_bits_equal(q._amax.float().cpu(), want_amax, f"{key} amax (of the tensor as it came)")
            got = q(xs[0])
            if exact:
                assert_bits_equal(got, want_y, f"{key} output")
            else:  # the per-call offset carries the device's summation order: within one quantization step
                step = float(want_amax.max()) / (127.0 if nb == 8 else 8.0)
                assert float((got.float().cpu() - want_y.float()).abs().max()) <= step, key
        q.reset_amax()
        assert q.bias_value is None and q.amax is None and q.bias_calibrator.compute_bias() is None


def test_tensor_quantizer_tiles_on_the_last_two_axes_of_any_rank(golden):
    """block_sizes on the last two axes of a rank-3 / rank-4 tensor (stacked experts [E, Cout, Cin], leading batch dims;
    tensor_quantizer.py:1018-1043): the matrix dims are zero-padded to whole tiles, the leading dims fold into the tile
    rows of the 2-D kernel, the amax buffer keeps the reference's (L..., R/br, 1, C/bc, 1) shape.  Grids on OTHER axis sets
    (rows only, a conv weight's input channels, three axes at once) take one permuted copy to the last-axis layout.
    Calibrated amax (running maximum over two calls for the general grids), output and the dynamic-amax output equal the
    reference's run."""
    g = golden("block2d")
    seen = 0
    for k, c in g.cases.items():
        if "lead" not in c:
            continue
        seen += 1
        dt = DT[c["dtype"]]
        x = g.t(f"{k}_x", dt).reshape(c["shape"]).to(DEV)
        nb = tuple(c["num_bits"]) if isinstance(c["num_bits"], list) else c["num_bits"]
        grid = {int(a): b for a, b in c["grid"].items()} if "grid" in c else {-1: c["bc"], -2: c["br"]}
        cfg = QuantizerAttributeConfig(num_bits=nb, block_sizes=grid)
        q = TensorQuantizer(cfg)
        q.disable_quant(); q.enable_calib()
        q(x)
        if "grid" in c:
            q(x * 0.5)
        q.load_calib_amax()
        q.enable_quant(); q.disable_calib()
        assert list(q._amax.shape) == c["amax_shape"] and str(q._amax.dtype).split(".")[-1] == c["amax_dtype"], k
        assert_bits_equal(q._amax.float().cpu(), g.t(f"{k}_amax").reshape(q._amax.shape), f"{k} amax")
        y = q(x)
        assert y.shape == x.shape
        assert_bits_equal(y.cpu(), g.t(f"{k}_y", dt).reshape(c["shape"]), f"{k} grid {grid} of a rank-{len(c['shape'])} tensor")
        assert_bits_equal(TensorQuantizer(cfg)(x).cpu(), g.t(f"{k}_ydyn", dt).reshape(c["shape"]), f"{k} dynamic amax")
        state = {kk: v.clone() for kk, v in q.state_dict().items()}
        q2 = TensorQuantizer(cfg)
        q2.amax = state["_amax"]  # a restored buffer in the reference's shape serves the kernels' folded view
        assert_bits_equal(q2(x).cpu(), y.cpu(), f"{k} from a restored amax")
        # the PRODUCT path: max_calibrate -> finish_stats_collection (every quantize() flow) leaves the same buffer shape
        # and values as load_calib_amax does (advisor, round 4: it kept the kernel's folded shape)
        q3 = TensorQuantizer(cfg)
        model_calib.max_calibrate(q3, (lambda qq: (qq(x), qq(x * 0.5))) if "grid" in c else (lambda qq: qq(x)), distributed_sync=False)
        assert list(q3._amax.shape) == c["amax_shape"], f"{k}: max_calibrate left amax {list(q3._amax.shape)}"
        assert_bits_equal(q3._amax.float().cpu(), q._amax.float().cpu(), f"{k} amax through max_calibrate")
        assert_bits_equal(q3(x).cpu(), y.cpu(), f"{k} output after max_calibrate")
    assert seen == 9 + 15


def test_one_statistics_launch_per_decoder_layer_gives_the_per_call_statistics():
    """max_calibrate(defer_stats=...) on a tiny Hugging Face Llama with FP8 weights, inputs and KV cache: the running maxima
    taken by ONE moq_mt_amax_running sweep per decoder layer (calib.DeferredAmax: q / k / v and gate / up share their
    tensor) equal the per-call launches bit for bit; a module that writes to its linear's input before the layer ends is an
    error, not a wrong amax."""
    import transformers as tf

    from model_optimizer_amd import calib as calib_mod

    torch.manual_seed(0)
    cfg = tf.LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                         vocab_size=64, max_position_embeddings=64, architectures=["LlamaForCausalLM"])
    base = tf.LlamaForCausalLM(cfg).to(torch.bfloat16).eval().to(DEV)
    batches = [torch.randint(0, 64, (4, 32), generator=torch.Generator().manual_seed(i)).to(DEV) for i in range(3)]
    qcfg = model_quant.update_quant_cfg_with_kv_cache_quant(copy.deepcopy(model_quant.FP8_DEFAULT_CFG), model_quant.FP8_KV_CFG["quant_cfg"])

    def run(defer):
        m = copy.deepcopy(base)
        c = copy.deepcopy(qcfg)
        c["algorithm"] = {"method": "max", "defer_stats": defer}
        with torch.no_grad():
            model_quant.quantize(m, c, lambda mm: [mm(b) for b in batches])
        amax = {n: q._amax.detach().float().cpu().clone() for n, q in m.named_modules()
                if isinstance(q, TensorQuantizer) and q.is_enabled and getattr(q, "_amax", None) is not None}
        return amax, dict(model_calib.MAX_CALIBRATE_STATS)

    per_call, st0 = run(False)
    per_layer, st1 = run(True)  # asked for: deferred from the first request that can be
    # (one flush per decoder layer and batch from the second batch on: a calibrator's first collect takes the general path)
    assert "deferred_stats" not in st0 and st1["deferred_stats"]["flushes"] >= 3 * (len(batches) - 1)
    d = st1["deferred_stats"]
    # 9 requests per layer-call (7 linear inputs + key / value states; the first call of every calibrator takes the general
    # path), 6 distinct tensors; far fewer table builds than flushes (the allocator repeats its addresses)
    assert d["requests"] >= 9 * 3 * (len(batches) - 1) and d["tensors"] * 9 <= d["requests"] * 6 + 6
    # automatic for a Hugging Face decoder stack on the GPU -- after one WATCHED pass (round 6: the second batch, the first one
    # in which calibrators ask): deferred from the third batch on, same statistics
    auto, st2 = run(None)
    assert st2["deferred_stats"]["flushes"] >= 3 * (len(batches) - 2) and not st2["deferred_stats"].get("disabled_by_inplace_write")
    assert sorted(auto) == sorted(per_call) and all(torch.equal(per_call[n], auto[n]) for n in per_call)
    assert len(per_call) == len(per_layer) and len(per_call) >= 3 * 9 + 3 * 7
    for n, a in per_call.items():
        assert torch.equal(a, per_layer[n]), n
    assert calib_mod.DeferredAmax.current is None

    class Scribbler(torch.nn.Module):  # writes to the activation AFTER the linear (and its input quantizer) have read it
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def forward(self, x):
            y = self.inner(x)
            x.mul_(0.5)
            return y

    m = copy.deepcopy(base)
    m.model.layers[1].mlp.down_proj = Scribbler(m.model.layers[1].mlp.down_proj)
    c = copy.deepcopy(qcfg)
    c["algorithm"] = {"method": "max", "defer_stats": True}
    with pytest.raises(RuntimeError, match="written in place"), torch.no_grad():
        model_quant.quantize(m, c, lambda mm: [mm(b) for b in batches])
    assert calib_mod.DeferredAmax.current is None
    # the AUTOMATIC choice (nobody vouched for the model) sees the write in its first, watch-only pass and backs off: no error,
    # no deferred launch, and the statistics of the per-call run of the same (scribbling) model
    def scribbled(defer):
        m = copy.deepcopy(base)
        m.model.layers[1].mlp.down_proj = Scribbler(m.model.layers[1].mlp.down_proj)
        c = copy.deepcopy(qcfg)
        c["algorithm"] = {"method": "max", "defer_stats": defer}
        with torch.no_grad():
            model_quant.quantize(m, c, lambda mm: [mm(b) for b in batches])
        return ({n: q._amax.detach().float().cpu().clone() for n, q in m.named_modules()
                 if isinstance(q, TensorQuantizer) and q.is_enabled and getattr(q, "_amax", None) is not None},
                dict(model_calib.MAX_CALIBRATE_STATS))

    plain, _ = scribbled(False)
    auto, st = scribbled(None)
    assert st["deferred_stats"].get("disabled_by_inplace_write") and st["deferred_stats"]["flushes"] == 0
    assert sorted(plain) == sorted(auto) and all(torch.equal(plain[n], auto[n]) for n in plain)
