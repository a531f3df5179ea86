"""The seam objects of modelopt_plugin on GPU tensors, without the reference (the GPU box has no checkout): what the
reference's call sites pass to the pybind modules (tensor_quant.py:83-111, :184-191, qtensor/int4_tensor.py:50,94), the
S3 backend entrypoint with a duck-typed quantizer, and the function seams (reduce_amax, create_asp_mask, create_sgpt_mask)
with GPU and CPU tensors."""

import types

import pytest
import torch

import _moa_import
from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import modelopt_plugin as mp  # noqa: E402
from model_optimizer_amd import ops, sparsity  # noqa: E402
from oracle import oracle  # noqa: E402

DEV = "cuda:0"


def _x(shape=(64, 256), dtype=torch.bfloat16, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * 0.7).to(dtype)


def test_int_extension_surface():
    ext = mp.IntExtension()
    x = _x()
    amax = x.abs().amax().float()
    y = ext.fake_tensor_quant(x.to(DEV), amax.to(DEV), 8, False, True)
    assert_bits_equal(y, oracle.fake_quant_int(x, amax.reshape(1), 8, False, True), "fake_tensor_quant")
    assert_bits_equal(ext.fake_tensor_quant(x.to(DEV), amax.to(DEV)), y, "defaults: 8 bits, signed, narrow")
    xi = x.to(DEV).clone()
    assert ext.fake_tensor_quant_(xi, amax.to(DEV), 4, False, False) is None  # in place, returns nothing
    assert_bits_equal(xi, oracle.fake_quant_int(x, amax.reshape(1), 4, False, False), "fake_tensor_quant_")
    am0 = x.float().abs().amax(1)
    y = ext.fake_tensor_quant_with_axis(x.to(DEV), am0.to(DEV), 0, 8, False, True)
    assert_bits_equal(y, oracle.fake_quant_int(x, am0, 8, False, True, axis_size=64, inner=256, per_axis=True), "with_axis")
    # INT4QTensor's call sites: flat input, scales [n/g, 1] = 7 / amax, block size
    flat = x.reshape(-1)
    scales = (7.0 / flat.float().view(-1, 128).abs().amax(1, keepdim=True)).to(x.dtype)
    q = ext.INT4_quantize(flat.to(DEV), scales.to(DEV), 128)
    assert q.dtype == torch.uint8 and q.numel() == flat.numel() // 2
    assert torch.equal(q.cpu(), oracle.int4_pack(flat, scales.reshape(-1), 128, rounding=1))
    d = ext.INT4_dequantize(q, scales.to(DEV), 128)
    assert_bits_equal(d, oracle.int4_unpack(q.cpu(), scales.reshape(-1), 128), "INT4_dequantize")
    # NF4 (outside the path): the two entries run the reference's own extension-less branch with its helpers (nf4_tensor.py:103-118,
    # :178-199) -- needs the reference importable; the round trip is NF4's nearest-table-value rounding
    try:
        import sys

        from conftest import GOLDEN
        sys.path.insert(0, GOLDEN)
        import ref_shim

        have_ref = ref_shim.reference_available() and bool(ref_shim.install())
    except Exception:
        have_ref = False
    if have_ref:
        xs = flat[:4096].to(DEV)
        sc = xs.float().view(-1, 64).abs().amax(1, keepdim=True).to(x.dtype)
        packed = ext.NF4_quantize(xs, sc, 64)
        assert packed.dtype == torch.uint8 and packed.numel() == xs.numel() // 2
        back = ext.NF4_dequantize(packed, sc.reshape(-1), 64)
        assert back.numel() == xs.numel() and (back.float() - xs.float()).abs().max() <= 0.2 * sc.float().max()
    with pytest.raises(RuntimeError, match="GPU"):
        ext.fake_tensor_quant(x, amax)  # a CPU tensor: the extension's "must be a GPU tensor" error


def test_fp8_and_mx_extension_surface():
    fp8, mx = mp.Fp8Extension(), mp.MxExtension()
    x = _x(seed=1)
    amax = x.abs().amax().float()
    assert_bits_equal(fp8.fake_e4m3fy(x.to(DEV), amax.to(DEV)), oracle.fake_quant_e4m3(x, amax.reshape(1)), "fake_e4m3fy")
    am1 = x.float().abs().amax(0)
    assert_bits_equal(fp8.fake_e4m3fy_with_axis(x.to(DEV), am1.to(DEV), 1),
                      oracle.fake_quant_e4m3(x, am1, axis_size=256, inner=1, per_axis=True), "fake_e4m3fy_with_axis")
    t = mx.Types
    assert (t.E4M3, t.E5M2, t.INT8, t.E0M3, t.E1M2, t.E3M0, t.E2M1, t.E3M2, t.E2M3, t.E8M0) == tuple(range(10))
    y = mx.fused_amax_convert(x.to(DEV), 32, t.E2M1, t.E8M0, None)
    assert_bits_equal(y, oracle.mx_fused_amax_convert(x, 32, "E2M1"), "fused_amax_convert mxfp4")
    y = mx.fused_amax_convert(x.to(DEV), 16, t.E2M1, t.E4M3, amax.to(DEV))
    assert_bits_equal(y, oracle.mx_fused_amax_convert(x, 16, "E2M1", "E4M3", amax), "fused_amax_convert two-level")
    assert mx.convert_to_exmy(2.4, t.E2M1) == 2.0


def _fake_tq(num_bits, amax=None, block_sizes=None, axis=None, unsigned=False, narrow=False):
    tq = types.SimpleNamespace(_num_bits=num_bits, block_sizes=block_sizes, _unsigned=unsigned, _narrow_range=narrow, _axis=axis)
    if amax is not None:
        tq._amax = amax
    return tq


def test_s3_backend_entrypoint():
    x = _x(seed=2)
    xg = x.to(DEV)
    amax = x.abs().amax().float().reshape(1)
    assert_bits_equal(mp.mi355x_backend(xg, _fake_tq((4, 3), amax.to(DEV))), oracle.fake_quant_e4m3(x, amax), "fp8 static")
    assert_bits_equal(mp.mi355x_backend(xg, _fake_tq(8, amax.to(DEV), narrow=True)), oracle.fake_quant_int(x, amax, 8, False, True),
                      "int8 static")
    assert_bits_equal(mp.mi355x_backend(xg, _fake_tq((4, 3))), oracle.fake_quant_e4m3(x, amax), "fp8, amax from the input")
    am0 = x.float().abs().amax(1)
    assert_bits_equal(mp.mi355x_backend(xg, _fake_tq(8, axis=0, narrow=True)),
                      oracle.fake_quant_int(x, am0, 8, False, True, axis_size=64, inner=256, per_axis=True), "int8 per channel, dynamic")
    xb = x.reshape(-1, 128).contiguous()
    y = mp.mi355x_backend(xb.to(DEV), _fake_tq(4, block_sizes={-1: 128}))
    wy, _ = oracle.amax_qdq_int_group(xb, 128, num_bits=4, narrow_range=False)
    assert_bits_equal(y, wy, "int4 static blocks without an amax: fused dynamic group path")


def test_function_seams_route_by_device():
    x = _x(seed=3)
    calls = []
    original = lambda input, axis=None, keepdims=True, squeeze_scalar=True: (calls.append("ref"), input.abs().max())[1]  # noqa: E731
    seam = mp._reduce_amax_seam(original)
    assert seam(x).item() == x.abs().max().item() and calls == ["ref"]          # CPU tensors: the reference's code
    got = seam(x.to(DEV))
    assert calls == ["ref"] and got.item() == x.abs().max().item()               # GPU tensors: ours
    got = seam(x.to(DEV), axis=1)
    assert_bits_equal(got.float().reshape(-1).cpu(), x.float().abs().amax(1), "axis amax through the seam")
    asp = mp._asp_mask_seam(lambda tensor, pattern: (calls.append("asp"), torch.ones_like(tensor, dtype=torch.bool))[1])
    assert asp(x, "2:4 sparsity").all() and calls[-1] == "asp"
    m = asp(x.to(DEV), "2:4 sparsity")
    assert calls.count("asp") == 1 and torch.equal(m.cpu(), oracle.mask_2to4(x))
    sg = mp._sgpt_mask_seam(lambda tensor, hessian, config: (calls.append("sgpt"), torch.ones_like(tensor, dtype=torch.bool))[1])
    w = _x((32, 128), torch.float32, 4)
    xs = _x((512, 128), torch.float32, 5)
    h = (xs.t() @ xs * (2.0 / 512)).float()
    cfg = {"pattern": "2:4 sparsity", "col_block_size": 128, "row_block_size": -1, "hessian_damp": 0.1}
    assert sg(w, h, cfg).all() and calls[-1] == "sgpt"
    mg = sg(w.to(DEV), h.to(DEV), cfg)
    assert calls.count("sgpt") == 1 and torch.equal(mg, sparsity.create_sgpt_mask(w.to(DEV), h.to(DEV), cfg))
    assert (mg.view(32, -1, 4).sum(-1) <= 2).all()


def test_sparsegpt_hessian_hook_seam():
    """SparseGPTSearcher._hook_compute_hessian replacement: 16-bit GPU activations of a linear take the MFMA
    accumulation into mod.hessian with the reference's running-average formula; anything else goes to the original."""
    calls = []
    original = classmethod(lambda cls, mod, inp, out: calls.append("ref"))
    hook = mp._sgpt_hessian_seam(original).__func__

    class FakeLinear:  # the hook looks at the type name, .hessian and .samples (sparsegpt.py:206-236)
        pass

    mod = FakeLinear()
    mod.hessian = torch.zeros(128, 128, dtype=torch.float32, device=DEV)
    mod.samples = 0
    state = sparsity.HessianState(128, DEV)
    for seed in (6, 7):
        x = _x((2, 40, 128), torch.bfloat16, seed).to(DEV)
        hook(None, mod, (x,), None)
        state.update(x)
    assert calls == [] and mod.samples == state.samples == 4
    assert torch.equal(torch.triu(mod.hessian), torch.triu(state.hessian))
    want = sum(2.0 / 4 * (x.float().reshape(-1, 128).t() @ x.float().reshape(-1, 128))
               for x in (_x((2, 40, 128), torch.bfloat16, s).to(DEV) for s in (6, 7)))
    assert ((mod.hessian - want).abs().max() <= 2e-5 * want.abs().max())
    hook(None, mod, (_x((2, 40, 128), torch.float32, 8).to(DEV),), None)   # fp32 activations: the reference's own code
    assert calls == ["ref"]
