"""MXFP8 real quantisation on the GPU (a15 QTensor family, (f)-1 checkpoint formats): `MXFP8QTensor` -- E8M0 scale
bytes from the block abs-max kernel, E4M3 bytes from the tile pack kernel with a 1 x 32 tile -- against the reference
run on CPU (tests/golden/mxfp8.npz: 2-D / 3-D / ragged shapes, all-zero blocks, block maxima on the 448 * 2^k boundary,
exponent clamps) and the size-independent properties of the format."""

import pytest
import torch

import _moa_import
from conftest import DT, assert_bits_equal

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import qtensor  # noqa: E402

DEV = "cuda:0"


def test_mxfp8_qtensor_matches_reference_run(golden):
    g = golden("mxfp8")
    assert len(g.cases) == 12
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        x = g.t(f"{k}_x", dt).to(DEV)
        qt, e8 = qtensor.MXFP8QTensor.quantize(x)
        want_e8 = torch.from_numpy(g.raw(f"{k}_e8m0").copy())
        assert e8.dtype == torch.uint8 and tuple(e8.shape) == tuple(want_e8.shape), k
        assert torch.equal(e8.cpu(), want_e8), f"{k}: {(e8.cpu() != want_e8).sum().item()} scale bytes differ"
        got = qt._quantized_data.view(torch.uint8).cpu()
        want = torch.from_numpy(g.raw(f"{k}_q").copy())
        assert got.shape == want.shape and torch.equal(got, want), f"{k}: {(got != want).sum().item()} bytes differ"
        assert_bits_equal(qt.dequantize(scale=e8).cpu(), g.t(f"{k}_deq", dt), f"{k} dequant")
        if c["kind"] != "ragged":  # the export path: scale first, then quantize_with_scale
            wsf = qtensor.MXFP8QTensor.get_weights_scaling_factor(x)
            assert torch.equal(wsf.cpu(), want_e8)
            q2 = qtensor.MXFP8QTensor.quantize_with_scale(x, wsf)
            assert q2.dtype == torch.float8_e4m3fn and torch.equal(q2.view(torch.uint8).cpu(), want)


@pytest.mark.parametrize("dn", ["bf16", "f16"])
def test_mxfp8_properties_at_weight_size(dn):
    """4096 x 4096: quantisation is idempotent on its own output, scaling a tensor by 2^k shifts every scale byte by k
    and leaves the element bytes alone, no block saturates (|q| <= 448, no NaN bytes)."""
    dt = DT[dn]
    x = (torch.randn(4096, 4096, generator=torch.Generator().manual_seed(5)) * 0.02).to(dt).to(DEV)
    qt, e8 = qtensor.MXFP8QTensor.quantize(x)
    q = qt._quantized_data.view(torch.uint8)
    assert not ((q & 0x7F) == 0x7F).any()
    deq = qt.dequantize(scale=e8)
    qt2, e82 = qtensor.MXFP8QTensor.quantize(deq)
    assert_bits_equal(qt2.dequantize(scale=e82), deq, "idempotence")
    qt3, e83 = qtensor.MXFP8QTensor.quantize(x * 4)
    assert torch.equal(e83.int(), e8.int() + 2) and torch.equal(qt3._quantized_data.view(torch.uint8), q)
    # the block maximum lands in the top binade of E4M3, [224, 448] (or the block is zero)
    top = deq.float().abs().view(-1, 32).amax(1) / torch.exp2(e8.float().view(-1) - 127)
    assert ((top >= 224) | (top == 0)).all() and (top <= 448).all()


def test_mxfp8_rejects_wrong_scale_dtype_and_block():
    x = torch.zeros(4, 64, dtype=torch.bfloat16, device=DEV)
    with pytest.raises(AssertionError):
        qtensor.MXFP8QTensor.quantize_with_scale(x, torch.zeros(4, 2, device=DEV))
    with pytest.raises(AssertionError):
        qtensor.MXFP8QTensor.get_weights_scaling_factor(torch.zeros(4, 40, dtype=torch.bfloat16, device=DEV))


def test_mxfp8_preset_exports_e4m3_weights_with_e8m0_scales():
    """MXFP8_DEFAULT_CFG (no calibration: dynamic blocks) -> checkpoint: float8_e4m3fn weights, uint8 [Cout, Cin / 32]
    scales, no input scale, quant_algo MXFP8 (unified_export_hf.py:671-679)."""
    torch.manual_seed(3)
    model = torch.nn.Sequential(torch.nn.Linear(128, 64, bias=False), torch.nn.ReLU(), torch.nn.Linear(64, 96)).to(torch.bfloat16).to(DEV)
    ref = [m.weight.detach().clone() for m in model if isinstance(m, torch.nn.Linear)]
    moa.quantize(model, moa.model_quant.MXFP8_DEFAULT_CFG, None)
    assert moa.export.get_quantization_format(model[0]) == "mxfp8"
    state = moa.export.export_state_dict(model, torch.bfloat16)
    assert moa.export.hf_quant_config(model)["quantization"]["quant_algo"] == "MXFP8"
    assert sorted(state) == ["0.weight", "0.weight_scale", "2.bias", "2.weight", "2.weight_scale"]
    for i, w in zip((0, 2), ref):
        qt, e8 = qtensor.MXFP8QTensor.quantize(w)
        assert state[f"{i}.weight"].dtype == torch.float8_e4m3fn and state[f"{i}.weight_scale"].dtype == torch.uint8
        assert tuple(state[f"{i}.weight_scale"].shape) == (w.shape[0], w.shape[1] // 32)
        assert torch.equal(state[f"{i}.weight"].view(torch.uint8), qt._quantized_data.view(torch.uint8))
        assert torch.equal(state[f"{i}.weight_scale"], e8)
        # the fake-quantized forward uses the same numbers: QDQ(w) == dequantised checkpoint weight
        fq = model[i].weight_quantizer(w)
        assert_bits_equal(fq, qt.dequantize(scale=e8), f"linear {i}: fake quant vs checkpoint")
