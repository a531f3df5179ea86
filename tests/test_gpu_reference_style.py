"""The reference's own tensor-quant tests, restated against this library's entries (same inputs, same expectations, same
tolerances): tests/_test_utils/torch/quantization/tensor_quant_common.py:37-136 (TensorQuantCommon /
FakeTensorQuantTester), tests/_test_utils/torch/quantization/quant_utils.py:21-32 (the `quant()` formula) and
tests/gpu/torch/quantization/test_tensor_quant_cuda.py:55-179 (TestCudaExt, TestScaledE4M3).  Gradient tests are not
mirrored (PTQ path, no autograd functions)."""

import pytest
import torch

import _moa_import

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import ops  # noqa: E402

DEV = "cuda:0"


def quant(x, amax, num_bits=8, fake=False, narrow_range=True):
    """quant_utils.py:21-32."""
    intmax = 2.0 ** (num_bits - 1) - 1.0
    intmin = -intmax if narrow_range else -intmax - 1
    scale = intmax / amax
    x_q = torch.clamp((x * scale).round_(), intmin, intmax)
    if fake:
        x_q /= scale
    return x_q


def fp8_eager(x, amax):
    """tensor_quant.py:46-59 with torch's own e4m3fn cast."""
    dtype = x.dtype
    if amax is None:
        return x.to(torch.float8_e4m3fn).to(dtype)
    amax = amax.to(torch.float32)
    scale = 448.0 / torch.where(amax <= 2.0 ** -24, torch.ones_like(amax), amax)
    xs = torch.clamp(x.to(torch.float32) * scale, -448.0, 448.0)
    return (xs.to(torch.float8_e4m3fn).to(torch.float32) * (1.0 / scale)).to(dtype)


class TestFakeTensorQuant:
    def test_per_tensor_scale(self):
        x = torch.randn(31).to(DEV)
        assert torch.allclose(quant(x, torch.max(x.abs()), fake=True), ops.fake_tensor_quant(x, torch.max(torch.abs(x))))

    def test_per_channel_scale(self):
        torch.manual_seed(123)
        x = torch.randn(3, 3, 6, 8).to(DEV)
        amax_x = 0.7 * torch.amax(x.abs(), dim=(1, 2, 3), keepdims=True)  # shrunk: the clip is exercised
        assert torch.allclose(ops.fake_tensor_quant(x, amax_x), quant(x, amax_x, fake=True))

    def test_unsigned(self):
        x = torch.randn(31).abs().to(DEV)
        ref = quant(x, torch.max(x.abs()), num_bits=9, fake=True)
        assert torch.allclose(ops.fake_tensor_quant(x, torch.max(torch.abs(x)), 8, True), ref)
        x = torch.randn(3, 7).to(DEV)
        with pytest.raises(TypeError, match="Negative values encountered"):
            ops.fake_tensor_quant(x, torch.max(torch.abs(x)), 8, True, check_inputs=True)

    def test_full_range(self):
        x = torch.randn(31).abs().to(DEV)
        amax = torch.max(x.abs())
        ref = quant(x, amax, num_bits=9, fake=True, narrow_range=False)
        assert torch.allclose(ops.fake_tensor_quant(x, amax, 8, True, False), ref)

    def test_overflow_fp16(self):
        x = torch.randn(31).to(DEV).half()
        y = ops.fake_tensor_quant(x, torch.tensor(1e-4).to(DEV).half(), 8, False)
        assert not (torch.isinf(y).any() or torch.isnan(y).any())


class TestExt:
    @pytest.mark.parametrize("num_bits,unsigned", [(3, False), (4, True), (8, False), (8, True)])
    def test_num_bits(self, num_bits, unsigned):
        x = torch.randn(31).to(DEV)
        if unsigned:
            x = x.abs()
        ref = quant(x, torch.max(x.abs()), num_bits=num_bits + int(unsigned), fake=True)
        assert torch.allclose(ops.fake_tensor_quant(x, torch.max(torch.abs(x)), num_bits, unsigned), ref)

    @pytest.mark.parametrize("dtype,atol", [(torch.float32, 1e-8), (torch.float16, 1e-3), (torch.bfloat16, 1e-1)])
    def test_in_place_and_dtypes(self, dtype, atol):
        x = torch.randn(31).to(DEV).to(dtype)
        ref = quant(x.clone(), torch.max(x.abs()), fake=True)
        ops.fake_tensor_quant(x, torch.max(torch.abs(x)), inplace=True)
        assert torch.allclose(x, ref, atol=atol)

    def test_with_axis(self):
        x = torch.randn(3, 4, 5).to(DEV)
        amax = x.abs().amax(dim=(0, 2))
        got = ops.fake_tensor_quant_with_axis(x, amax, 1)
        assert torch.allclose(got, quant(x, amax.view(1, 4, 1), fake=True))

    def test_tiny_amax(self):
        x = torch.rand(2, 3, 4).to(DEV)
        amax = torch.tensor([1.0, 1.0e-26, 1.0]).to(DEV).unsqueeze(-1).unsqueeze(1)
        quant_x = ops.fake_tensor_quant_with_axis(x, amax.reshape(-1), 1)
        assert quant_x[:, 1, :].sum() == 0
        assert ops.fake_tensor_quant(x, amax.reshape(1, 3, 1))[:, 1, :].sum() == 0


class TestScaledE4M3:
    def test_e4m3_no_scale(self):
        x = torch.randn(4, 4, device=DEV, dtype=torch.float32)
        ref = fp8_eager(x, torch.tensor(448.0, device=DEV))
        assert torch.allclose(ops.scaled_e4m3(x, None), ref, atol=1e-4, rtol=1e-4)

    @pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
    def test_with_amax(self, dtype):
        x = torch.randn(4, 4, device=DEV, dtype=dtype)
        amax = ops.reduce_amax(x, axis=None, keepdims=True)
        assert torch.allclose(ops.scaled_e4m3(x, amax), fp8_eager(x, amax))

    def test_e4m3_incontiguous(self):
        x = torch.randn(4, 4).to(DEV).transpose(1, 0)
        assert not x.is_contiguous()
        ref = fp8_eager(x, torch.tensor(448.0, device=DEV))
        assert torch.allclose(ops.scaled_e4m3(x, None), ref, atol=1e-4, rtol=1e-4)

    @pytest.mark.parametrize("axis", [0, 1, 2])
    def test_e4m3_per_channel(self, axis):
        x = torch.randn(4, 4, 4, dtype=torch.float32).to(DEV)
        amax = x.abs().amax(dim=[ax for ax in range(x.ndim) if ax != axis], keepdim=True)
        assert torch.allclose(ops.scaled_e4m3(x, amax), fp8_eager(x, amax))

    def test_zero_amax_is_finite(self):
        x = torch.randn(4, 4, device=DEV, dtype=torch.float32)
        assert torch.isfinite(ops.scaled_e4m3(x, torch.zeros((1,), device=DEV))).all()

    def test_zero_amax_per_channel_is_finite(self):
        x = torch.randn(2, 3, 4, device=DEV, dtype=torch.float32)
        amax = torch.tensor([1.0, 0.0, 1.0], device=DEV).view(1, 3, 1)
        assert torch.isfinite(ops.scaled_e4m3(x, amax)).all()
