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
        # seeded: the helper multiplies in the tensor's dtype, the kernel in fp32 -- an unlucky f16 draw lands a product on a
        # .5 tie of the 16-bit grid and the two round to neighbouring levels (seen once in ~50 unseeded runs)
        x = torch.randn(31, generator=torch.Generator().manual_seed(1234)).to(DEV).to(dtype)
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


# ------------------------------------------------------------------------------------------------------------------
# tests/unit/torch/quantization/test_calibrator.py restated (the tensors live on the GPU; same values, same bounds)
from model_optimizer_amd import calib  # noqa: E402


class TestMaxCalibrator:
    def test_simple_run(self):
        c = calib.MaxCalibrator(8, None, False)
        x_1, x_2 = torch.rand(16).to(DEV), torch.rand(16).to(DEV)
        c.collect(x_1)
        c.collect(x_2)
        assert torch.allclose(c.compute_amax(), torch.max(x_1.max(), x_2.max()), atol=0, rtol=0)
        calib.MaxCalibrator(8, None, True)

    @pytest.mark.parametrize("axis", [0, -4])
    def test_fine_grain(self, axis):
        c = calib.MaxCalibrator(8, axis, False)
        x_1, x_2 = torch.rand(3, 4, 2, 2).to(DEV), torch.rand(3, 4, 2, 2).to(DEV)
        c.collect(x_1)
        c.collect(x_2)
        assert c.compute_amax().shape[0] == 3
        assert torch.allclose(c.compute_amax(), ops.reduce_amax(torch.max(x_1, x_2), axis=(1, 2, 3)), atol=0, rtol=0)
        c.reset()
        assert c.compute_amax() is None

    def test_track_amax(self):
        import numpy as np
        c = calib.MaxCalibrator(8, None, False, track_amax=True)
        x_1, x_2 = torch.rand(16).to(DEV), torch.rand(16).to(DEV)
        c.collect(x_1)
        c.collect(x_2)
        assert torch.allclose(c.compute_amax(), torch.max(x_1.max(), x_2.max()), atol=0, rtol=0)
        np.testing.assert_array_equal(c.amaxs[0], x_1.max().cpu().numpy())
        np.testing.assert_array_equal(c.amaxs[1], x_2.max().cpu().numpy())

    def test_shape_change_raises(self):
        c = calib.MaxCalibrator(8, 0, False)
        c.collect(torch.rand(3, 4, 2, 2).to(DEV))
        with pytest.raises(RuntimeError, match="shape changed"):
            c.collect(torch.rand(4, 4, 2, 2).to(DEV))


class TestHistogramCalibrators:
    def test_skip_zeros(self):
        c = calib.HistogramCalibrator(8, None, False, num_bins=2048, skip_zeros=True)
        c.collect(torch.tensor([0, 0, 0, 0, 0, 1, 2, 3, 4, 5]).to(DEV))
        c.collect(torch.tensor([0, 0, 0, 0, 0, 6, 7, 8, 9, 10]).to(DEV))
        amax = c.compute_amax("percentile", percentile=50, start_bin=128)
        assert (amax - 5.0).abs() < 10 / 2048

    def test_grown_histogram_equals_numpy(self):
        """test_torch_hist: counts after growth equal numpy's histogram over the grown range."""
        import numpy as np
        torch.manual_seed(0)
        x_1 = torch.rand(15)
        x_1[0] = 0
        x_2 = torch.rand(15) + 1
        x_2[1] = 0
        c = calib.HistogramCalibrator(8, None, False, num_bins=19, torch_hist=True)
        c.collect(x_1.to(DEV))
        assert c._calib_hist.numel() == c._calib_bin_edges.numel() - 1
        want, edges = np.histogram(x_1.numpy(), bins=19, range=(0, x_1.max().item()))
        np.testing.assert_array_equal(want, c._calib_hist.cpu().numpy())
        np.testing.assert_array_almost_equal(edges, c._calib_bin_edges.cpu().numpy())
        for _ in range(3):
            c.collect(x_2.to(DEV))
            c.collect(x_1.to(DEV))
            c.compute_amax("percentile", percentile=99.99)
            assert c._calib_hist.numel() == c._calib_bin_edges.numel() - 1
        assert int(c._calib_hist.sum()) == 15 * 7

    @pytest.mark.parametrize("unsigned", [False, True])
    def test_entropy_one_tensor(self, unsigned):
        c = calib.HistogramCalibrator(8, None, unsigned, num_bins=512, grow_method="stretch")
        x_2 = torch.rand(11, 7, 3, 3)
        x_2[1, 1, 1, 1] = 10.0  # the outlier must be discarded by the KL search
        c.collect(x_2.to(DEV))
        assert c.compute_amax("entropy", start_bin=32) < 1.1

    def test_entropy_two_tensor(self):
        c = calib.HistogramCalibrator(8, None, False, num_bins=512)
        x_2 = torch.rand(11, 7, 3, 3)
        x_2[1, 1, 1, 1] = 10.0
        c.collect(x_2.to(DEV))
        c.collect(torch.rand(11, 7, 3, 3).to(DEV))
        assert c.compute_amax("entropy", start_bin=32) < 1.1

    def test_mse_one_tensor(self):
        c = calib.HistogramCalibrator(8, None, False, num_bins=32)
        x_1 = torch.ones(4, 4, 4) * 255.0
        x_1[1, 1, 1] = 256.0
        c.collect(x_1.to(DEV))
        amax = c.compute_amax("mse", start_bin=16).cpu()
        assert (amax - 255.0).abs() < (amax - 256.0).abs()

    def test_mse_unsigned_one_tensor(self):
        c = calib.HistogramCalibrator(8, None, True, num_bins=32)
        x_1 = torch.ones(11, 7, 3, 3) * 512.0
        x_1[1, 1, 1, 1] = 513.0
        c.collect(x_1.to(DEV))
        amax = c.compute_amax("mse", start_bin=8).cpu()
        assert (amax - 512.0).abs() < (amax - 513.0).abs()

    def test_mse_two_tensor(self):
        c = calib.HistogramCalibrator(8, None, False)
        x_1 = torch.ones(11, 7, 3, 3) * 255.0
        x_1[1, 1, 1, 1] = 256.0
        c.collect(x_1.to(DEV))
        c.collect((torch.ones(11, 7, 3, 3) * 255.0).to(DEV))
        amax = c.compute_amax("mse").cpu()
        assert (amax - 255.0).abs() < (amax - 256.0).abs()

    def test_percentile_one_tensor(self):
        c = calib.HistogramCalibrator(8, None, False)
        c.collect(torch.arange(100).to(DEV))
        assert (c.compute_amax("percentile", percentile=90) - 89.0).abs() < 100 / 1024

    def test_percentile_unsigned_one_tensor(self):
        c = calib.HistogramCalibrator(8, None, True)
        c.collect(torch.arange(100).to(DEV))
        assert (c.compute_amax("percentile", percentile=80) - 79.0).abs() < 100 / 2048

    def test_percentile_two_tensor(self):
        c = calib.HistogramCalibrator(8, None, False)
        c.collect(torch.arange(100).to(DEV))
        c.collect(torch.arange(0, 50, 0.5).to(DEV))
        assert (c.compute_amax("percentile", percentile=99) - 97.0).abs() < 100 / 1024

    def test_percentile_range_and_repr(self):
        c = calib.HistogramCalibrator(8, None, False)
        c.collect(torch.arange(100).to(DEV))
        with pytest.raises(ValueError, match="range"):
            c.compute_amax("percentile", percentile=-10)
        with pytest.raises(ValueError, match="range"):
            c.compute_amax("percentile", percentile=200)
        repr(c)
