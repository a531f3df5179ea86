"""Real-quantised tensors of the path -- mirrors of modelopt.torch.quantization.qtensor.{INT4QTensor, FP8QTensor,
MXFP4QTensor} (qtensor/int4_tensor.py:39-130, fp8_tensor.py:40-151, mxfp4_tensor.py:37-144): same classmethod
`quantize(...) -> (qtensor, scales)` and `dequantize(dtype, scale=..., block_sizes=...)` surface, every pass over
tensor data is one HIP kernel (block amax, pack, unpack).  GPU tensors only."""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import _lib, numerics, ops
from ._lib import MoquantUnsupported


class BaseQuantizedTensor:
    """qtensor/base_qtensor.py: original shape / dtype + the packed data."""

    def __init__(self, original_shape, original_dtype, quantized_data):
        self.metadata = {"shape": torch.Size(original_shape), "dtype": original_dtype}
        self._quantized_data = quantized_data


def _div448(amax: torch.Tensor) -> torch.Tensor:
    """amax / 448.0.  torch's GPU `tensor / python_float` multiplies by the reciprocal, which is one ulp off the CPU result
    for some inputs (tensor / tensor divides on both devices).  numerics "host" (default): the TRUE division = the
    reference's CPU run, whatever the device; "device": torch's own kernel on the tensor's device = the reference's run
    there (numerics.py; tools/ops_fuzz.py compares that mode with the reference on the GPU)."""
    if numerics.on_host() or not amax.is_cuda:
        return amax / torch.full((), 448.0, dtype=amax.dtype, device=amax.device)
    return amax / 448.0


def _pad_last(x: torch.Tensor, block: int) -> torch.Tensor:
    pad = (-x.shape[-1]) % block
    return F.pad(x, (0, pad), "constant", 0) if pad else x


class INT4QTensor(BaseQuantizedTensor):
    @classmethod
    def quantize(cls, input: torch.Tensor, block_size: int):
        """int4_tensor.py:39-87: flat view padded to a block multiple, scales = 7 / block amax (input dtype),
        byte = ((q_even + 8) << 4) | (q_odd + 8).  Rounding follows the reference's CUDA kernel (the path taken
        for GPU tensors there): clamp, then round-half-away of v + 8 (tensor_quant_gpu.cu:322-333)."""
        assert input.shape[-1] % 2 == 0, "Input tensor must have even number on last dimension."
        flat = _pad_last(input.reshape(-1), block_size).contiguous()
        amax = ops.reduce_amax(flat.view(-1, block_size), axis=(1,))  # input dtype, [n/g, 1]
        scales = (7.0 / amax).view(-1, 1)
        packed = ops.int4_quantize(flat, scales.reshape(-1), block_size, rounding=_lib.ROUND_HALF_AWAY)
        packed = packed.reshape(*input.shape[:-1], -1)
        return cls(input.shape, input.dtype, packed), scales

    def dequantize(self, dtype: torch.dtype = None, **kwarg):
        """int4_tensor.py:89-130 (CUDA branch): (nibble - 8) / scale in the scale dtype."""
        dtype = dtype or self.metadata["dtype"]
        scales, block = kwarg["scale"], kwarg["block_sizes"][-1]
        out = ops.int4_dequantize(self._quantized_data.reshape(-1), scales.to(self._quantized_data.device).reshape(-1),
                                  block)
        n = math.prod(self.metadata["shape"])
        return out.view(-1)[:n].view(self.metadata["shape"]).to(dtype)


class FP8QTensor(BaseQuantizedTensor):
    @classmethod
    def quantize(cls, input: torch.Tensor, scales: torch.Tensor = None, axis=None, block_sizes: dict | None = None):
        """fp8_tensor.py:40-112.  Supported layouts: per-tensor, one kept axis whose scales run along the flattened
        leading dims (per-channel rows), 1-D blocks along the last dim, and blocks on both axes of a 2-D tensor
        (the FP8 2-D blockwise weight format); other N-D block grids raise."""
        x = input
        if block_sizes:
            dims = {(d if d >= 0 else input.dim() + d): b for d, b in block_sizes.items() if isinstance(d, int)}
            if input.dim() == 2 and set(dims) == {0, 1}:
                # blocks on both axes (fp8_tensor.py:62-100): pad, tile amax, one scale per br x bc tile
                x = ops.reduce_block_padding(input, dims).contiguous()
                br, bc = dims[0], dims[1]
                if scales is None:
                    scales = _div448(ops.reduce_block_amax(x, dims))
                else:
                    scales = scales.reshape(x.shape[0] // br, x.shape[1] // bc)
                q = ops.fp8_quantize_tile(x, scales, br, bc)
                if q.shape != input.shape:
                    q = q[tuple(slice(0, d) for d in input.shape)]
                return cls(input.shape, input.dtype, q), scales
            if list(dims) != [input.dim() - 1]:
                raise MoquantUnsupported("FP8QTensor: blocks along the last dim, or on both axes of a 2-D tensor "
                                         "(reference N-D blocks: fp8_tensor.py:77-100)")
            block = dims[input.dim() - 1]
            x = _pad_last(input, block).contiguous()
            if scales is None:
                amax = ops.reduce_amax(x.view(-1, block), axis=(1,))
                scales = _div448(amax).reshape(*x.shape[:-1], x.shape[-1] // block)
            else:
                scales = scales.reshape(*x.shape[:-1], x.shape[-1] // block)
        elif scales is None:
            if axis is None:
                scales = _div448(ops.reduce_amax(x))
            else:
                ax = [axis] if isinstance(axis, int) else list(axis)
                reduce_axis = [i for i in range(x.dim()) if i not in ax and (i - x.dim()) not in ax]
                scales = _div448(ops.reduce_amax(x, axis=reduce_axis))
        q = ops.fp8_quantize(x, scales)
        if q.shape != input.shape:
            q = q[tuple(slice(0, d) for d in input.shape)]
        return cls(input.shape, input.dtype, q), scales

    def dequantize(self, dtype: torch.dtype = None, **kwarg):
        """fp8_tensor.py:114-151."""
        dtype = dtype or self.metadata["dtype"]
        assert "scale" in kwarg, "Require scale for FP8 dequantization."
        scales, block_sizes = kwarg["scale"], kwarg.get("block_sizes")
        q = self._quantized_data
        dims = {(d if d >= 0 else q.dim() + d): b for d, b in (block_sizes or {}).items() if isinstance(d, int)}
        if q.dim() == 2 and set(dims) == {0, 1}:
            qp = ops.reduce_block_padding(q.view(torch.uint8), dims).contiguous()
            out = ops.fp8_dequantize_tile(qp, scales, dtype, dims[0], dims[1])
            return out[tuple(slice(0, d) for d in self.metadata["shape"])]
        if block_sizes:
            block = block_sizes.get(-1) or block_sizes.get(q.dim() - 1)
            q = _pad_last(q.view(torch.uint8), block).contiguous()
        out = ops.fp8_dequantize(q, scales, dtype)
        return out[tuple(slice(0, d) for d in self.metadata["shape"])]


class MXFP4QTensor(BaseQuantizedTensor):
    E2M1_max = 6.0

    @classmethod
    def quantize(cls, input: torch.Tensor, block_size: int | None):
        """mxfp4_tensor.py:37-81."""
        block_size = block_size or 32
        packed, e8m0 = ops.mxfp4_quantize(input, block_size)
        return cls(input.shape, input.dtype, packed), e8m0

    def dequantize(self, dtype: torch.dtype = None, **kwarg):
        """mxfp4_tensor.py:83-144."""
        dtype = dtype or self.metadata["dtype"]
        return ops.mxfp4_dequantize(self._quantized_data, kwarg["scale"], dtype, kwarg["block_sizes"][-1])


class MXFP8QTensor(BaseQuantizedTensor):
    """E4M3 elements with one E8M0 scale byte per 32 elements of the last dim (qtensor/mxfp8_tensor.py:26-268).

    Two passes over the tensor, both existing kernels: the per-block abs-max (row reduction of the (-1, 32) view,
    `moq_amax_axis`) and the tile pack with a 1 x 32 tile (`moq_fp8_pack_tile`, fp32 scales: the quotient is not
    rounded to the tensor dtype, as torch promotes `weight * scale_factor` to fp32).  The scale bytes in between are the
    reference's own formula on the [..., K / 32] amax tensor.  Dividing by 2^(e-127) here is the reference's
    multiplication by 2^(127-e): both are exact up to the same final rounding (fp32 denormals are preserved on gfx950).
    """

    E4M3_MAX = 448.0
    BLOCK_SIZE = 32
    SCALE_DTYPE = torch.uint8

    @classmethod
    def _compute_e8m0_exponent(cls, amax: torch.Tensor) -> torch.Tensor:
        """mxfp8_tensor.py:43-68: ceil(log2(amax / 448)) in fp32, -127 for blocks without a positive amax."""
        descale = amax.float() / cls.E4M3_MAX
        log2_descale = torch.where(descale > 0, torch.log2(descale), torch.tensor(-127.0, device=descale.device))
        return torch.clamp(torch.ceil(log2_descale), min=-127, max=127)

    @classmethod
    def get_weights_scaling_factor(cls, weight: torch.Tensor) -> torch.Tensor:
        """mxfp8_tensor.py:70-97: uint8 biased exponents [..., K / 32]."""
        assert weight.dim() >= 2, f"Weight must be at least 2D, got {weight.dim()}D"
        assert weight.shape[-1] % cls.BLOCK_SIZE == 0, (
            f"Weight inner dimension ({weight.shape[-1]}) must be divisible by MXFP8 block size ({cls.BLOCK_SIZE})")
        amax = ops.reduce_block_amax(weight, {-1: cls.BLOCK_SIZE})
        return (cls._compute_e8m0_exponent(amax) + 127).to(cls.SCALE_DTYPE)

    @classmethod
    def get_weights_scaling_factor_from_quantizer(cls, weight: torch.Tensor, weight_quantizer) -> torch.Tensor:
        """mxfp8_tensor.py:99-147."""
        assert weight_quantizer.block_sizes[-1] == cls.BLOCK_SIZE, (
            f"MXFP8 requires block size {cls.BLOCK_SIZE}, got {weight_quantizer.block_sizes[-1]}")
        expected = (*weight.shape[:-1], weight.shape[-1] // cls.BLOCK_SIZE)
        scale = getattr(weight_quantizer, "_scale", None)
        if scale is not None:
            assert scale.dtype == cls.SCALE_DTYPE, f"MXFP8 scale must be {cls.SCALE_DTYPE} (E8M0 format), got {scale.dtype}"
            assert tuple(scale.shape) == expected, f"Scale shape {scale.shape} does not match expected shape {expected}"
            return scale
        return cls.get_weights_scaling_factor(weight)

    @classmethod
    def quantize_with_scale(cls, weight: torch.Tensor, weights_scaling_factor: torch.Tensor) -> torch.Tensor:
        """mxfp8_tensor.py:149-197: float8_e4m3fn tensor of the weight's shape."""
        assert weights_scaling_factor.dtype == cls.SCALE_DTYPE, (
            f"weights_scaling_factor must be {cls.SCALE_DTYPE} (E8M0 format), got {weights_scaling_factor.dtype}")
        k = weight.shape[-1]
        assert k % cls.BLOCK_SIZE == 0, f"Weight inner dimension ({k}) must be divisible by MXFP8 block size ({cls.BLOCK_SIZE})"
        descale = torch.exp2(weights_scaling_factor.float() - 127)
        q = ops.fp8_quantize_tile(weight.detach().reshape(-1, k), descale.reshape(-1, k // cls.BLOCK_SIZE), 1, cls.BLOCK_SIZE)
        return q.view(weight.shape)

    @classmethod
    def quantize(cls, input: torch.Tensor, weights_scaling_factor: torch.Tensor | None = None):
        """mxfp8_tensor.py:199-233: ragged last dims are zero-padded to a whole block and cropped afterwards."""
        shape, dtype = input.shape, input.dtype
        x = ops.reduce_block_padding(input, {-1: cls.BLOCK_SIZE})
        if weights_scaling_factor is None:
            amax = ops.reduce_block_amax(x, {-1: cls.BLOCK_SIZE})
            weights_scaling_factor = (cls._compute_e8m0_exponent(amax) + 127).to(cls.SCALE_DTYPE)
        q = cls.quantize_with_scale(x, weights_scaling_factor)[..., : shape[-1]]
        return cls(shape, dtype, q), weights_scaling_factor

    def dequantize(self, dtype: torch.dtype = None, **kwarg):
        """mxfp8_tensor.py:235-268: e4m3 value * 2^(e - 127), computed in fp32, cast to `dtype`."""
        assert "scale" in kwarg, "dequantize requires 'scale' in kwargs"
        dtype = dtype or self.metadata["dtype"]
        shape = self.metadata["shape"]
        q = ops.reduce_block_padding(self._quantized_data.view(torch.uint8), {-1: self.BLOCK_SIZE})
        k = q.shape[-1]
        descale = torch.exp2(kwarg["scale"].float() - 127)
        # the product of an e4m3 value and a power of two is exact in fp32, so rounding it once to `dtype` is the
        # reference's fp32 product followed by .to(dtype)
        out = ops.fp8_dequantize_tile(q.reshape(-1, k).view(torch.float8_e4m3fn), descale.reshape(-1, k // self.BLOCK_SIZE),
                                      torch.float32, 1, self.BLOCK_SIZE)
        return out.view(*q.shape)[..., : shape[-1]].to(dtype)
