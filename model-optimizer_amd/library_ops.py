"""torch.library custom operators for the fake-quant entries -- the mirror of the reference's S2 seam
(`tensorrt::quantize_op`, `tensorrt::dynamic_block_quantize_op`, quantization/tensor_quant.py:115-270): the same
schemas, GPU implementations on the HIP kernels, and fake (meta) implementations so that graphs containing them can be
traced / exported.  They live in the `moquant::` namespace -- `tensorrt::` belongs to the reference and is already
defined when it is imported next to this package; `modelopt_plugin.install(library_ops=True)` re-points the reference's
module globals `tensor_quant.quantize_op / dynamic_block_quantize_op` at these operators.
"""

from __future__ import annotations

import torch

from . import ops

_DEFINED = False


def _quantize_impl(input, amax, num_bits=8, exponent_bits=0, unsigned=False, narrow_range=True):
    # tensor_quant.py:115-136
    if num_bits == 8 and exponent_bits == 4:
        return ops.scaled_e4m3(input, amax)
    if isinstance(num_bits, int):
        return ops.fake_tensor_quant(input, amax, num_bits, unsigned, narrow_range)
    raise ValueError(f"Invalid combination of (num_bits, exponent_bits): ({num_bits}, {exponent_bits}).")


def _formats(num_bits, exponent_bits, scale_num_bits, scale_exponent_bits):
    scale_bits = (scale_exponent_bits, scale_num_bits - scale_exponent_bits - 1)
    if exponent_bits != 0:
        num_bits = (exponent_bits, num_bits - exponent_bits - 1)
    return num_bits, scale_bits


def _dynamic_block_quantize_impl(input, block_size, amax, num_bits, exponent_bits, scale_num_bits, scale_exponent_bits):
    # tensor_quant.py:157-195
    nb, sb = _formats(num_bits, exponent_bits, scale_num_bits, scale_exponent_bits)
    if amax is not None and sb != (8, 0) and amax.numel() != 1:
        amax = amax.amax()
    return ops.dynamic_block_quant(input, block_size, amax, nb, sb)


def define() -> bool:
    """Register the operators once; False when torch.library is unavailable in this torch build."""
    global _DEFINED
    if _DEFINED:
        return True
    try:
        torch.library.define("moquant::quantize_op",
                             "(Tensor input, Tensor amax, int num_bits, int exponent_bits, bool unsigned, "
                             "bool narrow_range) -> Tensor")
        torch.library.define("moquant::dynamic_block_quantize_op",
                             "(Tensor input, int block_size, Tensor? amax, int num_bits, int exponent_bits, "
                             "int scale_num_bits, int scale_exponent_bits) -> Tensor")
        # the kernels run on the GPU only ("cuda" is the ROCm device type); no CPU implementation is registered
        torch.library.impl("moquant::quantize_op", "cuda")(_quantize_impl)
        torch.library.impl("moquant::dynamic_block_quantize_op", "cuda")(_dynamic_block_quantize_impl)
        torch.library.register_fake("moquant::quantize_op")(
            lambda input, amax, num_bits, exponent_bits, unsigned, narrow_range: torch.empty_like(input))
        torch.library.register_fake("moquant::dynamic_block_quantize_op")(
            lambda input, block_size, amax, num_bits, exponent_bits, scale_num_bits, scale_exponent_bits:
            torch.empty_like(input))
    except (AttributeError, RuntimeError):
        return False
    _DEFINED = True
    return True


def quantize_op(input, amax, num_bits=8, exponent_bits=0, unsigned=False, narrow_range=True):
    define()
    return torch.ops.moquant.quantize_op(input, amax, num_bits, exponent_bits, unsigned, narrow_range)


def dynamic_block_quantize_op(input, block_size, amax, num_bits, exponent_bits, scale_num_bits, scale_exponent_bits):
    define()
    return torch.ops.moquant.dynamic_block_quantize_op(input, block_size, amax, num_bits, exponent_bits, scale_num_bits,
                                                       scale_exponent_bits)
