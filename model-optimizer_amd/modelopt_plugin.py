"""Drop-in seams into modelopt.torch.quantization / modelopt.torch.sparsity.

`install()` makes an unmodified Model-Optimizer checkout run this path on our HIP kernels:

  S1  extension modules : modelopt caches its JIT-built CUDA extensions on function attributes
      (quantization/extensions.py:30,42,58).  We pre-seed `get_cuda_ext.extension`,
      `get_cuda_ext_fp8.extension`, `get_cuda_ext_mx.extension` with adapter objects exposing the pybind
      surface (tensor_quant.cpp:63-77, tensor_quant_gpu_fp8.cu:109-114, tensor_quant_mx.cu:393-411).  On ROCm
      the loader otherwise returns None (utils/cpp_extension.py:57-58) and modelopt runs eager ops.
  S3  quant backend     : register_quant_backend("mi355x", entrypoint) -- a fused per-quantizer path.
  S5  sparsity          : magnitude.create_asp_mask, sparsegpt.create_sgpt_mask and the SparseGPT Hessian hook are
                          re-pointed at our kernels.
  S6  utilities         : core_utils.reduce_amax (and its re-export) is re-pointed at our reductions.
  S7  algorithms        : install(algorithms=True) -- the `_calib_func` hooks of the calibration modes, weight_only_quantize,
                          fold_weight and the export packers run this package's fused flows on the reference's own model
                          objects (modelopt_algorithms.py).

Nothing here imports modelopt at module import time; `install()` raises ImportError if it is absent.
"""

from __future__ import annotations

import collections
import types

import torch

from . import _lib, ops, sparsity

# Calls served per seam entry point, and every call a seam handed BACK to the reference ("<seam>:fallback[:reason]"):
# coverage holes under the real reference are visible instead of silent (tests assert that no unexpected fallback
# happened; `STATS.clear()` resets).
STATS: collections.Counter = collections.Counter()


def _count(key: str):
    STATS[key] += 1


def _int_bits(num_bits, who):
    """A bit width the INT kernels do not take is the seam's "layout not supported": ValueError, which the reference's
    callers answer with their eager path (tensor_quant.py:386-389) -- e.g. its own test_overflow_fp16
    (tests/_test_utils/torch/quantization/tensor_quant_common.py:131-136) passes `8, False` into the (bias, num_bits)
    slots of FakeTensorQuantFunction, i.e. num_bits = False; the CUDA kernel shifts by -1 there, eager computes something
    finite, an exception other than ValueError / AttributeError would fail the caller."""
    if isinstance(num_bits, bool) or not isinstance(num_bits, int) or not 2 <= num_bits <= 16:
        _count(f"S1:{who}:fallback:num_bits={num_bits!r}")
        raise ValueError(f"{who}: num_bits={num_bits!r} is outside the INT kernels' range [2, 16]")
    return num_bits


class IntExtension:
    """Stands in for the `modelopt_cuda_ext` pybind module."""

    @staticmethod
    def fake_tensor_quant(inputs, amax, num_bits=8, unsigned=False, narrow_range=True):
        num_bits = _int_bits(num_bits, "fake_tensor_quant")
        _count("S1:fake_tensor_quant")
        return ops.fake_tensor_quant(inputs, amax.reshape(-1)[:1], num_bits, unsigned, narrow_range)

    @staticmethod
    def fake_tensor_quant_(inputs, amax, num_bits=8, unsigned=False, narrow_range=True):
        num_bits = _int_bits(num_bits, "fake_tensor_quant_")
        _count("S1:fake_tensor_quant_")
        ops.fake_tensor_quant(inputs, amax.reshape(-1)[:1], num_bits, unsigned, narrow_range, inplace=True)

    @staticmethod
    def fake_tensor_quant_with_axis(inputs, amax, axis, num_bits=8, unsigned=False, narrow_range=True):
        num_bits = _int_bits(num_bits, "fake_tensor_quant_with_axis")
        _count("S1:fake_tensor_quant_with_axis")
        return ops.fake_tensor_quant_with_axis(inputs, amax, axis, num_bits, unsigned, narrow_range)

    @staticmethod
    def INT4_quantize(input, scales, block_size):  # noqa: N802
        # the CUDA kernel's rounding (clamp, then roundf(v + 8)) -- tensor_quant_gpu.cu:322-333
        _count("S1:INT4_quantize")
        return ops.int4_quantize(input.reshape(-1), scales.reshape(-1), block_size, _lib.ROUND_HALF_AWAY)

    @staticmethod
    def INT4_dequantize(quantized_data, scales, block_size):  # noqa: N802
        _count("S1:INT4_dequantize")
        return ops.int4_dequantize(quantized_data, scales.reshape(-1), block_size)

    # NF4 is outside the MI355X PTQ path -- but an extension object cannot be absent for two of its entries only,
    # and without an extension the reference has a working branch of its own (qtensor/nf4_tensor.py:103-118, :178-199).  These
    # two entries therefore run THAT branch, with the reference's own helpers and table, on the tensor's device (counted): what a
    # ROCm user of the reference gets today stays what they get with the seams installed.
    @staticmethod
    def NF4_quantize(input, scales, block_size):  # noqa: N802
        from modelopt.torch.quantization.qtensor import nf4_tensor as ref_nf4

        _count("S1:NF4_quantize:reference-eager")
        blocks = input.view(-1, block_size)
        scaled = blocks / scales.view(blocks.shape[0], -1)
        q = ref_nf4._quantize_to_nearest_lut(scaled.flatten(), ref_nf4.nf4_table.to(device=input.device, dtype=input.dtype))
        q = q.to(torch.uint8)
        return q[::2] << 4 | q[1::2]

    @staticmethod
    def NF4_dequantize(quantized_data, scales, block_size):  # noqa: N802
        from modelopt.torch.quantization.qtensor import nf4_tensor as ref_nf4

        _count("S1:NF4_dequantize:reference-eager")
        first = ref_nf4._nf4_lookup((quantized_data >> 4).to(torch.long)).view(-1, block_size // 2) * scales.view(-1, 1)
        second = ref_nf4._nf4_lookup((quantized_data & 0x0F).to(torch.long)).view(-1, block_size // 2) * scales.view(-1, 1)
        return torch.stack([first.flatten(), second.flatten()], dim=-1).view(-1)


class Fp8Extension:
    """Stands in for `modelopt_cuda_ext_fp8`."""

    @staticmethod
    def fake_e4m3fy(inputs, amax):
        _count("S1:fake_e4m3fy")
        return ops.scaled_e4m3(inputs, amax.reshape(-1)[:1])

    @staticmethod
    def fake_e4m3fy_with_axis(inputs, amax, axis):
        _count("S1:fake_e4m3fy_with_axis")
        return ops.fake_e4m3fy_with_axis(inputs, amax, axis)


class MxExtension:
    """Stands in for `modelopt_cuda_ext_mx` (Types numbering: tensor_quant_mx.h:39)."""

    Types = types.SimpleNamespace(**_lib.MX_TYPES)

    @staticmethod
    def fused_amax_convert(inputs, block_size, format, scale_format, global_amax=None):
        _count("S1:fused_amax_convert")
        return ops.fused_amax_convert(inputs, block_size, int(format), int(scale_format), global_amax)

    @staticmethod
    def convert_to_exmy(x, format):
        _count("S1:convert_to_exmy")
        return ops.convert_to_exmy(x, int(format))


def mi355x_backend(inputs: torch.Tensor, tq) -> torch.Tensor:
    """S3 entrypoint(inputs, tensor_quantizer): fused dynamic-amax QDQ for static-block INT quantizers,
    plain kernels otherwise.  `tq` is a *modelopt* TensorQuantizer (duck-typed)."""
    _count("S3:mi355x_backend")
    nb = tq._num_bits
    amax = getattr(tq, "_amax", None)
    if isinstance(nb, int) and tq.block_sizes and amax is None and inputs.dim() == 2:
        y, _ = ops.amax_qdq_int_group(inputs, inputs.shape[-1], nb, tq._unsigned, tq._narrow_range,
                                      return_amax=False)
        return y
    if amax is None:
        from .calib import convert_quantization_axis_to_reduce_axis

        amax = ops.reduce_amax(inputs, axis=convert_quantization_axis_to_reduce_axis(inputs, tq._axis))
    if isinstance(nb, tuple):
        return ops.scaled_e4m3(inputs, amax)
    return ops.fake_tensor_quant(inputs, amax, nb, tq._unsigned, tq._narrow_range)


def _takes(t: torch.Tensor) -> bool:
    """The seams serve GPU tensors and leave everything else to the reference's own code."""
    return t.is_cuda


def _reduce_amax_seam(original):
    def reduce_amax(input, axis=None, keepdims=True, squeeze_scalar=True):
        if not _takes(input):
            return original(input, axis=axis, keepdims=keepdims, squeeze_scalar=squeeze_scalar)
        try:
            out = ops.reduce_amax(input, axis=axis, keepdims=keepdims, squeeze_scalar=squeeze_scalar)
            _count("S6:reduce_amax")
            return out
        except _lib.MoquantUnsupported as e:
            # the reference's convention for its own extensions (tensor_quant.py:386-389): an unsupported layout
            # falls back to eager -- counted, so that coverage holes show up
            _count(f"S6:reduce_amax:fallback:{type(e).__name__}:{str(e)[:60]}")
            return original(input, axis=axis, keepdims=keepdims, squeeze_scalar=squeeze_scalar)

    return reduce_amax


def _asp_mask_seam(original):
    def create_asp_mask(tensor, pattern):
        if not _takes(tensor):
            return original(tensor, pattern)
        _count("S5:create_asp_mask")
        return sparsity.create_asp_mask(tensor, pattern)

    return create_asp_mask


def _sgpt_mask_seam(original):
    def create_sgpt_mask(tensor, hessian, config):
        if not _takes(tensor):
            return original(tensor, hessian, config)
        _count("S5:create_sgpt_mask")
        return sparsity.create_sgpt_mask(tensor, hessian, dict(config))

    return create_sgpt_mask


def _sgpt_hessian_seam(original):
    """SparseGPTSearcher._hook_compute_hessian (sparsegpt.py:238-276): 16-bit GPU activations of a linear take the
    MFMA accumulation; everything else (CPU Hessians, conv layers, fp32 inputs) stays on the reference's code."""

    def hook(cls, mod, inp, out):
        x = inp[0] if isinstance(inp, tuple) else inp
        if not ("Linear" in type(mod).__name__ and _takes(x) and _takes(mod.hessian)
                and x.dtype in (torch.bfloat16, torch.float16) and x.shape[-1] % 4 == 0):
            return original.__func__(cls, mod, inp, out)
        _count("S5:sgpt_hessian")
        b = 1 if x.dim() == 2 else x.shape[0]
        decay = mod.samples / (mod.samples + b)
        mod.samples += b
        ops.hessian_accum(mod.hessian, x.reshape(-1, x.shape[-1]), decay, 2.0 / mod.samples)

    return classmethod(hook)


def _library_op_seam(original, ours):
    def op(inputs, *args, **kwargs):
        if not _takes(inputs):
            return original(inputs, *args, **kwargs)
        _count("S2:" + getattr(ours, "__name__", "library_op"))
        return ours(inputs, *args, **kwargs)

    op._moq_seam = True
    return op


# what install() replaced, so that uninstall() can put the reference back as it was (A/B runs of the un-installed
# reference in the same process: tests/test_gpu_reference_live.py)
_UNSET = object()
_SAVED: list = []


def _swap(obj, name, new):
    _SAVED.append((obj, name, obj.__dict__.get(name, _UNSET) if hasattr(obj, "__dict__") else getattr(obj, name, _UNSET)))
    setattr(obj, name, new)


def uninstall():
    """Undo install(): every re-pointed attribute gets its previous value back (or is deleted if it did not exist).  The
    registered quant backend name stays (the reference has no unregister call); nothing selects it unless configured."""
    while _SAVED:
        obj, name, old = _SAVED.pop()
        if old is _UNSET:
            try:
                delattr(obj, name)
            except AttributeError:
                pass
        else:
            setattr(obj, name, old)


def install(extensions: bool = True, backend: bool = True, utilities: bool = True, sparsity_seam: bool = True,
            library_ops: bool = False, algorithms: bool = False):
    """Wire the seams into an importable modelopt.  Returns the list of seams installed.  `library_ops` additionally
    re-points the S2 module globals `tensor_quant.quantize_op / dynamic_block_quantize_op` at the `moquant::`
    torch.library operators for GPU tensors (redundant with S1 for results; it removes the reference's Python
    dispatch layers from the call and keeps graphs traceable through our fake implementations).

    `algorithms` (S7, modelopt_algorithms.py): the reference's calibration-algorithm hooks (`_calib_func` of its mode
    descriptors, `weight_only_quantize`, `fold_weight`) and export packers are re-pointed at this package's fused
    implementations, run on the reference's own model objects -- an unmodified `mtq.quantize(model, cfg, loop)` then takes
    the multi-tensor / deferred-statistics / Gram-screen path instead of one kernel call per quantizer call."""
    import modelopt.torch.quantization.extensions as ext  # ImportError if modelopt is absent

    installed = []
    if extensions:
        _swap(ext.get_cuda_ext, "extension", IntExtension())
        _swap(ext.get_cuda_ext_fp8, "extension", Fp8Extension())
        _swap(ext.get_cuda_ext_mx, "extension", MxExtension())
        installed.append("S1:extensions")
    if backend:
        from modelopt.torch.quantization.nn.modules import tensor_quantizer as mtq_tq

        mtq_tq.register_quant_backend("mi355x", mi355x_backend)
        installed.append("S3:backend=mi355x")
    if utilities:
        import modelopt.torch.quantization.utils as qutils
        from modelopt.torch.quantization.utils import core_utils

        if not getattr(core_utils.reduce_amax, "_moq_seam", False):
            seam = _reduce_amax_seam(core_utils.reduce_amax)
            seam._moq_seam = True
            _swap(core_utils, "reduce_amax", seam)
            _swap(qutils, "reduce_amax", seam)
        installed.append("S6:reduce_amax")
    if sparsity_seam:
        from modelopt.torch.sparsity.weight_sparsity import magnitude

        if not getattr(magnitude.create_asp_mask, "_moq_seam", False):
            seam = _asp_mask_seam(magnitude.create_asp_mask)
            seam._moq_seam = True
            _swap(magnitude, "create_asp_mask", seam)
        installed.append("S5:create_asp_mask")
        from modelopt.torch.sparsity.weight_sparsity import sparsegpt

        if not getattr(sparsegpt.create_sgpt_mask, "_moq_seam", False):
            seam = _sgpt_mask_seam(sparsegpt.create_sgpt_mask)
            seam._moq_seam = True
            _swap(sparsegpt, "create_sgpt_mask", seam)
            _swap(sparsegpt.SparseGPTSearcher, "_hook_compute_hessian", _sgpt_hessian_seam(
                sparsegpt.SparseGPTSearcher.__dict__["_hook_compute_hessian"]))
        installed.append("S5:create_sgpt_mask")
    if library_ops:
        from modelopt.torch.quantization import tensor_quant as ref_tq

        from . import library_ops as lo

        if lo.define():
            if not getattr(ref_tq.quantize_op, "_moq_seam", False):
                _swap(ref_tq, "quantize_op", _library_op_seam(ref_tq.quantize_op, lo.quantize_op))
                _swap(ref_tq, "dynamic_block_quantize_op", _library_op_seam(ref_tq.dynamic_block_quantize_op,
                                                                           lo.dynamic_block_quantize_op))
            installed.append("S2:library_ops")
    if algorithms:
        from . import modelopt_algorithms

        installed += modelopt_algorithms.install_algorithms(_swap)
    return installed
