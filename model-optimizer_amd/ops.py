"""Functional layer over the C-ABI: torch tensors in, torch tensors out, HIP kernels in between.

Names, argument meaning and error behaviour mirror the reference's L1 functional ops
(modelopt/torch/quantization/tensor_quant.py, utils/core_utils.py, sparsity/weight_sparsity/magnitude.py)
so that parity tests read like the reference's own tests.  torch is used for device memory, streams and
tiny host-side bookkeeping only; every pass over tensor data is one of our kernels.
"""

from __future__ import annotations

import ctypes
from contextlib import contextmanager

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import MoquantError, MoquantUnsupported, check

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise MoquantUnsupported(f"dtype {t.dtype} is not supported by libmoquant (f32/f16/bf16)") from None


def _require_gpu(t: torch.Tensor, what: str) -> None:
    # reference: TORCH_CHECK(inputs.is_cuda()) (tensor_quant.cpp:46,56) -> RuntimeError
    if not t.is_cuda:
        raise MoquantError(f"{what}: tensor must live on the GPU (got {t.device}); "
                           "there is no CPU path in model_optimizer_amd")


def _is_gpu(t: torch.Tensor) -> bool:
    """Does this tensor live where the library runs?  (One place to ask, so that the CPU tier's host-memory stand-in for the
    C-ABI can answer yes for host tensors: tests/hostmem_backend.py.)"""
    return t.is_cuda


@contextmanager
def _on(t: torch.Tensor):
    """Device guard + current stream handle (reference: same_device_as, tensor_quantizer.py:1205)."""
    with torch.cuda.device(t.device):
        yield ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _p(t: torch.Tensor | None):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


# ----------------------------------------------------------------------------------------------- amax
def _is_dense(x: torch.Tensor) -> bool:
    """Non-overlapping and dense: some permutation of the dims is contiguous (transposed views of a contiguous
    buffer -- HF key / value states [B, heads, S, D] are `.view(B, S, heads, D).transpose(1, 2)`)."""
    if x.is_contiguous():
        return True
    expect = 1
    for st, sz in sorted((st, sz) for sz, st in zip(x.shape, x.stride()) if sz != 1):
        if st != expect:
            return False
        expect *= sz
    return True


def _flat_alias(x: torch.Tensor) -> torch.Tensor:
    """The memory of a dense tensor as one 1-D run (what a per-tensor statistic or an elementwise per-tensor QDQ may
    walk in any order): no `.contiguous()` copy -- one of the extra passes listed for TensorQuantizer.forward."""
    return x if x.is_contiguous() else x.as_strided((x.numel(),), (1,))


def _reduce_layout(shape, reduce_axes):
    """Factor `shape` as [outer, kept, inner] when exactly one block of adjacent dims is kept."""
    nd = len(shape)
    red = sorted({a % nd for a in reduce_axes})
    keep = [d for d in range(nd) if d not in red]
    if not keep:
        return None
    if keep != list(range(keep[0], keep[-1] + 1)):
        raise MoquantUnsupported(f"reduce_amax: kept dims {keep} are not adjacent")
    outer = 1
    for d in range(keep[0]):
        outer *= shape[d]
    kept = 1
    for d in keep:
        kept *= shape[d]
    inner = 1
    for d in range(keep[-1] + 1, nd):
        inner *= shape[d]
    return outer, kept, inner, keep


@torch.no_grad()
def reduce_amax(input: torch.Tensor, axis=None, keepdims=True, squeeze_scalar=True,
                out: torch.Tensor | None = None, accumulate: bool = False) -> torch.Tensor:
    """Abs-max over `axis` (the dims to REDUCE; None = all) -- core_utils.py:146-183.

    Returns the input dtype like the reference (exact: a max of representable values).  `out`/`accumulate`
    expose the fused running max used by MaxCalibrator (fp32 buffer, updated in place).
    """
    _require_gpu(input, "reduce_amax")
    x = input.detach()
    if x.dtype in (torch.float8_e4m3fn, torch.float8_e5m2):
        x = x.to(torch.get_default_dtype())  # core_utils.py:169-170
    nd = x.dim()
    if isinstance(axis, int):
        axis = (axis,)
    per_tensor = axis is None or len({a % nd for a in axis}) == nd or nd == 0
    x = _flat_alias(x) if per_tensor and _is_dense(x) else x.contiguous()
    with _on(x) as stream:
        if per_tensor:
            buf = out if out is not None else torch.empty(1, dtype=torch.float32, device=x.device)
            check(_lib.lib().moq_amax(_p(x), x.numel(), _dt(x), _p(buf), int(accumulate), stream))
            if out is not None:
                return out
            if axis is None:
                return buf.reshape(()).to(x.dtype)  # torch.max(input) is 0-dim
            res = buf.to(x.dtype).reshape([1] * nd if keepdims else [])
            if squeeze_scalar:
                res = res.reshape(())
            return res
        if len(axis) == 0:
            raise MoquantUnsupported("reduce_amax: empty reduce axis list")
        if _is_block2d_view(x, reduce_axes=axis):
            # [A, br, B, bc] reduced over (1, 3): 2-D block amax (reduce_block_amax, core_utils.py:43-90)
            buf = block2d(x, 0, amax=out, accumulate=accumulate)
            if out is not None:
                return out
            res = buf.to(x.dtype)
            return res if keepdims else res.reshape(x.shape[0], x.shape[2])
        try:
            outer, kept, inner, keep = _reduce_layout(list(x.shape), axis)
        except MoquantUnsupported:
            # kept dims that are not one adjacent block (the reference's N-D block views keep e.g. dims 0, 1, 2, 4 of a 5-D
            # tensor): ONE permuted copy brings the kept dims to the front, then it is the per-row reduction.  A maximum
            # is exact, so the copy changes nothing but the cost (a cold path: no BASELINE configuration reduces this way)
            red = sorted({a % nd for a in axis})
            keep = [d for d in range(nd) if d not in red]
            x = x.permute(keep + red).contiguous()
            outer, kept, inner = 1, 1, 1
            for d in range(len(keep)):
                kept *= x.shape[d]
            for d in range(len(keep), nd):
                inner *= x.shape[d]
            shape_in = list(input.shape)
            buf = out if out is not None else torch.empty(kept, dtype=torch.float32, device=x.device)
            check(_lib.lib().moq_amax_axis(_p(x), outer, kept, inner, _dt(x), _p(buf), int(accumulate), stream))
            if out is not None:
                return out
            shape = [shape_in[d] if d in keep else 1 for d in range(nd)] if keepdims else [shape_in[d] for d in keep]
            res = buf.to(x.dtype).reshape(shape)
            return res.reshape(()) if squeeze_scalar and res.numel() == 1 else res
        buf = out if out is not None else torch.empty(kept, dtype=torch.float32, device=x.device)
        check(_lib.lib().moq_amax_axis(_p(x), outer, kept, inner, _dt(x), _p(buf), int(accumulate), stream))
        if out is not None:
            return out
        shape = [x.shape[d] if d in keep else 1 for d in range(nd)] if keepdims else [x.shape[d] for d in keep]
        res = buf.to(x.dtype).reshape(shape)
        if squeeze_scalar and res.numel() == 1:
            res = res.reshape(())
        return res


def reduce_block_padding(input: torch.Tensor, block_sizes: dict, pad_value: float = 0) -> torch.Tensor:
    """core_utils.py:93-125: right-pad every blocked dim to a multiple of its block size."""
    nd = input.dim()
    pad = [0] * (2 * nd)
    for dim, block in block_sizes.items():
        if not isinstance(dim, int):
            continue
        d = dim if dim >= 0 else nd + dim
        rem = input.size(d) % block
        if rem:
            pad[(nd - 1 - d) * 2 + 1] = block - rem
    return F.pad(input, pad, value=pad_value) if any(pad) else input


@torch.no_grad()
def reduce_block_amax(input_tensor: torch.Tensor, block_sizes: dict) -> torch.Tensor:
    """core_utils.py:43-90: abs-max over blocks along every dim named in block_sizes; the result has the input's
    rank with each blocked dim divided by its block size, in the input dtype.

    A 2-D tensor blocked on both axes is ONE kernel with the tile in registers (moq_block2d); otherwise one reduction
    per blocked dim, like the reference, but without its clone: rows of a (-1, g) view for the last dim, lanes along
    the trailing dims for any other (moq_amax_mid).  Max is exact, so the order of the steps does not matter."""
    _require_gpu(input_tensor, "reduce_block_amax")
    x = input_tensor.detach().contiguous()
    nd = x.dim()
    dims = {(d if d >= 0 else nd + d): b for d, b in block_sizes.items() if isinstance(d, int)}
    for d, b in dims.items():
        assert x.shape[d] % b == 0, f"Tensor dimension {d}, {x.shape[d]} is not divisible by {b}"
    vec = 4 if x.dtype == torch.float32 else 8
    if nd == 2 and set(dims) == {0, 1} and dims[1] % vec == 0 and dims[0] * dims[1] <= 16 * 256 * vec \
            and x.data_ptr() % 16 == 0:
        x4 = x.view(x.shape[0] // dims[0], dims[0], x.shape[1] // dims[1], dims[1])
        return block2d(x4, 0).to(x.dtype).reshape(x.shape[0] // dims[0], x.shape[1] // dims[1])
    cur = x
    for d in sorted(dims, reverse=True):  # last dim first: the big pass uses the row kernel
        b = dims[d]
        shape = list(cur.shape)
        outer = 1
        for k in range(d):
            outer *= shape[k]
        inner = 1
        for k in range(d + 1, nd):
            inner *= shape[k]
        nblk = shape[d] // b
        if inner == 1:
            red = reduce_amax(cur.reshape(-1, b), axis=[1], keepdims=False).reshape(-1)
        else:
            buf = torch.empty(outer * nblk * inner, dtype=torch.float32, device=cur.device)
            with _on(cur) as stream:
                check(_lib.lib().moq_amax_mid(_p(cur), outer * nblk, b, inner, _dt(cur), _p(buf), stream))
            red = buf.to(cur.dtype)
        shape[d] = nblk
        cur = red.reshape(shape)
    return cur


# ----------------------------------------------------------------------------------------------- QDQ
def _amax_mode(inputs: torch.Tensor, amax: torch.Tensor):
    """(mode, axis_size, inner) from an amax tensor, following fake_quant_impl / scaled_e4m3_impl
    (tensor_quant.py:83-91, :103-111): numel 1 -> per tensor; else one non-singleton dim = the axis."""
    if amax.numel() == 1:
        return _lib.AMAX_SCALAR, 1, 1
    if amax.dim() != inputs.dim() and amax.squeeze().dim() != 1:
        raise MoquantUnsupported("amax must be a scalar or have exactly one non-singleton dim")
    if amax.dim() == inputs.dim():
        if amax.squeeze().dim() > 1:
            # amax over a leading prefix of dims (per-token activations: [B, T, 1] for [B, T, H]) is axis 0 of the
            # [B * T, H] view of a contiguous input
            k = max(d for d in range(amax.dim()) if amax.shape[d] != 1) + 1
            if tuple(amax.shape[:k]) == tuple(inputs.shape[:k]) and inputs.is_contiguous():
                inner = 1
                for d in range(k, inputs.dim()):
                    inner *= inputs.shape[d]
                return _lib.AMAX_AXIS, amax.numel(), inner
            # reference: ValueError -> caller falls back to eager (tensor_quant.py:386-389)
            raise MoquantUnsupported("multi-dimensional amax is not supported by the kernel")
        axis = list(amax.shape).index(amax.numel())
    elif amax.dim() == 1 and inputs.dim() >= 1:
        axis = None
        for d in range(inputs.dim()):
            if inputs.shape[d] == amax.numel():
                axis = d
                break
        if axis is None:
            raise MoquantError("amax length does not match any input dim")
    else:
        raise MoquantUnsupported("unsupported amax shape")
    if inputs.shape[axis] != amax.numel():
        raise MoquantError(f"amax.numel()={amax.numel()} != inputs.size({axis})={inputs.shape[axis]}")
    inner = 1
    for d in range(axis + 1, inputs.dim()):
        inner *= inputs.shape[d]
    return _lib.AMAX_AXIS, inputs.shape[axis], inner


def _gather_kept_axes(inputs: torch.Tensor, amax: torch.Tensor):
    """An amax that keeps SEVERAL axes which are not a leading prefix of the input (TensorQuantizer(axis=(0, 2)) on a rank-3
    tensor: amax [A, 1, C]): the reference's CUDA kernel refuses these and its eager path broadcasts (tensor_quant.py:83-91,
    :386-389, :607-645).  Here: (input permuted so that the kept axes lead, as a contiguous copy; the amax permuted alike
    = a leading-prefix amax the per-axis kernel takes; the inverse permutation for the result) -- or None when the amax is
    of a shape the kernels take as it is.  Two copies, a cold path found by tools/quantizer_fuzz.py."""
    if amax.dim() != inputs.dim() or amax.numel() == 1:
        return None
    keep = [d for d in range(amax.dim()) if amax.shape[d] != 1]
    if len(keep) < 2 or keep == list(range(len(keep))):
        return None
    if any(amax.shape[d] != inputs.shape[d] for d in keep):
        raise MoquantError(f"amax shape {tuple(amax.shape)} does not broadcast over the input {tuple(inputs.shape)}")
    order = keep + [d for d in range(inputs.dim()) if d not in keep]
    inverse = [order.index(d) for d in range(inputs.dim())]
    return inputs.permute(order).contiguous(), amax.permute(order).contiguous(), inverse


@torch.no_grad()
def fake_tensor_quant(inputs: torch.Tensor, amax: torch.Tensor, num_bits: int = 8, unsigned: bool = False,
                      narrow_range: bool = True, inplace: bool = False,
                      check_inputs: bool = False) -> torch.Tensor:
    """INT-k quantize-dequantize -- tensor_quant.py:607-645 / tensor_quant_gpu.cu:43-140."""
    _require_gpu(inputs, "fake_tensor_quant")
    am = _f32(amax, inputs.device)
    apart = None if inplace else _gather_kept_axes(inputs, am)
    if apart is not None:
        y = fake_tensor_quant(apart[0], apart[1], num_bits, unsigned, narrow_range, False, check_inputs)
        return y.permute(apart[2]).contiguous()
    if not inplace and am.numel() == 1 and not inputs.is_contiguous() and _is_dense(inputs) and not check_inputs:
        # per-tensor amax on a permuted dense tensor: elementwise over its memory, output keeps the strides
        y = torch.empty_like(inputs)
        xa, ya = _flat_alias(inputs.detach()), _flat_alias(y)
        with _on(xa) as stream:
            check(_lib.lib().moq_fake_quant_int(_p(xa), _p(ya), xa.numel(), _dt(xa), _p(am), _lib.AMAX_SCALAR, 1, 1,
                                                int(num_bits), int(unsigned), int(narrow_range), stream))
        return y
    x = inputs if inplace else inputs.contiguous()
    if inplace and not x.is_contiguous():
        raise MoquantError("in-place fake quant needs a contiguous tensor")  # tensor_quant.cpp:41-42
    if _is_block2d_view(x, amax_shape=am.shape):
        return block2d(x, 1, amax=am, fp8=False, num_bits=num_bits, unsigned=unsigned, narrow_range=narrow_range,
                       out=x if inplace else None)
    if check_inputs:
        # the eager reference raises here (tensor_quant.py:611, :619-620); the CUDA extension only has
        # device asserts.  Both checks cost a device->host sync, so they are opt-in.
        if unsigned and x.numel() and bool((x.min() < 0).item()):
            raise TypeError("Negative values encountered in unsigned quantization.")
        if bool((am.min() < 0).item()):
            raise ValueError("Negative values in amax")
    mode, axis_size, inner = _amax_mode(x, am)
    y = x if inplace else torch.empty_like(x)
    with _on(x) as stream:
        check(_lib.lib().moq_fake_quant_int(_p(x), _p(y), x.numel(), _dt(x), _p(am), mode, axis_size, inner,
                                            int(num_bits), int(unsigned), int(narrow_range), stream))
    return y


@torch.no_grad()
def scaled_e4m3(inputs: torch.Tensor, amax: torch.Tensor | None) -> torch.Tensor:
    """FP8-E4M3 quantize-dequantize -- tensor_quant.py:46-92 / tensor_quant_gpu_fp8.cu:35-107."""
    _require_gpu(inputs, "scaled_e4m3")
    if (amax is None or amax.numel() == 1) and not inputs.is_contiguous() and _is_dense(inputs):
        y = torch.empty_like(inputs)  # keeps the (permuted) strides of a dense input
        xa, ya = _flat_alias(inputs.detach()), _flat_alias(y)
        am = None if amax is None else _f32(amax, inputs.device)
        with _on(xa) as stream:
            check(_lib.lib().moq_fake_quant_e4m3(_p(xa), _p(ya), xa.numel(), _dt(xa), _p(am), _lib.AMAX_SCALAR, 1, 1,
                                                 stream))
        return y
    if amax is not None:
        apart = _gather_kept_axes(inputs, _f32(amax, inputs.device))
        if apart is not None:
            return scaled_e4m3(apart[0], apart[1]).permute(apart[2]).contiguous()
    x = inputs.contiguous()
    y = torch.empty_like(x)
    with _on(x) as stream:
        if amax is None:
            check(_lib.lib().moq_fake_quant_e4m3(_p(x), _p(y), x.numel(), _dt(x), None, _lib.AMAX_SCALAR, 1,
                                                 1, stream))
        else:
            am = _f32(amax, x.device)
            if _is_block2d_view(x, amax_shape=am.shape):
                return block2d(x, 1, amax=am, fp8=True, out=y)
            mode, axis_size, inner = _amax_mode(x, am)
            check(_lib.lib().moq_fake_quant_e4m3(_p(x), _p(y), x.numel(), _dt(x), _p(am), mode, axis_size,
                                                 inner, stream))
    return y


@torch.no_grad()
def amax_qdq_int_group(inputs: torch.Tensor, group_size: int, num_bits: int = 4, unsigned: bool = False,
                       narrow_range: bool = False, return_amax: bool = True):
    """Fused dynamic per-group amax + INT-k QDQ over the (-1, g) view of a contiguous tensor whose
    element count is a multiple of g (the caller pads the last dim like _process_for_blockquant,
    tensor_quantizer.py:1045-1055).  Returns (y, amax[n/g] fp32 or None)."""
    _require_gpu(inputs, "amax_qdq_int_group")
    x = inputs.contiguous()
    if x.numel() % group_size:
        raise MoquantError("numel must be a multiple of group_size (pad the last dim first)")
    ng = x.numel() // group_size
    y = torch.empty_like(x)
    am = torch.empty(ng, dtype=torch.float32, device=x.device) if return_amax else None
    with _on(x) as stream:
        check(_lib.lib().moq_amax_qdq_int_group(_p(x), _p(y), _p(am), ng, int(group_size), _dt(x),
                                                int(num_bits), int(unsigned), int(narrow_range), stream))
    return y, am


_MX_FORMAT_MAP = {(4, 3): "E4M3", (5, 2): "E5M2", (3, 2): "E3M2", (2, 3): "E2M3", 8: "INT8", (8, 0): "E8M0",
                  (2, 1): "E2M1", (1, 2): "E1M2", (0, 3): "E0M3", (3, 0): "E3M0"}  # tensor_quant.py:30-41


@torch.no_grad()
def fused_amax_convert(inputs: torch.Tensor, block_size: int, fmt: str | int, scale_fmt: str | int = "E8M0",
                       global_amax: torch.Tensor | None = None, out: torch.Tensor | None = None) -> torch.Tensor:
    """MX dynamic block QDQ along the last dim -- cuda_ext_mx.fused_amax_convert
    (tensor_quant_mx.cu:355-387)."""
    _require_gpu(inputs, "fused_amax_convert")
    x = inputs.contiguous()
    f = _lib.MX_TYPES[fmt] if isinstance(fmt, str) else int(fmt)
    sf = _lib.MX_TYPES[scale_fmt] if isinstance(scale_fmt, str) else int(scale_fmt)
    cols = x.shape[-1] if x.dim() else 1
    rows = x.numel() // max(cols, 1)
    y = torch.empty_like(x) if out is None else out  # out may be x itself (in place)
    ga = None if global_amax is None else _f32(global_amax, x.device)
    if ga is not None and ga.numel() != 1:
        # the reference's per-channel variant (global_amax.numel() == rows) indexes the amax with the PADDED flat
        # index divided by the unpadded width (tensor_quant_mx.cu:277) and reads past the end for the last rows;
        # its Python front door always reduces to one value first (tensor_quant.py:171-172)
        raise MoquantUnsupported("fused_amax_convert: global_amax must have one element")
    with _on(x) as stream:
        check(_lib.lib().moq_mx_fused_amax_convert(_p(x), _p(y), rows, cols, int(block_size), _dt(x), f, sf,
                                                   _p(ga), stream))
    return y


@torch.no_grad()
def convert_to_exmy(x, fmt: str | int):
    """cuda_ext_mx.convert_to_exmy(x, format): round to the element format's grid, no scale.  A Python float gives a
    Python float (the pybind surface); a GPU tensor is converted elementwise (fp32 in / out)."""
    f = _lib.MX_TYPES[fmt] if isinstance(fmt, str) else int(fmt)
    scalar = not isinstance(x, torch.Tensor)
    t = torch.tensor([float(x)], dtype=torch.float32, device="cuda") if scalar else x.detach().float().contiguous()
    _require_gpu(t, "convert_to_exmy")
    y = torch.empty_like(t)
    with _on(t) as stream:
        check(_lib.lib().moq_mx_convert(_p(t), _p(y), t.numel(), f, stream))
    return float(y.item()) if scalar else y


@torch.no_grad()
def dynamic_block_quant(inputs, block_size, amax, num_bits, scale_bits):
    """tensor_quant.dynamic_block_quant front door (tensor_quant.py:157-195): num_bits / scale_bits are
    (E, M) tuples or 8."""
    if num_bits not in _MX_FORMAT_MAP or scale_bits not in _MX_FORMAT_MAP:
        raise NotImplementedError(f"Unsupported num_bits: {num_bits}, scale_bits: {scale_bits}")
    return fused_amax_convert(inputs, block_size, _MX_FORMAT_MAP[num_bits], _MX_FORMAT_MAP[scale_bits],
                              None if scale_bits == (8, 0) else amax)


# ----------------------------------------------------------------------------------------------- histogram
@torch.no_grad()
def hist_abs(x: torch.Tensor, bins: int, max_edge: float, skip_zeros: bool = False,
             counts: torch.Tensor | None = None) -> torch.Tensor:
    """counts[b] += #|x| in bin b of torch.histc(|x|, bins, 0, max_edge); int64 counts on device."""
    _require_gpu(x, "hist_abs")
    x = x.detach().contiguous()
    if counts is None:
        counts = torch.zeros(bins, dtype=torch.int64, device=x.device)
    elif counts.dtype != torch.int64 or counts.numel() != bins or not counts.is_contiguous():
        raise MoquantError("counts must be a contiguous int64 tensor with `bins` entries")
    with _on(x) as stream:
        check(_lib.lib().moq_hist_abs(_p(x), x.numel(), _dt(x), _p(counts), int(bins), float(max_edge),
                                      int(skip_zeros), stream))
    return counts


INPUT_QUANT_HIST_BINS = (1, 16384)  # bin counts the histogram stage of moq_input_quant takes


@torch.no_grad()
def input_quant(x: torch.Tensor, pre_quant_scale: torch.Tensor | None = None, amax_running: torch.Tensor | None = None,
                qdq_amax: torch.Tensor | None = None, num_bits=None, unsigned: bool = False, narrow_range: bool = False,
                hist_counts: torch.Tensor | None = None, hist_max_edge: float = 0.0, hist_skip_zeros: bool = False,
                out: torch.Tensor | None = None) -> torch.Tensor | None:
    """The per-tensor input-quantizer pass of TensorQuantizer.forward in one read of x (nn/modules/tensor_quantizer.py:
    1119-1221): optional `x * pre_quant_scale` in x.dtype, running abs-max into `amax_running` (fp32 [1]), |.| histogram
    into `hist_counts` (int64 [bins], range [0, hist_max_edge]) and INT-k (num_bits int) / FP8-E4M3 (num_bits (4, 3))
    quantize-dequantize with `qdq_amax`.  Returns the output tensor (the scaled and / or fake-quantized activation) or
    None when only statistics were asked for."""
    _require_gpu(x, "input_quant")
    xc = x.detach()
    if not xc.is_contiguous():
        xc = xc.contiguous()
    cols = xc.shape[-1]
    rows = xc.numel() // max(cols, 1)
    dev = xc.device
    fmt = 0
    if qdq_amax is not None:
        if isinstance(num_bits, int):
            fmt = 1
        elif num_bits is not None and tuple(num_bits) == (4, 3):
            fmt = 2
        else:
            raise MoquantUnsupported(f"input_quant: format {num_bits}")
        qa = _f32(qdq_amax, dev).reshape(-1)
        if qa.numel() != 1:
            raise MoquantUnsupported("input_quant: per-tensor amax only")
    else:
        qa = None
    pqs = None
    if pre_quant_scale is not None:
        if pre_quant_scale.numel() != cols:
            raise MoquantError("input_quant: pre_quant_scale must have one entry per column")
        pqs = _f32(pre_quant_scale.to(xc.dtype), dev).reshape(-1).contiguous()
    if amax_running is not None and (amax_running.dtype != torch.float32 or amax_running.numel() != 1
                                     or amax_running.device != dev):
        raise MoquantError("input_quant: amax_running must be a 1-element fp32 tensor on x's device")
    bins = 0
    if hist_counts is not None:
        bins = hist_counts.numel()
        if hist_counts.dtype != torch.int64 or not hist_counts.is_contiguous() or hist_counts.device != dev:
            raise MoquantError("input_quant: hist_counts must be a contiguous int64 tensor on x's device")
    y = None
    if fmt or pqs is not None:
        y = out if out is not None else torch.empty_like(xc)
        if y.shape != xc.shape or y.dtype != xc.dtype or not y.is_contiguous():
            raise MoquantError("input_quant: `out` must be contiguous with x's shape and dtype")
    nb = int(num_bits) if fmt == 1 else 0
    with _on(xc) as stream:
        check(_lib.lib().moq_input_quant(_p(xc), _p(pqs), _p(y), rows, cols, _dt(xc), _p(amax_running), _p(qa), fmt, nb,
                                         int(bool(unsigned)), int(bool(narrow_range)), _p(hist_counts), int(bins),
                                         float(hist_max_edge), int(bool(hist_skip_zeros)), stream))
    return y


@torch.no_grad()
def row_hist_np(w: torch.Tensor, bins: int):
    """Per-row histograms of |w| with np.histogram(a, bins, range=(0, a.max())) semantics (calibrate_weights,
    calib/histogram.py:346-433): returns (counts int32 [rows, bins], edges fp32 [rows, bins + 1]).  A 1-D / flattened
    call (rows = 1) is the per-tensor histogram.  16-bit inputs are binned as fp32 values (numpy itself has no bf16
    and would bin fp16 against fp16 edges)."""
    _require_gpu(w, "row_hist_np")
    x = w.detach().contiguous()
    x = x.reshape(1, -1) if x.dim() < 2 else x.reshape(x.shape[0], -1)
    rows, cols = x.shape
    mx = reduce_amax(x, axis=[1]).float().reshape(-1) if cols else torch.zeros(rows, device=x.device)
    zero = mx == 0
    first = torch.where(zero, torch.full_like(mx, -0.5), torch.zeros_like(mx)).contiguous()  # _get_outer_edges
    last = torch.where(zero, torch.full_like(mx, 0.5), mx).contiguous()
    counts = torch.empty(rows, bins, dtype=torch.int32, device=x.device)  # (the call clears what it accumulates into)
    with _on(x) as stream:
        check(_lib.lib().moq_row_hist_np(_p(x), rows, cols, _dt(x), int(bins), _p(first), _p(last), _p(counts), stream))
    # np.linspace in float32: fp32(fp32(k * step) + first), last edge = stop
    k = torch.arange(bins + 1, dtype=torch.float32, device=x.device)
    step = (last - first) / torch.full_like(last, float(bins))  # tensor / tensor: a true division on the GPU too
    edges = k[None, :] * step[:, None] + first[:, None]
    edges[:, -1] = last
    return counts, edges


@torch.no_grad()
def hist_entropy_divergences(hist: torch.Tensor, num_quant_bins: int, start_bin: int = 128, stride: int = 1) -> torch.Tensor:
    """KL divergence of every clipping candidate of the entropy threshold search (calib/histogram.py:210-283) over ONE
    collected histogram (int64 [bins], on the GPU): fp64 [n_candidates], candidate c clips after source bin
    start_bin + c * stride - 1.  Nothing is read back here."""
    _require_gpu(hist, "hist_entropy_divergences")
    h = hist.detach().reshape(-1).to(torch.int64).contiguous()
    n_cand = max(0, (h.numel() - start_bin) // stride + 1) if h.numel() >= start_bin else 0
    out = torch.empty(n_cand, dtype=torch.float64, device=h.device)
    if n_cand:
        with _on(h) as stream:
            check(_lib.lib().moq_hist_entropy(_p(h), h.numel(), int(num_quant_bins), int(start_bin), int(stride), _p(out),
                                              stream))
    return out


@torch.no_grad()
def hist_percentile_index(hist: torch.Tensor, q: float) -> torch.Tensor:
    """np.searchsorted(np.cumsum(hist[r] / hist[r].sum()), q) for every row of an int32 / int64 [rows, bins] histogram on
    the GPU (calib/histogram.py:326-343, :400-412): int64 [rows], bit-exact (sequential fp64 running sums)."""
    _require_gpu(hist, "hist_percentile_index")
    h = hist.detach()
    if h.dtype not in (torch.int32, torch.int64):
        h = h.to(torch.int64)
    h = h.reshape(1, -1) if h.dim() < 2 else h.reshape(h.shape[0], -1)
    h = h.contiguous()
    idx = torch.empty(h.shape[0], dtype=torch.int64, device=h.device)
    with _on(h) as stream:
        check(_lib.lib().moq_hist_percentile(_p(h), h.element_size(), h.shape[0], h.shape[1], float(q), _p(idx), stream))
    return idx


# ----------------------------------------------------------------------------------------------- sparsity
@torch.no_grad()
def mask_2to4(w: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """2:4 magnitude mask over groups of 4 along the last dim of a contiguous 2-D view."""
    _require_gpu(w, "mask_2to4")
    x = w.detach().contiguous()
    cols = x.shape[-1]
    rows = x.numel() // cols
    mask = torch.empty(x.shape, dtype=torch.uint8, device=x.device) if out is None else out.view(torch.uint8)
    with _on(x) as stream:
        check(_lib.lib().moq_mask_2to4(_p(x), rows, cols, _dt(x), _p(mask), stream))
    return mask.view(torch.bool)


# ----------------------------------------------------------------------------------------------- INT4
@torch.no_grad()
def int4_quantize(flat_input: torch.Tensor, scales: torch.Tensor, block_size: int,
                  rounding: int = _lib.ROUND_HALF_EVEN) -> torch.Tensor:
    """cuda_ext.INT4_quantize(input, scales, block_size) -> uint8[n/2] (tensor_quant_gpu.cu:342-366)."""
    _require_gpu(flat_input, "INT4_quantize")
    x = flat_input.contiguous()
    s = scales.to(x.dtype).contiguous()
    out = torch.empty(x.numel() // 2, dtype=torch.uint8, device=x.device)
    with _on(x) as stream:
        check(_lib.lib().moq_int4_pack(_p(x), _p(s), _p(out), x.numel(), int(block_size), _dt(x),
                                       int(rounding), stream))
    return out


@torch.no_grad()
def int4_dequantize(quantized: torch.Tensor, scales: torch.Tensor, block_size: int) -> torch.Tensor:
    """cuda_ext.INT4_dequantize(uint8, scales, block_size) -> scales.dtype[2n] (tensor_quant_gpu.cu:283-308)."""
    _require_gpu(quantized, "INT4_dequantize")
    q = quantized.contiguous().view(-1)
    if q.dtype != torch.uint8:
        raise MoquantError("quantized data must be uint8")
    s = scales.contiguous()
    out = torch.empty(2 * q.numel(), dtype=s.dtype, device=q.device)
    with _on(q) as stream:
        check(_lib.lib().moq_int4_unpack(_p(q), _p(s), _p(out), q.numel(), int(block_size), _dt(s), stream))
    return out


@torch.no_grad()
def pack_int4_in_uint8(weight: torch.Tensor, weights_scaling_factor: torch.Tensor) -> torch.Tensor:
    """Checkpoint packer (export/quant_utils.py:792-833): uint8 [..., out/2, in]."""
    _require_gpu(weight, "pack_int4_in_uint8")
    out_dim, in_dim = weight.shape[-2], weight.shape[-1]
    if out_dim % 2:
        raise AssertionError(f"Cannot pack weight. Out dimension {out_dim} is not an even number.")
    g = in_dim // weights_scaling_factor.shape[-1]
    w = weight.contiguous()
    wsf = _f32(weights_scaling_factor, w.device)
    lead = w.shape[:-2]
    batch = 1
    for d in lead:
        batch *= d
    out = torch.empty(*lead, out_dim // 2, in_dim, dtype=torch.uint8, device=w.device)
    w3, s3, o3 = w.reshape(batch, out_dim, in_dim), wsf.reshape(batch, out_dim, -1), out.view(batch, out_dim // 2, in_dim)
    with _on(w) as stream:
        for b in range(batch):  # experts: one launch each
            check(_lib.lib().moq_int4_pack_export(_p(w3[b]), _p(s3[b]), _p(o3[b]), out_dim, in_dim, int(g),
                                                  _dt(w), stream))
    return out


# ----------------------------------------------------------------------------------------------- AWQ / smooth
@torch.no_grad()
def int8_pack_rows(weight: torch.Tensor, weights_scaling_factor: torch.Tensor) -> torch.Tensor:
    """(weight / wsf[:, None]).round().clamp(-128, 127).to(int8) with an fp32 scaling factor per output channel
    (export/quant_utils.py:868-869), one kernel."""
    _require_gpu(weight, "int8_pack_rows")
    w = weight.detach().contiguous()
    rows, cols = w.shape
    wsf = _f32(weights_scaling_factor, w.device).reshape(-1)
    if wsf.numel() == 1 and rows != 1:
        wsf = wsf.expand(rows).contiguous()  # a per-tensor amax: `wsf[:, None]` of a [1] factor broadcasts over the rows
    if wsf.numel() != rows:
        raise MoquantError("int8_pack_rows: one scaling factor per output channel expected")
    vec = 4 if w.dtype == torch.float32 else 8
    if cols % vec:
        return (w / wsf[:, None]).round().clamp(-128, 127).to(torch.int8)  # ragged rows: the reference's own ops
    out = torch.empty(rows, cols, dtype=torch.int8, device=w.device)
    with _on(w) as stream:
        check(_lib.lib().moq_int8_pack_rows(_p(w), _p(wsf), _p(out), rows, cols, _dt(w), stream))
    return out


@torch.no_grad()
def scale_cols(weight: torch.Tensor, scale: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """(W * s_fp32[None, :]).to(W.dtype) -- _apply_weight_pre_quant_scale (model_calib.py:1208-1216)."""
    _require_gpu(weight, "scale_cols")
    w = weight.contiguous()
    s = _f32(scale, w.device).reshape(-1)
    cols = w.shape[-1]
    if s.numel() != cols:
        raise MoquantError("scale length must equal the last weight dim")
    y = torch.empty_like(w) if out is None else out
    with _on(w) as stream:
        check(_lib.lib().moq_scale_cols(_p(w), _p(s), _p(y), w.numel() // cols, cols, _dt(w), stream))
    return y


@torch.no_grad()
def awq_scale_qdq(weight: torch.Tensor, awq_scale: torch.Tensor, group_size: int, num_bits: int = 4,
                  out: torch.Tensor | None = None) -> torch.Tensor:
    """QDQ_int_g((W * s).to(W.dtype)) with dynamic per-group amax: the weight side of one AWQ-lite search
    step (model_calib.py:1552-1554), one read + one write of W."""
    _require_gpu(weight, "awq_scale_qdq")
    w = weight.contiguous()
    s = awq_scale.detach().to(device=w.device, dtype=w.dtype).contiguous().reshape(-1)
    cols = w.shape[-1]
    y = torch.empty_like(w) if out is None else out
    with _on(w) as stream:
        check(_lib.lib().moq_awq_scale_qdq(_p(w), _p(s), _p(y), w.numel() // cols, cols, int(group_size),
                                           _dt(w), int(num_bits), stream))
    return y


@torch.no_grad()
def col_abs_stats(x: torch.Tensor, sum_out: torch.Tensor | None = None, amax_out: torch.Tensor | None = None,
                  accumulate: bool = False, want_sum: bool = True, want_amax: bool = True):
    """Column sum of |x| (fp32) and column abs-max of x[tokens, cols] in one read."""
    _require_gpu(x, "col_abs_stats")
    x2 = x.detach().contiguous().view(-1, x.shape[-1])
    tokens, cols = x2.shape
    dev = x2.device
    if want_sum and sum_out is None:
        sum_out = torch.zeros(cols, dtype=torch.float32, device=dev)
    if want_amax and amax_out is None:
        amax_out = torch.zeros(cols, dtype=torch.float32, device=dev)
    ws = None
    if want_sum:
        ws = torch.empty(max(int(_lib.lib().moq_col_stats_workspace(tokens, cols)), 1), dtype=torch.float32,
                         device=dev)
    with _on(x2) as stream:
        check(_lib.lib().moq_col_abs_stats(_p(x2), tokens, cols, _dt(x2), _p(sum_out if want_sum else None),
                                           _p(amax_out if want_amax else None), _p(ws), int(accumulate),
                                           stream))
    return sum_out, amax_out


@torch.no_grad()
def col_abs_mean_accum(x: torch.Tensor, acc: torch.Tensor) -> torch.Tensor:
    """acc += x.abs().mean(0).to(float32) with the mean rounded to x.dtype like the reference's get_act_scale
    (model_calib.py:1471-1472) -- one read of the batch, the division and the rounding inside the kernel (IEEE
    division: torch's GPU `tensor / python_scalar` multiplies by the reciprocal, which is not the CPU result)."""
    _require_gpu(x, "col_abs_mean_accum")
    x2 = x.detach().contiguous().view(-1, x.shape[-1])
    tokens, cols = x2.shape
    if acc.dtype != torch.float32 or acc.numel() != cols or not acc.is_contiguous() or acc.device != x2.device:
        raise MoquantError("col_abs_mean_accum: acc must be a contiguous fp32 [cols] tensor on the batch's device")
    vec = 4 if x2.dtype == torch.float32 else 8
    if cols % vec or x2.data_ptr() % 16:
        ssum, _ = col_abs_stats(x2, want_amax=False)
        acc += torch.div(ssum, torch.full((), float(tokens), device=x2.device)).to(x2.dtype).float()
        return acc
    ws = torch.empty(max(int(_lib.lib().moq_col_stats_workspace(tokens, cols)), 1), dtype=torch.float32,
                     device=x2.device)
    with _on(x2) as stream:
        check(_lib.lib().moq_col_abs_mean_accum(_p(x2), tokens, cols, _dt(x2), _p(acc), _p(ws), stream))
    return acc


@torch.no_grad()
def awq_weight_scale(weight: torch.Tensor, group_size: int) -> torch.Tensor:
    """get_weight_scale (model_calib.py:1453-1469): fp32 [Cin] = mean over Cout of |W| / (group amax + tiny),
    computed in W.dtype like the reference.  One read of W."""
    _require_gpu(weight, "awq_weight_scale")
    w = weight.detach().contiguous()
    rows, cols = w.shape
    out = torch.empty(cols, dtype=torch.float32, device=w.device)
    ws = torch.empty(max(int(_lib.lib().moq_col_stats_workspace(rows, cols)), 1), dtype=torch.float32,
                     device=w.device)
    with _on(w) as stream:
        check(_lib.lib().moq_awq_weight_scale(_p(w), rows, cols, int(group_size), _dt(w), _p(out), _p(ws),
                                              stream))
    return out


def _axis_layout(inputs: torch.Tensor, amax: torch.Tensor, axis: int):
    axis = axis % inputs.dim()
    if amax.numel() != inputs.shape[axis]:
        raise MoquantError(f"amax.numel()={amax.numel()} != inputs.size({axis})={inputs.shape[axis]}")  # TORCH_CHECK
    inner = 1
    for d in range(axis + 1, inputs.dim()):
        inner *= inputs.shape[d]
    return inputs.shape[axis], inner


@torch.no_grad()
def fake_tensor_quant_with_axis(inputs, amax, axis, num_bits=8, unsigned=False, narrow_range=True):
    """cuda_ext.fake_tensor_quant_with_axis (tensor_quant.cpp:54-61): amax is 1-D of length inputs.size(axis)."""
    _require_gpu(inputs, "fake_tensor_quant_with_axis")
    x = inputs.contiguous()
    am = _f32(amax, x.device).reshape(-1)
    axis_size, inner = _axis_layout(x, am, axis)
    y = torch.empty_like(x)
    with _on(x) as stream:
        check(_lib.lib().moq_fake_quant_int(_p(x), _p(y), x.numel(), _dt(x), _p(am), _lib.AMAX_AXIS, axis_size,
                                            inner, int(num_bits), int(unsigned), int(narrow_range), stream))
    return y


@torch.no_grad()
def fake_e4m3fy_with_axis(inputs, amax, axis):
    """cuda_ext_fp8.fake_e4m3fy_with_axis (tensor_quant_gpu_fp8.cu:66-86)."""
    _require_gpu(inputs, "fake_e4m3fy_with_axis")
    x = inputs.contiguous()
    am = _f32(amax, x.device).reshape(-1)
    axis_size, inner = _axis_layout(x, am, axis)
    y = torch.empty_like(x)
    with _on(x) as stream:
        check(_lib.lib().moq_fake_quant_e4m3(_p(x), _p(y), x.numel(), _dt(x), _p(am), _lib.AMAX_AXIS, axis_size,
                                             inner, stream))
    return y


# ----------------------------------------------------------------------------------------------- AWQ error GEMM
def mfma_gemm_supported(x: torch.Tensor, w: torch.Tensor) -> bool:
    """Shapes / dtypes the MFMA kernels take: bf16 / f16 on the 16-bit loop, fp32 on the fp32 matrix cores
    (moq_gemm_f32.hip).  What is left for the library: odd widths."""
    return (x.dtype in (torch.bfloat16, torch.float16, torch.float32) and w.dtype == x.dtype and w.shape[-1] % 8 == 0
            and w.shape[0] % 4 == 0)


@torch.no_grad()
def awq_err_gemm(xs: torch.Tensor, w_hat: torch.Tensor, out_actual: torch.Tensor, bias: torch.Tensor | None,
                 loss_acc: torch.Tensor) -> torch.Tensor:
    """loss_acc += mean((linear(xs, w_hat, bias) - out_actual).float() ** 2) without materialising the output:
    update_loss of awq_lite (model_calib.py:1489-1495) fused into the MFMA contraction.  loss_acc: fp32 [1]."""
    _require_gpu(xs, "awq_err_gemm")
    x2 = xs.detach().contiguous().view(-1, xs.shape[-1])
    w = w_hat.detach().contiguous()
    ref = out_actual.detach().contiguous().view(-1, w.shape[0])
    if ref.shape[0] != x2.shape[0] or w.shape[1] != x2.shape[1] or ref.dtype != x2.dtype or w.dtype != x2.dtype:
        raise MoquantError("awq_err_gemm: shape / dtype mismatch between xs, w_hat and out_actual")
    if loss_acc.dtype != torch.float32 or loss_acc.numel() != 1 or not loss_acc.is_cuda:
        raise MoquantError("awq_err_gemm: loss_acc must be a 1-element fp32 GPU tensor")
    b = None if bias is None else bias.detach().to(x2.dtype).contiguous()
    tokens, cin = x2.shape
    cout = w.shape[0]
    ws = torch.empty(int(_lib.lib().moq_awq_err_gemm_workspace(tokens, cout)), dtype=torch.float32, device=x2.device)
    with _on(x2) as stream:
        check(_lib.lib().moq_awq_err_gemm(_p(x2), _p(w), _p(ref), _p(b), tokens, cout, cin, _dt(x2), _p(ws),
                                          _p(loss_acc), stream))
    return loss_acc


@torch.no_grad()
def gemm_nt(x: torch.Tensor, w: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """F.linear(x, w, bias) for bf16 / f16 / fp32 on the MFMA main loops (store epilogue)."""
    _require_gpu(x, "gemm_nt")
    x2 = x.detach().contiguous().view(-1, x.shape[-1])
    wc = w.detach().contiguous()
    b = None if bias is None else bias.detach().to(x2.dtype).contiguous()
    out = torch.empty(x2.shape[0], wc.shape[0], dtype=x2.dtype, device=x2.device)
    with _on(x2) as stream:
        check(_lib.lib().moq_gemm_nt(_p(x2), _p(wc), _p(b), _p(out), x2.shape[0], wc.shape[0], x2.shape[1],
                                     _dt(x2), stream))
    return out.view(*x.shape[:-1], wc.shape[0])


@torch.no_grad()
def scale_cols_multi(x: torch.Tensor, scales: torch.Tensor) -> torch.Tensor:
    """y[a] = (x * scales[a]).to(x.dtype) for every row a of the fp32 matrix `scales` [A, cols]: one read of x,
    A writes -- the pre-scaled inputs of all AWQ candidates at once."""
    _require_gpu(x, "scale_cols_multi")
    x2 = x.detach().contiguous().view(-1, x.shape[-1])
    s = _f32(scales, x2.device)
    if s.dim() != 2 or s.shape[1] != x2.shape[1]:
        raise MoquantError("scale_cols_multi: scales must be [A, cols]")
    y = torch.empty(s.shape[0], x2.shape[0], x2.shape[1], dtype=x2.dtype, device=x2.device)
    with _on(x2) as stream:
        check(_lib.lib().moq_scale_cols_multi(_p(x2), _p(s), _p(y), x2.shape[0], x2.shape[1], s.shape[0], _dt(x2),
                                              stream))
    return y


@torch.no_grad()
def awq_err_gemm_multi(xs: torch.Tensor, w_hat: torch.Tensor, out_actual: torch.Tensor,
                       bias: torch.Tensor | None, loss_acc: torch.Tensor) -> torch.Tensor:
    """All candidates of one linear in one launch: xs [A, T, K] (or [T, K] shared), w_hat [A, N, K] (or [N, K]
    shared); loss_acc[a] += mean((linear(xs[a], w_hat[a], bias) - out_actual).float() ** 2).  fp32 [A]."""
    _require_gpu(xs, "awq_err_gemm_multi")
    n_cand = loss_acc.numel()
    x3 = xs.detach().contiguous()
    w3 = w_hat.detach().contiguous()
    tokens, cin = x3.shape[-2], x3.shape[-1]
    cout = w3.shape[-2]
    x_stride = tokens * cin if x3.dim() == 3 else 0
    w_stride = cout * cin if w3.dim() == 3 else 0
    if (x3.dim() == 3 and x3.shape[0] != n_cand) or (w3.dim() == 3 and w3.shape[0] != n_cand):
        raise MoquantError("awq_err_gemm_multi: leading dim of xs / w_hat must equal loss_acc.numel()")
    ref = out_actual.detach().contiguous().view(-1, cout)
    if ref.shape[0] != tokens or w3.shape[-1] != cin or ref.dtype != x3.dtype or w3.dtype != x3.dtype:
        raise MoquantError("awq_err_gemm_multi: shape / dtype mismatch between xs, w_hat and out_actual")
    if loss_acc.dtype != torch.float32 or loss_acc.device != x3.device or not loss_acc.is_contiguous():
        raise MoquantError("awq_err_gemm_multi: loss_acc must be a contiguous fp32 tensor on the inputs' device")
    b = None if bias is None else bias.detach().to(x3.dtype).contiguous()
    ws = torch.empty(n_cand * int(_lib.lib().moq_awq_err_gemm_workspace(tokens, cout)), dtype=torch.float32,
                     device=x3.device)
    with _on(x3) as stream:
        check(_lib.lib().moq_awq_err_gemm_multi(_p(x3), _p(w3), _p(ref), _p(b), tokens, cout, cin, _dt(x3), n_cand,
                                                x_stride, w_stride, _p(ws), _p(loss_acc), stream))
    return loss_acc


# ----------------------------------------------------------------------------------------------- MSE sweep
@torch.no_grad()
def mse_sweep(x: torch.Tensor, cand_amax: torch.Tensor, reduce_axis, num_bits=8, unsigned: bool = False,
              narrow_range: bool = False, loss: torch.Tensor | None = None) -> torch.Tensor:
    """loss[k, ...] (+)= sum over `reduce_axis` of (x - QDQ(x, cand_amax[k]))^2 for all K candidates in one read
    of x -- the body of MseCalibrator.collect (calib/mse.py:99-113).  cand_amax: [K, C] (C = kept elements, 1 for
    per-tensor); num_bits int -> INT-k, (4, 3) -> FP8-E4M3.  Returns fp32 [K, C]."""
    _require_gpu(x, "mse_sweep")
    xc = x.detach().contiguous()
    nd = xc.dim()
    if reduce_axis is None:
        outer, kept, inner = 1, 1, xc.numel()
    else:
        if isinstance(reduce_axis, int):
            reduce_axis = (reduce_axis,)
        if len({a % nd for a in reduce_axis}) == nd:
            outer, kept, inner = 1, 1, xc.numel()
        else:
            outer, kept, inner, _ = _reduce_layout(list(xc.shape), reduce_axis)
    cand = _f32(cand_amax, xc.device).reshape(cand_amax.shape[0], -1)
    if cand.shape[1] != kept:
        raise MoquantError(f"mse_sweep: cand_amax has {cand.shape[1]} entries per candidate, layout keeps {kept}")
    k = cand.shape[0]
    if isinstance(num_bits, int):
        fp8, bits = 0, num_bits
    elif tuple(num_bits) == (4, 3):
        fp8, bits = 1, 8
    else:
        raise MoquantUnsupported(f"mse_sweep: num_bits {num_bits} not supported")
    accumulate = loss is not None
    if loss is None:
        loss = torch.empty(k, kept, dtype=torch.float32, device=xc.device)
    n_ws = int(_lib.lib().moq_mse_sweep_workspace(outer, kept, inner, k))
    ws = torch.empty(max(n_ws, 1), dtype=torch.float32, device=xc.device)
    with _on(xc) as stream:
        check(_lib.lib().moq_mse_sweep(_p(xc), outer, kept, inner, _dt(xc), _p(cand), k, _p(loss), _p(ws),
                                       int(accumulate), fp8, int(bits), int(unsigned), int(narrow_range), stream))
    return loss


@torch.no_grad()
def rescale_cols(weight: torch.Tensor, mul: torch.Tensor, div: torch.Tensor, out: torch.Tensor | None = None):
    """(W.float() * mul.float() / div.float()).to(W.dtype) per column -- _update_pre_quant_scale of the export
    resmooth step (export/quant_utils.py:1285-1296)."""
    _require_gpu(weight, "rescale_cols")
    w = weight.contiguous()
    cols = w.shape[-1]
    m, d = _f32(mul, w.device).reshape(-1), _f32(div, w.device).reshape(-1)
    if m.numel() != cols or d.numel() != cols:
        raise MoquantError("rescale_cols: mul / div length must equal the last weight dim")
    y = torch.empty_like(w) if out is None else out
    with _on(w) as stream:
        check(_lib.lib().moq_rescale_cols(_p(w), _p(m), _p(d), _p(y), w.numel() // cols, cols, _dt(w), stream))
    return y


# ----------------------------------------------------------------------------------------------- AWQ clip
@torch.no_grad()
def awq_clip_loss(inputs: torch.Tensor, weight: torch.Tensor, w_amax: torch.Tensor, shrinks: torch.Tensor,
                  block_size: int, num_bits: int, loss: torch.Tensor, token_step: int = 1) -> torch.Tensor:
    """loss[k, b, r] += mean_t (cur_k - org)^2 for every clip ratio k -- _clip_search's block branch
    (model_calib.py:1817-1868) in one pass over the weight.  inputs [T, Cin] (rows inputs[0::token_step] are
    used, :1820), weight [Cout, Cin], w_amax [Cout * nblk] in the dtype the reference's w_amax has, shrinks fp32
    [K] on device, loss fp32 [K, nblk, Cout] (block-major; use .transpose(1, 2) for the reference's layout)."""
    _require_gpu(weight, "awq_clip_loss")
    x2 = inputs.detach().reshape(-1, inputs.shape[-1])
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    w = weight.detach().contiguous()
    cout, cin = w.shape
    nblk = -(-cin // block_size)
    n_tok = -(-x2.shape[0] // token_step)
    if x2.dtype != w.dtype or x2.shape[1] != cin:
        raise MoquantError("awq_clip_loss: inputs / weight mismatch")
    adt = _lib.F32 if w_amax.dtype == torch.float32 else _dt(w_amax)
    am = _f32(w_amax, w.device).reshape(-1)
    if am.numel() != cout * nblk:
        raise MoquantError(f"awq_clip_loss: w_amax has {am.numel()} entries, expected {cout * nblk}")
    if (loss.dtype != torch.float32 or not loss.is_contiguous()
            or tuple(loss.shape) != (shrinks.numel(), nblk, cout)):
        raise MoquantError("awq_clip_loss: loss must be contiguous fp32 [n_shrink, nblk, cout]")
    sh = shrinks.to(device=w.device, dtype=torch.float32).contiguous()
    with _on(w) as stream:
        check(_lib.lib().moq_awq_clip_loss(_p(x2), n_tok, token_step * cin, _p(w), cout, cin, int(block_size),
                                           _dt(w), _p(am), adt, _p(sh), sh.numel(), int(num_bits), _p(loss),
                                           stream))
    return loss


# ----------------------------------------------------------------------------------------------- real FP8 / MXFP4
def _scale_layout(x: torch.Tensor, scales: torch.Tensor):
    """(mode, axis_size, inner) of a scale tensor over contiguous x: one scale, or one scale per run of `inner`
    consecutive elements (per-channel rows of a 2-D weight, 1-D blocks along the last dim)."""
    ns = scales.numel()
    if ns == 1:
        return _lib.AMAX_SCALAR, 1, 1
    if x.numel() % ns:
        raise MoquantUnsupported("scale count does not divide the element count")
    inner = x.numel() // ns
    if x.shape[-1] % inner and inner % x.shape[-1]:
        raise MoquantUnsupported("scales must run along the flattened last dims (per-row or last-dim blocks)")
    return _lib.AMAX_AXIS, ns, inner


@torch.no_grad()
def fp8_quantize(inputs: torch.Tensor, scales: torch.Tensor, fp32_scales: bool = False) -> torch.Tensor:
    """(inputs / scales).to(torch.float8_e4m3fn) -- FP8QTensor.quantize's cast (fp8_tensor.py:103-107) / the FP8
    branch of to_quantized_weight.  scales: 1 element, per row, or per last-dim block (row-major order); they are
    used in the tensor dtype (FP8QTensor) or, with fp32_scales, in fp32 (export: the quotient of a 16-bit weight and
    an fp32 scaling factor is still rounded to the weight dtype before the cast)."""
    _require_gpu(inputs, "fp8_quantize")
    x = inputs.detach().contiguous()
    sdt = torch.float32 if fp32_scales else x.dtype
    s = scales.detach().to(device=x.device, dtype=sdt).contiguous().reshape(-1)
    mode, axis_size, inner = _scale_layout(x, s)
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    try:
        with _on(x) as stream:
            check(_lib.lib().moq_fp8_pack(_p(x), _p(s), _DT[sdt], _p(out), x.numel(), _dt(x), mode, axis_size, inner, stream))
    except MoquantUnsupported:
        # runs shorter than a 16-byte packet / odd sizes (the reference's literal test tensors are 2 x 4): the
        # element-wise tile kernel with one tile per scale run
        if fp32_scales or (mode == _lib.AMAX_AXIS and axis_size * inner != x.numel()):
            raise
        x2 = x.reshape(1, -1) if mode == _lib.AMAX_SCALAR else x.reshape(axis_size, inner)
        return fp8_quantize_tile(x2, s, 1, x2.shape[1]).reshape(x.shape)
    return out.view(torch.float8_e4m3fn)


@torch.no_grad()
def fp8_dequantize(quantized: torch.Tensor, scales: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """quantized.to(dtype) * scales.to(dtype) -- FP8QTensor.dequantize (fp8_tensor.py:151)."""
    _require_gpu(quantized, "fp8_dequantize")
    q = quantized.detach().contiguous().view(torch.uint8)
    s = scales.detach().to(device=q.device, dtype=dtype).contiguous().reshape(-1)
    mode, axis_size, inner = _scale_layout(q, s)
    out = torch.empty(q.shape, dtype=dtype, device=q.device)
    try:
        with _on(q) as stream:
            check(_lib.lib().moq_fp8_unpack(_p(q), _p(s), _p(out), q.numel(), _dt(out), mode, axis_size, inner, stream))
    except MoquantUnsupported:
        if mode == _lib.AMAX_AXIS and axis_size * inner != q.numel():
            raise
        q2 = q.reshape(1, -1) if mode == _lib.AMAX_SCALAR else q.reshape(axis_size, inner)
        return fp8_dequantize_tile(q2, s, dtype, 1, q2.shape[1]).reshape(q.shape)
    return out


@torch.no_grad()
def fp8_quantize_tile(inputs: torch.Tensor, scales: torch.Tensor, br: int, bc: int) -> torch.Tensor:
    """(inputs / scales expanded over br x bc tiles).to(float8_e4m3fn) for a 2-D tensor -- FP8QTensor.quantize with
    blocks on both axes (fp8_tensor.py:60-112).  scales: [R/br, C/bc] in the tensor dtype (quotient rounded to it) or
    fp32 (fp32 quotient, as torch promotes)."""
    _require_gpu(inputs, "fp8_quantize_tile")
    x = inputs.detach().contiguous()
    sdt = torch.float32 if scales.dtype == torch.float32 else x.dtype
    s = scales.detach().to(device=x.device, dtype=sdt).contiguous().reshape(-1)
    rows, cols = x.shape
    if s.numel() != (rows // br) * (cols // bc):
        raise MoquantError(f"fp8_quantize_tile: {s.numel()} scales for a {rows}x{cols} tensor in {br}x{bc} tiles")
    out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    with _on(x) as stream:
        check(_lib.lib().moq_fp8_pack_tile(_p(x), _p(s), _DT[sdt], _p(out), rows, cols, int(br), int(bc), _dt(x), stream))
    return out.view(torch.float8_e4m3fn)


@torch.no_grad()
def fp8_dequantize_tile(quantized: torch.Tensor, scales: torch.Tensor, dtype: torch.dtype, br: int, bc: int):
    """quantized.to(dtype) * scales expanded over br x bc tiles (fp8_tensor.py:114-151)."""
    _require_gpu(quantized, "fp8_dequantize_tile")
    q = quantized.detach().contiguous().view(torch.uint8)
    s = scales.detach().to(device=q.device, dtype=dtype).contiguous().reshape(-1)
    rows, cols = q.shape
    out = torch.empty(q.shape, dtype=dtype, device=q.device)
    with _on(q) as stream:
        check(_lib.lib().moq_fp8_unpack_tile(_p(q), _p(s), _p(out), rows, cols, int(br), int(bc), _dt(out), stream))
    return out


@torch.no_grad()
def mxfp4_quantize(inputs: torch.Tensor, block_size: int = 32):
    """MXFP4QTensor.quantize (mxfp4_tensor.py:37-81): (uint8 [..., K/2], e8m0 uint8 [n/block, 1])."""
    _require_gpu(inputs, "mxfp4_quantize")
    x = inputs.detach().contiguous()
    if x.numel() % block_size or x.shape[-1] % 2:
        raise MoquantError("mxfp4_quantize: numel must be a multiple of block_size and the last dim even")
    nb = x.numel() // block_size
    packed = torch.empty(*x.shape[:-1], x.shape[-1] // 2, dtype=torch.uint8, device=x.device)
    e8 = torch.empty(nb, 1, dtype=torch.uint8, device=x.device)
    with _on(x) as stream:
        check(_lib.lib().moq_mxfp4_pack(_p(x), _p(packed), _p(e8), nb, int(block_size), _dt(x), stream))
    return packed, e8


@torch.no_grad()
def mxfp4_dequantize(packed: torch.Tensor, e8m0: torch.Tensor, dtype: torch.dtype, block_size: int = 32):
    """MXFP4QTensor.dequantize (mxfp4_tensor.py:83-144)."""
    _require_gpu(packed, "mxfp4_dequantize")
    p = packed.detach().contiguous()
    e = e8m0.detach().to(p.device).contiguous().reshape(-1)
    if p.dtype != torch.uint8 or e.dtype != torch.uint8 or (2 * p.numel()) != e.numel() * block_size:
        raise MoquantError("mxfp4_dequantize: packed / scale size mismatch")
    out = torch.empty(*p.shape[:-1], p.shape[-1] * 2, dtype=dtype, device=p.device)
    with _on(p) as stream:
        check(_lib.lib().moq_mxfp4_unpack(_p(p), _p(e), _p(out), e.numel(), int(block_size), _dt(out), stream))
    return out


# ----------------------------------------------------------------------------------------------- SparseGPT
@torch.no_grad()
def transpose16(x: torch.Tensor) -> torch.Tensor:
    """x.t().contiguous() for a 2-D bf16 / f16 tensor (LDS-tiled)."""
    _require_gpu(x, "transpose16")
    if x.dim() != 2 or x.element_size() != 2:
        raise MoquantUnsupported("transpose16: 2-D tensor of a 2-byte dtype expected")
    xc = x.detach().contiguous()
    y = torch.empty(xc.shape[1], xc.shape[0], dtype=xc.dtype, device=xc.device)
    with _on(xc) as stream:
        check(_lib.lib().moq_transpose16(_p(xc), _p(y), xc.shape[0], xc.shape[1], stream))
    return y


@torch.no_grad()
def hessian_accum(hessian: torch.Tensor, inputs: torch.Tensor, decay: float, scale: float,
                  upper_only: bool = False) -> torch.Tensor:
    """hessian <- hessian * decay + scale * inputs^T @ inputs (fp32 [Cin, Cin], in place) for inputs [tokens, Cin]
    in bf16 / f16: transpose + MFMA contraction (exact products, fp32 accumulation).  upper_only: update the tiles on
    and above the diagonal only (half the work); call symmetrize() once after the last batch."""
    _require_gpu(inputs, "hessian_accum")
    x2 = inputs.detach().reshape(-1, inputs.shape[-1])
    cin = x2.shape[1]
    if hessian.dtype != torch.float32 or tuple(hessian.shape) != (cin, cin) or not hessian.is_contiguous():
        raise MoquantError("hessian_accum: hessian must be a contiguous fp32 [Cin, Cin] tensor")
    if x2.dtype not in (torch.bfloat16, torch.float16) or cin % 4:
        raise MoquantUnsupported("hessian_accum: bf16 / f16 inputs with Cin % 4 == 0 run on the MFMA path")
    pad = (-x2.shape[0]) % 8  # zero tokens add nothing to X^T X
    if pad:
        x2 = torch.nn.functional.pad(x2, (0, 0, 0, pad))
    xt = transpose16(x2)
    with _on(xt) as stream:
        check(_lib.lib().moq_hessian_accum(_p(xt), cin, xt.shape[1], _dt(xt), _p(hessian), float(decay), float(scale),
                                           int(upper_only), stream))
    return hessian


class GramStage:
    """Several calibration batches per Gram launch: every batch [T, Cin] is transposed straight into a column block of
    a [Cin, slots * Tpad] staging buffer (no extra copy -- the transpose is needed anyway), and ONE hessian_accum launch
    with K = slots * Tpad replaces `slots` launches.  What it saves is the read-modify-write of the fp32 [Cin, Cin]
    matrix per launch (for Cin = 14336 and 4096 tokens a third of the launch).  Batches of another length flush the
    stage and are accumulated directly; unused columns stay zero (zero tokens add nothing to X^T X)."""

    def __init__(self, gram: torch.Tensor, tokens: int, dtype: torch.dtype, slots: int = 4):
        self.gram, self.tokens, self.slots = gram, int(tokens), int(slots)
        self.tpad = (self.tokens + 7) // 8 * 8
        self.buf = torch.zeros(gram.shape[0], self.slots * self.tpad, dtype=dtype, device=gram.device)
        self.fill = 0

    @staticmethod
    def nbytes(cin: int, tokens: int, slots: int = 4) -> int:
        return cin * slots * ((tokens + 7) // 8 * 8) * 2

    def add(self, x2: torch.Tensor):
        """G += X^T X / T for one batch (deferred until the stage is full)."""
        if x2.shape[0] != self.tokens or x2.dtype != self.buf.dtype:
            self.flush()
            hessian_accum(self.gram, x2, 1.0, 1.0 / x2.shape[0], upper_only=True)
            return
        xc = x2.detach().contiguous()
        dst = self.buf[:, self.fill * self.tpad:]
        with _on(xc) as stream:
            check(_lib.lib().moq_transpose16_ld(_p(xc), _p(dst), xc.shape[0], xc.shape[1], self.buf.shape[1], stream))
        self.fill += 1
        if self.fill == self.slots:
            self.flush()

    def flush(self):
        if self.fill == 0:
            return
        if self.fill < self.slots:
            self.buf[:, self.fill * self.tpad:].zero_()  # stale columns of an earlier, full round
        cin = self.gram.shape[0]
        with _on(self.buf) as stream:
            check(_lib.lib().moq_hessian_accum(_p(self.buf), cin, self.buf.shape[1], _dt(self.buf), _p(self.gram), 1.0,
                                               1.0 / self.tokens, 1, stream))
        self.fill = 0


@torch.no_grad()
def symmetrize(h: torch.Tensor) -> torch.Tensor:
    """h[r, c] = h[c, r] for r > c (in place): completes a matrix accumulated with hessian_accum(upper_only=True)."""
    _require_gpu(h, "symmetrize")
    if h.dtype != torch.float32 or h.dim() != 2 or h.shape[0] != h.shape[1] or not h.is_contiguous():
        raise MoquantError("symmetrize: contiguous square fp32 matrix expected")
    with _on(h) as stream:
        check(_lib.lib().moq_symmetrize(_p(h), h.shape[0], stream))
    return h


@torch.no_grad()
def sgpt_block_sweep(w: torch.Tensor, i1: int, bs: int, hinv: torch.Tensor, prune_n: int = 2,
                     prune_m: int = 4) -> torch.Tensor:
    """In-place column sweep of create_sgpt_mask over block [i1, i1 + bs) (sparsegpt.py:96-127); returns delta."""
    _require_gpu(w, "sgpt_block_sweep")
    if w.dtype != torch.float32 or hinv.dtype != torch.float32 or not w.is_contiguous() or not hinv.is_contiguous():
        raise MoquantError("sgpt_block_sweep: contiguous fp32 tensors expected")
    rows, ld = w.shape
    delta = torch.empty(rows, bs, dtype=torch.float32, device=w.device)
    with _on(w) as stream:
        check(_lib.lib().moq_sgpt_block_sweep(_p(w), rows, ld, int(i1), int(bs), _p(hinv), _p(delta), int(prune_n),
                                              int(prune_m), stream))
    return delta


@torch.no_grad()
def gptq_block_sweep(w: torch.Tensor, i1: int, bs: int, hinv: torch.Tensor, amax: torch.Tensor, amax_row_stride: int,
                     g: int, fmt: int, num_bits: int = 8, unsigned: bool = False, narrow_range: bool = False) -> torch.Tensor:
    """In-place column sweep of gptq_blockwise_update over block [i1, i1 + bs) (utils/calib_utils.py:241-276) for a
    quantizer with a calibrated amax: the block's columns become their quantized-dequantized values, the errors
    err_j = (w_j - q_j) / hinv_jj are returned [rows, bs].  fmt 1: INT-num_bits, 2: FP8-E4M3; amax: fp32, entry
    amax[r * amax_row_stride + c / g] for element (r, c).  fmt 3: MX blocks of g columns with E8M0 scales taken from the
    block's current abs-max at every column (dynamic block quantization); num_bits is the element format (name or
    moq_mx_type code), amax is not used.  fmt 4: block scales in the element format `unsigned` (name or code) relative to
    the calibrated tensor-wide `amax` (NVFP4-style two-level scales)."""
    _require_gpu(w, "gptq_block_sweep")
    if w.dtype != torch.float32 or hinv.dtype != torch.float32 or not w.is_contiguous() or not hinv.is_contiguous():
        raise MoquantError("gptq_block_sweep: contiguous fp32 tensors expected")
    rows, ld = w.shape
    if int(fmt) in (3, 4):
        num_bits = _lib.MX_TYPES[num_bits] if isinstance(num_bits, str) else int(num_bits)
        if int(fmt) == 4:  # `unsigned` carries the scale format, amax the tensor-wide value
            unsigned = _lib.MX_TYPES[unsigned] if isinstance(unsigned, str) else int(unsigned)
            am = _f32(amax, w.device).reshape(-1)[:1].contiguous()
        else:
            am = None
    else:
        am = _f32(amax, w.device).reshape(-1)
        need = (rows - 1) * int(amax_row_stride) + (ld - 1) // int(g) + 1 if rows else 0
        if am.numel() < need:
            raise MoquantError(f"gptq_block_sweep: {am.numel()} amax entries, the layout needs {need}")
    delta = torch.empty(rows, bs, dtype=torch.float32, device=w.device)
    with _on(w) as stream:
        check(_lib.lib().moq_gptq_block_sweep(_p(w), rows, ld, int(i1), int(bs), _p(hinv), _p(delta), _p(am),
                                              int(amax_row_stride), int(g), int(fmt), int(num_bits),
                                              int(unsigned) if int(fmt) == 4 else int(bool(unsigned)),
                                              int(bool(narrow_range)), stream))
    return delta


# ----------------------------------------------------------------------------------------------- AWQ Gram search
def sgpt_trailing_update(w: torch.Tensor, i1: int, delta: torch.Tensor, hinv: torch.Tensor) -> torch.Tensor:
    """In place: w[:, i2:] -= delta @ hinv[i1:i2, i2:] (sparsegpt.py:124, i2 = i1 + delta.shape[1]) as the fp32 fma chain
    over the block's columns in ascending order on the fp32 matrix cores -- a defined summation order where the reference
    has the BLAS library's."""
    _require_gpu(w, "sgpt_trailing_update")
    if not (w.dtype == delta.dtype == hinv.dtype == torch.float32 and w.is_contiguous() and delta.is_contiguous()
            and hinv.is_contiguous() and w.dim() == 2 and delta.dim() == 2 and delta.shape[0] == w.shape[0]):
        raise MoquantError("sgpt_trailing_update: contiguous fp32 w [rows, ld], delta [rows, bs], hinv [ld, ld] expected")
    rows, ld = w.shape
    with _on(w) as stream:
        check(_lib.lib().moq_sgpt_trailing_update(_p(w), rows, ld, int(i1), int(delta.shape[1]), _p(delta), _p(hinv), stream))
    return w


def split_bf16(x: torch.Tensor):
    """fp32 -> (hi, lo) bf16 with hi + lo == x to ~2^-17 relative."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


@torch.no_grad()
def gram_operand(gram: torch.Tensor, planes: int = 3) -> torch.Tensor:
    """The `b` operand of awq_quadform for a symmetric fp32 Gram matrix, bf16 [Cin, planes * Cin]: planes 3 =
    [G_hi | G_lo | G_hi] (against a = [E_hi | E_hi | E_lo]: E_hi G_hi + E_hi G_lo + E_lo G_hi), 2 = [G_hi | G_hi] (against
    [E_hi | E_lo]: (E_hi + E_lo) G_hi), 1 = [G_hi]."""
    if planes == 1:
        return gram.to(torch.bfloat16).contiguous()
    if planes == 2:
        hi = gram.to(torch.bfloat16)
        return torch.cat([hi, hi], dim=1).contiguous()
    hi, lo = split_bf16(gram)
    return torch.cat([hi, lo, hi], dim=1).contiguous()


@torch.no_grad()
def awq_err_weight(weight: torch.Tensor, awq_scale_dt: torch.Tensor, inv_scale_f32: torch.Tensor, group_size: int,
                   num_bits: int = 4, planes: int = 3):
    """(E fp32 [Cout, Cin], A bf16 [Cout, planes * Cin]) of one candidate: E = QDQ((W * s).to(dtype)) * r - W and its
    split-precision MFMA operand (see gram_operand), from one read of W."""
    _require_gpu(weight, "awq_err_weight")
    w = weight.detach().contiguous()
    rows, cols = w.shape
    s = awq_scale_dt.detach().to(device=w.device, dtype=w.dtype).contiguous().reshape(-1)
    r = _f32(inv_scale_f32, w.device).reshape(-1)
    err = torch.empty(rows, cols, dtype=torch.float32, device=w.device)
    a = torch.empty(rows, planes * cols, dtype=torch.bfloat16, device=w.device)
    with _on(w) as stream:
        check(_lib.lib().moq_awq_err_weight(_p(w), _p(s), _p(r), _p(err), _p(a), rows, cols, int(group_size), _dt(w),
                                            int(num_bits), int(planes), stream))
    return err, a


@torch.no_grad()
def awq_quadform(err: torch.Tensor, gram_op: torch.Tensor, loss_acc: torch.Tensor, inv_count: float,
                 a_operand: torch.Tensor | None = None) -> torch.Tensor:
    """loss_acc[0] += inv_count * trace(E G E^T) = inv_count * <E G, E> for the fp32 error weight E [Cout, Cin] and
    gram_op = gram_operand(G, planes): one MFMA contraction over K = planes * Cin (planes 3: split precision
    E_hi G_hi + E_hi G_lo + E_lo G_hi) with the product against E fused into the epilogue."""
    _require_gpu(err, "awq_quadform")
    if gram_op.dtype == torch.float32:
        # fp32 models: E and the (symmetric) Gram matrix as they are, on the fp32 matrix cores -- no split planes
        rows, cols = err.shape
        if err.dtype != torch.float32 or not err.is_contiguous() or not gram_op.is_contiguous() \
                or tuple(gram_op.shape) != (cols, cols) or loss_acc.dtype != torch.float32 or loss_acc.numel() != 1:
            raise MoquantError("awq_quadform (fp32): err fp32 [Cout, Cin], gram fp32 [Cin, Cin] (contiguous) expected")
        ws = torch.empty(int(_lib.lib().moq_awq_err_gemm_workspace(rows, cols)), dtype=torch.float32, device=err.device)
        with _on(err) as stream:
            check(_lib.lib().moq_awq_quadform(_p(err), _p(gram_op), _p(err), rows, cols, cols, _lib.F32, _p(ws),
                                              _p(loss_acc), float(inv_count), stream))
        return loss_acc
    if err.dtype != torch.float32 or not err.is_contiguous() or gram_op.dtype != torch.bfloat16:
        raise MoquantError("awq_quadform: err must be contiguous fp32, gram_op bf16 (or fp32 for fp32 models)")
    rows, cols = err.shape
    planes = gram_op.shape[1] // cols if gram_op.dim() == 2 and cols else 0
    if planes not in (1, 2, 3) or tuple(gram_op.shape) != (cols, planes * cols) or loss_acc.dtype != torch.float32 \
            or loss_acc.numel() != 1:
        raise MoquantError("awq_quadform: operand shapes do not match")
    if a_operand is None:
        hi, lo = split_bf16(err)
        a = torch.cat({1: [hi], 2: [hi, lo], 3: [hi, hi, lo]}[planes], dim=1).contiguous()
    else:
        a = a_operand
        if tuple(a.shape) != (rows, planes * cols):
            raise MoquantError("awq_quadform: a_operand and gram_op were built for different plane counts")
    ws = torch.empty(int(_lib.lib().moq_awq_err_gemm_workspace(rows, cols)), dtype=torch.float32, device=err.device)
    with _on(err) as stream:
        check(_lib.lib().moq_awq_quadform(_p(a), _p(gram_op), _p(err), rows, cols, planes * cols, _lib.BF16, _p(ws),
                                          _p(loss_acc), float(inv_count), stream))
    return loss_acc


# ----------------------------------------------------------------------------------------------- 2-D blocks
def _is_block2d_view(x: torch.Tensor, amax_shape=None, reduce_axes=None) -> bool:
    """x is the contiguous [A, br, B, bc] view of a 2-D tensor and the amax / kept axes are (0, 2)."""
    if x.dim() != 4 or not x.is_contiguous():
        return False
    if amax_shape is not None:
        return tuple(amax_shape) == (x.shape[0], 1, x.shape[2], 1) and x.shape[0] * x.shape[2] > 1
    nd = 4
    return sorted({a % nd for a in reduce_axes}) == [1, 3]


@torch.no_grad()
def block2d(x4: torch.Tensor, mode: int, amax: torch.Tensor | None = None, accumulate: bool = False,
            fp8: bool = True, num_bits: int = 8, unsigned: bool = False, narrow_range: bool = False,
            out: torch.Tensor | None = None):
    """2-D block abs-max / QDQ on the [A, br, B, bc] view of a [A * br, B * bc] tensor (moq_block2d).
    mode 0: returns amax fp32 [A, 1, B, 1]; mode 1: QDQ with `amax`; mode 2: (y, amax)."""
    _require_gpu(x4, "block2d")
    a_, br, b_, bc = x4.shape
    rows, cols = a_ * br, b_ * bc
    x2 = x4.reshape(rows, cols)
    if not x2.is_contiguous() or x2.data_ptr() != x4.data_ptr():
        raise MoquantUnsupported("block2d: the 4-D tensor must be a view of a contiguous 2-D tensor")
    am = amax
    if mode != 1:
        if am is None:
            am = torch.zeros(a_, 1, b_, 1, dtype=torch.float32, device=x4.device)
    else:
        am = _f32(amax, x4.device)
    y = None
    if mode != 0:
        y = torch.empty_like(x4) if out is None else out
    with _on(x4) as stream:
        check(_lib.lib().moq_block2d(_p(x2), _p(y), _p(am), rows, cols, int(br), int(bc), _dt(x4), int(mode),
                                     int(accumulate), int(fp8), int(num_bits), int(unsigned), int(narrow_range), stream))
    if mode == 0:
        return am
    return (y, am) if mode == 2 else y
