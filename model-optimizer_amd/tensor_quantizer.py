"""TensorQuantizer -- host-side mirror of modelopt.torch.quantization.nn.TensorQuantizer for the PTQ hot
path (nn/modules/tensor_quantizer.py:136-1413), running on our HIP kernels.

Same attribute names, same calibrate -> load_calib_amax -> quantize life cycle, same block-quant padding
rules, same error behaviour; what differs is underneath: where the reference chains several eager ops
(pre_quant_scale multiply, F.pad, reduce_amax, QDQ) this module issues ONE fused kernel when the layout
allows it (static-block INT with dynamic amax, optionally with a per-column pre_quant_scale -- the AWQ
search inner loop), and otherwise one kernel per stage.  GPU tensors only: there is no CPU path.

Scope: fake quantization for INT-k (per-tensor / per-channel / static last-axis blocks), FP8-E4M3
(per-tensor / per-channel), dynamic MX blocks, static block grids on any axes (last axis and last-two-axes tiles without a copy), and the affine
offset (`bias`) of the KV-cache presets.  Rotation and real-quant QTensors are outside this path and raise.
"""

from __future__ import annotations

import math
import warnings
from dataclasses import dataclass, field

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from . import ops
from ._lib import MoquantUnsupported
from .calib import BiasCalibrator, HistogramCalibrator, MaxCalibrator, _Calibrator, convert_quantization_axis_to_reduce_axis


@dataclass
class QuantizerAttributeConfig:
    """The fields of quantization/config.py:322-711 that matter on this path."""

    num_bits: int | tuple = 8
    axis: int | tuple | None = None
    block_sizes: dict | None = None
    unsigned: bool = False
    narrow_range: bool = False  # config default (config.py:459-463)
    calibrator: str | tuple = "max"
    fake_quant: bool = True
    enable: bool = True
    type: str = "static"  # "dynamic": amax recomputed from every input, never calibrated (config.py:478-489)
    learn_amax: bool = False
    # affine quantization (config.py:523-588): {reduced dim: None, ..., "type": "static" | "dynamic", "method": "mean" | "max_min"}
    bias: dict | None = None
    # amax without calibration (config.py:665-709): use_constant_amax = the FP8 E4M3 range (448) in the forward, no
    # `_amax` buffer (the cast-style KV-cache presets); constant_amax = `_amax` pinned to the value, forward and export
    use_constant_amax: bool = False
    constant_amax: float | None = None
    extra: dict = field(default_factory=dict)

    def __post_init__(self):
        assert self.constant_amax is None or self.constant_amax > 0, "constant_amax must be a positive value."
        assert not (self.use_constant_amax and self.constant_amax is not None), \
            "use_constant_amax and constant_amax are mutually exclusive; set only one."
        v = self.bias
        if v is None:
            return
        if "type" in v and v["type"] not in ("static", "dynamic"):
            raise ValueError(f"Invalid bias type: {v['type']}, expected 'static' or 'dynamic'")
        if "method" in v and v["method"] not in ("mean", "max_min"):
            raise ValueError(f"Invalid bias method: {v['method']}, expected 'mean' or 'max_min'")
        axis = [k for k in v if k not in ("type", "method")]
        assert len(axis) > 0, "The axis for bias computation is not specified."
        for x in axis:
            if not isinstance(x, int):
                raise ValueError(f"Invalid axis type {type(axis)}, expected int")


def _group_kernel_takes(g: int, dtype) -> bool:
    """The fused per-group abs-max + QDQ kernel walks a group in 16-byte packets, a power of two (<= 64) of them per
    group (csrc/moq_stream.hip); other sizes -- the block products of N-D grids can be anything -- take the per-row
    abs-max and the per-row QDQ instead."""
    vec = 4 if dtype == torch.float32 else 8
    p = g // vec
    return g % vec == 0 and 0 < p <= 64 and p & (p - 1) == 0


class _BufferState:
    """Descriptor for quantizer state kept in a registered buffer `_<name>` (so it travels with state_dict / .to()).

    get: the buffer, or None while it does not exist (or while `shown(q)` is false: a disabled smoothing scale reads None).
    set: None is refused; numbers become tensors; the FIRST assignment registers a detached clone as the buffer; later ones
         copy into it on the buffer's device -- `same_shape`: refusing another shape (amax, offset; a quantizer whose layout
         changed must drop the buffer first), else replacing the buffer's tensor (the smoothing scale); then `after(q)`."""

    def __init__(self, name, label=None, same_shape=True, shown=None, after=None):
        self.slot, self.label, self.same_shape, self.shown, self.after = "_" + name, label or name, same_shape, shown, after

    def __get__(self, q, owner=None):
        if q is None:
            return self
        if self.shown is not None and not self.shown(q):
            return None
        return getattr(q, self.slot, None)

    def __set__(self, q, value):
        assert value is not None, f"{self.label} cannot be set to None."
        fresh = (value if isinstance(value, torch.Tensor) else torch.tensor(value)).clone().detach()
        held = getattr(q, self.slot, None)
        if held is None:
            q.register_buffer(self.slot, fresh)
        elif not self.same_shape:
            setattr(q, self.slot, fresh.to(held.device))
        elif held.shape != fresh.shape:
            raise RuntimeError(f"Changing shape when setting {self.label} is not allowed.")
        else:
            held.data.copy_(fresh.to(held.device))
        if self.after is not None:
            self.after(q)

    @staticmethod
    def drop(q, name):
        if hasattr(q, "_" + name):
            delattr(q, "_" + name)


from torch.nn.modules.module import _global_forward_hooks as _GLOBAL_FORWARD_HOOKS  # noqa: E402
from torch.nn.modules.module import _global_forward_pre_hooks as _GLOBAL_FORWARD_PRE_HOOKS  # noqa: E402


class TensorQuantizer(nn.Module):
    def __init__(self, quant_attribute_cfg: QuantizerAttributeConfig | None = None, if_quant=True,
                 if_calib=False, amax=None):
        super().__init__()
        cfg = quant_attribute_cfg or QuantizerAttributeConfig()
        if not cfg.fake_quant:
            raise MoquantUnsupported("real quantization (fake_quant=False) is outside this path")
        # plain (non-tensor, non-module) state goes straight into the instance dict: nn.Module.__setattr__ checks every
        # assignment against its parameter / buffer / module tables, and a Llama-3-8B conversion makes 800 quantizers x 28 of
        # them (half of the convert + set_quantizers stages)
        self.__dict__.update(_if_quant=if_quant, _if_calib=if_calib, _enable_pre_quant_scale=True)
        self._take_config(cfg)
        if amax is not None:
            self.amax = amax

    def _take_config(self, cfg: QuantizerAttributeConfig):
        d = self.__dict__
        d.update(_num_bits=cfg.num_bits, _axis=cfg.axis, _block_sizes=dict(cfg.block_sizes) if cfg.block_sizes else None,
                 _unsigned=cfg.unsigned, _narrow_range=cfg.narrow_range, _fake_quant=cfg.fake_quant, _disabled=not cfg.enable,
                 _dynamic=cfg.type == "dynamic", _bias=dict(cfg.bias) if cfg.bias else None,
                 _bias_calibrator=None,  # made on first use (tensor_quantizer.py:222-223, :490-503)
                 _use_constant_amax=bool(cfg.use_constant_amax), _constant_amax=cfg.constant_amax)
        d["_calibrator"] = self._make_calibrator(cfg.calibrator)
        if cfg.constant_amax is not None:  # pinned on the buffer: forward and export read it (tensor_quantizer.py:256-261)
            self.amax = float(cfg.constant_amax)

    # ------------------------------------------------------------------ configuration
    _weight_stats_done = None  # (data_ptr, version, shape) of the weight whose max statistics this calibration already holds

    _LAYOUT_CACHES = ("_block_reshape_size", "_padding", "_slices", "_original_shape", "_amax_shape_for_export", "_block_amax_view",
                      "_nd_split", "_nd_perm", "_nd_inverse")

    def drop_layout_caches(self):
        """Shapes remembered from the last forward under the current block layout; whoever changes the layout drops them."""
        for name in self._LAYOUT_CACHES:
            self.__dict__.pop(name, None)

    def set_from_attribute_config(self, cfg: QuantizerAttributeConfig):
        """tensor_quantizer.py:228-290: (re)configure in place.  Attributes only: the buffers (`_amax`, `_pre_quant_scale`,
        `_bias_value`) stay, as in the reference -- a calibrated model keeps its calibration and smoothing through a
        temporary configuration (set_quantizer_by_cfg_context) or a full re-assignment; an `_amax` whose shape no longer
        fits the new layout is refused where it is next written (`Changing shape when setting amax is not allowed`), and a
        caller that wants a clean slate calls reset_amax()."""
        if not cfg.fake_quant:
            raise MoquantUnsupported("real quantization (fake_quant=False) is outside this path")
        self.drop_layout_caches()
        self._take_config(cfg)

    _PARTIAL_KEYS = ("enable", "num_bits", "axis", "block_sizes", "type", "calibrator", "unsigned", "narrow_range", "fake_quant",
                     "bias", "use_constant_amax", "constant_amax")

    def update_attributes(self, partial: dict):
        """set_from_attribute_config with a plain dict (tensor_quantizer.py:228-290, the form set_quantizer_attributes_partial
        passes): ONLY the named attributes are written, each through the reference's setter -- axis also reaches the calibrator,
        block sizes clear the axis, a constant amax is pinned on the buffer -- and everything else, calibration state included,
        stays as it is."""
        for key in partial:
            if key in ("rotate", "backend", "backend_extra_args", "learn_amax", "trt_high_precision_dtype", "pass_through_bwd"):
                raise MoquantUnsupported(f"quantizer attribute {key!r} is outside this path")
            assert key in self._PARTIAL_KEYS, f"{key} is not a valid `TensorQuantizer` attribute"
        d = self.__dict__
        layout = False
        for key, val in partial.items():
            if key == "enable":
                d["_disabled"] = val is False
            elif key == "type":
                d["_dynamic"] = val == "dynamic"
            elif key == "calibrator":
                d["_calibrator"] = self._make_calibrator(val)
            elif key == "axis":
                d["_axis"] = val
                if self._calibrator is not None:
                    self._calibrator._axis = val
                layout = True
            elif key == "block_sizes":
                d["_block_sizes"] = dict(val) if val else None
                if val is not None:
                    d["_axis"] = None
                    if self._calibrator is not None:
                        self._calibrator._axis = None
                layout = True
            elif key == "fake_quant":
                if not val:
                    raise MoquantUnsupported("real quantization (fake_quant=False) is outside this path")
            elif key == "bias":
                d["_bias"] = dict(val) if val else None
            elif key == "use_constant_amax":
                d["_use_constant_amax"] = bool(val)
            elif key == "constant_amax":
                d["_constant_amax"] = val
                if val is not None:
                    self.amax = float(val)
            else:  # num_bits, unsigned, narrow_range
                d["_" + key] = val
        if layout:  # shapes cached for the previous layout
            self.drop_layout_caches()

    def _make_calibrator(self, spec) -> _Calibrator:
        # config.py:599-613 / tensor_quantizer.py:235-241: "max", "histogram" or (cls, args, kwargs)
        nb = self._num_bits if isinstance(self._num_bits, int) else 8
        if spec == "max":
            return MaxCalibrator(nb, self._axis, self._unsigned)
        if spec == "histogram":
            return HistogramCalibrator(nb, self._axis, self._unsigned)
        cls, args, kwargs = (list(spec) + [(), {}])[:3] if isinstance(spec, (tuple, list)) else (spec, (), {})
        return cls(*args, **kwargs)

    num_bits = property(lambda self: self._num_bits)
    unsigned = property(lambda self: self._unsigned)
    narrow_range = property(lambda self: self._narrow_range)
    block_sizes = property(lambda self: self._block_sizes)
    fake_quant = property(lambda self: self._fake_quant)
    is_enabled = property(lambda self: not self._disabled)

    @property
    def axis(self):
        return self._axis

    @axis.setter
    def axis(self, value):
        self._axis = value
        self._calibrator._axis = value  # tensor_quantizer.py:309-316

    @property
    def _block_dynamic(self):
        """Dynamic BLOCK quantization (block_sizes["type"] == "dynamic": MX formats, NVFP4-style two-level scaling):
        block scales come from every input.  Not the same as the top-level `type: dynamic` (`_dynamic`): a
        block-dynamic quantizer with E4M3 block scales still calibrates a per-tensor amax, which becomes the
        tensor-wide amax of the two-level scale (tensor_quantizer.py:890-920, tensor_quant.py:157-195)."""
        return self._block_sizes is not None and self._block_sizes.get("type", "static") == "dynamic"

    @property
    def is_mx_format(self):
        # block scales in E8M0 (tensor_quantizer.py: is_mx_format)
        return self._block_dynamic and self._block_sizes.get("scale_bits", None) == (8, 0)

    @property
    def is_static_block_quant(self):
        return (self._block_sizes is not None and self._block_sizes.get("type", "static") == "static"
                and self._fake_quant)

    @property
    def maxbound(self):
        if self._num_bits == (4, 3):
            return 448.0
        if self._num_bits == (2, 1):
            return 6.0
        if isinstance(self._num_bits, int):
            return float((1 << (self._num_bits - 1 + int(self._unsigned))) - 1)
        raise MoquantUnsupported(f"maxbound of {self._num_bits}")

    # ------------------------------------------------------------------ state
    # The three tensors a quantizer carries -- amax, the affine offset, the smoothing scale -- live in registered buffers
    # (`_amax`, `_bias_value`, `_pre_quant_scale`: the names the reference's state_dicts use) and are reached through ONE
    # descriptor, _BufferState below: first assignment registers the buffer, later ones copy into it.  Names, error texts
    # and the None rules are the reference's interface (nn/modules/tensor_quantizer.py:341-376, :472-520).
    # amax reads None on an MX-format quantizer even when a max calibration left an `_amax` buffer on it (:358-363: E8M0 block
    # scales come from every input; the export writes no input_scale for such a quantizer)
    def _amax_shown(q):
        if q.is_mx_format or getattr(q, "_amax", None) is None:
            return False
        assert not q._dynamic, "Dynamic quantization does not have fixed amax"
        return True

    amax = _BufferState("amax", shown=_amax_shown,
                        after=lambda q: q._preserve_amax_in_fp32() if getattr(q, "_is_static_block_scale_quantizer", False) else None)
    bias_value = _BufferState("bias_value", label="bias")
    pre_quant_scale = _BufferState("pre_quant_scale", same_shape=False, shown=lambda q: q._enable_pre_quant_scale)

    def _preserve_amax_in_fp32(self):
        """StaticBlockScaleQuantizer._preserve_amax_in_fp32 (tensor_quantizer.py:1501-1518)."""
        amax = getattr(self, "_amax", None)
        if amax is not None and amax.dtype != torch.float32:
            self._amax = amax.to(dtype=torch.float32)

    def promote_static_block(self):
        """StaticBlockScaleQuantizer.from_tensor_quantizer (tensor_quantizer.py:1521-1546) for INT static-block
        weight quantizers: from now on the per-block amax is kept in fp32."""
        self._is_static_block_scale_quantizer = True
        self._preserve_amax_in_fp32()

    def reset_amax(self):
        self._weight_stats_done = None
        _BufferState.drop(self, "amax")
        self._calibrator.reset()
        self.reset_bias()

    # ------------------------------------------------------------------ affine offset (tensor_quantizer.py:389-503, :722-734, :775-786)
    bias = property(lambda self: self._bias)
    bias_method = property(lambda self: (self._bias or {}).get("method", "mean") if self._bias is not None else None)
    bias_axis = property(lambda self: getattr(self, "_bias_axis", None))

    @bias_axis.setter
    def bias_axis(self, value):
        assert value is not None, "bias_axis cannot be set to None."
        assert isinstance(value, (tuple, list)), "bias_axis must be a tuple or a list."
        self._bias_axis = value

    @property
    def bias_type(self):
        return (self._bias or {}).get("type", "static") if self._bias is not None else None

    @bias_type.setter
    def bias_type(self, value):
        assert value in ("static", "dynamic"), "bias_type must be either 'static' or 'dynamic'."
        self._bias["type"] = value

    @property
    def bias_calibrator(self):
        """Made on first use from the config's offset entry: its integer keys are the axes the offset keeps."""
        if self._bias is not None and self._bias_calibrator is None:
            self.bias_axis = tuple(k for k in self._bias if isinstance(k, int))
            self._bias_calibrator = BiasCalibrator(method=self.bias_method, axis=self.bias_axis)
        return self._bias_calibrator

    def reset_bias(self):
        _BufferState.drop(self, "bias_value")
        if self._bias_calibrator is not None:
            self._bias_calibrator.reset()

    def load_calib_bias(self, *args, **kwargs):
        assert not self._dynamic, "Dynamic quantization does not need calibration."
        found = self.bias_calibrator.compute_bias(*args, **kwargs)
        if found is None:
            raise RuntimeError("Calibrator returned None. This usually happens when calibrator hasn't seen any tensor.")
        _BufferState.drop(self, "bias_value") if self.bias_value is not None and self.bias_value.shape != found.shape else None
        self.bias_value = found

    def _get_bias(self, inputs):
        """The offset of this call: the calibrated buffer, or (type dynamic) the statistic of the input itself."""
        cal, kind = self.bias_calibrator, self.bias_type
        if cal is None:
            return None
        if kind not in ("static", "dynamic"):
            raise ValueError(f"Unsupported bias type: {kind}")
        return self._bias_value if kind == "static" else cal.compute_dynamic_bias(inputs)

    @property
    def step_size(self):
        """amax / maxbound of an integer format (tensor_quantizer.py:396-405)."""
        if not hasattr(self, "_amax"):
            warnings.warn("step_size is undefined under dynamic amax mode!")
            return None
        assert isinstance(self._num_bits, int), "Step size is not defined for non-integer quantization."
        return self._amax / (2.0 ** (self._num_bits - 1 + int(self._unsigned)) - 1.0)

    @property
    def is_fp8(self):
        """Per-tensor FP8 E4M3: no block scales, no per-channel axis (:553-556)."""
        return self._num_bits == (4, 3) and self._block_sizes is None and self._axis is None

    def is_mxfp(self, bits):
        """MXFP4 / MXFP6 / MXFP8: E8M0 scales over blocks of 32 (:583-604)."""
        elem = {4: (2, 1), 6: (3, 2), 8: (4, 3)}.get(bits)
        if elem is None:
            raise NotImplementedError()
        nb = tuple(self._num_bits) if isinstance(self._num_bits, (list, tuple)) else self._num_bits
        return bool(self.is_mx_format and nb == elem and self._block_sizes.get(-1, None) == 32)

    def disable_pre_quant_scale(self):
        """Context manager: the pre-quant scale is not applied inside (:1387-1395)."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            was = self._enable_pre_quant_scale
            self._enable_pre_quant_scale = False
            try:
                yield
            finally:
                self._enable_pre_quant_scale = was

        return ctx()

    def validate_attr(self, attr_value=None, attr_name="amax", raise_error=False, warn_error=False, name=""):
        """True when the attribute is absent or holds only finite values >= 0 (:753-773)."""
        attr_value = attr_value if attr_value is not None else getattr(self, attr_name, None)
        if attr_value is None or (isinstance(attr_value, torch.Tensor) and attr_value.is_meta):
            return True
        if bool(torch.all(attr_value >= 0)) and not bool(torch.any(torch.isinf(attr_value))) \
                and not bool(torch.any(torch.isnan(attr_value))):
            return True
        msg = f"{f'{name}.' if name else ''}{attr_name} contains invalid values: {attr_value}"
        if warn_error:
            warnings.warn(msg)
        if raise_error:
            raise ValueError(msg)
        return False

    def disable(self):
        self._disabled = True

    def enable(self):
        self._disabled = False

    def mark_weight_stats_done(self, weight):
        """model_calib.weight_only_quantize collected this quantizer's max statistics from `weight`: calls with the same,
        unchanged tensor are pass-throughs until the calibration ends (disable_calib / reset_amax).  Only running-max
        calibrators qualify (a histogram would count the weight once per forward in the reference)."""
        if (type(self._calibrator).__name__ == "MaxCalibrator" and self.pre_quant_scale is None and self._bias is None
                and not weight.is_inference()):  # (an inference tensor has no version counter to recognise it unchanged by)
            self._weight_stats_done = (weight.data_ptr(), weight._version, tuple(weight.shape))

    def enable_calib(self):
        if self._dynamic:  # dynamic quantization does not need calibration (tensor_quantizer.py:676-684)
            return
        self._if_calib = True

    def disable_calib(self):
        self._weight_stats_done = None
        self._if_calib = False

    def enable_quant(self):
        self._if_quant = True

    def disable_quant(self):
        self._if_quant = False

    def load_calib_amax(self, *args, **kwargs):
        """tensor_quantizer.py:693-720."""
        assert not self._dynamic, "Dynamic quantization does not need calibration."
        strict = kwargs.pop("strict", True)
        calib_amax = self._calibrator.compute_amax(*args, **kwargs)
        if calib_amax is None:
            msg = "Calibrator returned None. This usually happens when calibrator hasn't seen any tensor."
            if strict:
                raise RuntimeError(msg + " Passing 'strict=False' to `load_calib_amax()` will ignore the error.")
            warnings.warn(msg)
            warnings.warn("Set amax to NaN!")
            calib_amax = torch.tensor(math.nan)
        self.replace_amax(calib_amax)

    def in_amax_buffer_shape(self, amax: torch.Tensor) -> torch.Tensor:
        """A kernel-side (folded) amax in the shape the `_amax` BUFFER keeps -- the reference's: one entry per leading
        index and tile / one per block with the block parts as 1 (`_block_amax_view`, set by _setup_for_blockquant for tiles
        on tensors of rank > 2 and for blocks on other than the last axis); every other layout is its own buffer shape."""
        view = getattr(self, "_block_amax_view", None)
        if view is not None and amax.numel() == math.prod(view):
            return amax.reshape(view)
        return amax

    def replace_amax(self, amax: torch.Tensor):
        """Every place that LOADS a calibrated amax (load_calib_amax, model_calib.finish_stats_collection, the MoE expert
        sync, layer-by-layer restores) goes through here: buffer shape of the reference, a buffer of another shape dropped
        first (the setter refuses a shape change)."""
        amax = self.in_amax_buffer_shape(amax)
        if hasattr(self, "_amax") and self._amax.shape != amax.shape:
            delattr(self, "_amax")
        self.amax = amax

    def export_amax(self):
        """tensor_quantizer.py:1087-1117: 0 / NaN entries are replaced by maxbound, values clamped to the dtype's
        finite positive range; static last-axis block amax is reshaped to (*shape[:-1], -1); without blocks a per-tensor
        amax gets a leading dim (runtimes expect dim >= 1) and a single-axis amax is squeezed to 1-D."""
        if self._block_dynamic:
            return self.amax
        if self.amax is None:
            return None
        amax = self.amax.detach().clone()
        if hasattr(self, "_amax_shape_for_export"):
            amax = amax.reshape(self._amax_shape_for_export)
        amax[(amax == 0) | torch.isnan(amax)] = self.maxbound
        fi = torch.finfo(amax.dtype)
        amax = amax.clamp(min=fi.tiny, max=fi.max)
        if self._block_sizes is None:
            if self._axis is None:
                if amax.dim() == 0:
                    amax = amax.unsqueeze(0)
            elif isinstance(self._axis, int) or (isinstance(self._axis, (list, tuple)) and len(self._axis) == 1):
                amax = amax.squeeze()
        return amax

    def sync_amax_across_distributed_group(self, group=None):
        """tensor_quantizer.py:1373-1385 (one all-reduce; use distributed.sync_amax_bucketed for many)."""
        if dist.is_available() and dist.is_initialized() and getattr(self, "_amax", None) is not None:
            dist.all_reduce(self._amax, op=dist.ReduceOp.MAX, group=group)

    # ------------------------------------------------------------------ block layout (last axis only)
    def _block_size_last(self, inputs):
        bs = self._block_sizes
        g = bs.get(-1, None) or bs.get(inputs.dim() - 1, None)
        other = [k for k in bs if isinstance(k, int) and k not in (-1, inputs.dim() - 1)]
        if g is None or other:
            raise MoquantUnsupported("only last-axis block quantization is implemented on this path "
                                     "(reference general N-D path: tensor_quantizer.py:1018-1043)")
        return g

    def _setup_for_blockquant(self, inputs):
        """tensor_quantizer.py:975-1043.  Last-axis blocks: right-pad the last dim with zeros to a block multiple,
        view as (-1, g), quantization axis (0,).  Blocks on the LAST TWO axes (the FP8 2-D blockwise preset; any
        rank): zero-pad to whole tiles, view as (L * R/br, br, C/bc, bc), quantization axes (0, 2) -- served by the 2-D
        block kernel, the amax buffer in the reference's (L..., R/br, 1, C/bc, 1) shape.  Any other set of blocked axes:
        one permuted copy brings it to the last-axis layout."""
        if hasattr(self, "_block_reshape_size"):
            return
        bs = self._block_sizes
        axes = {(k if k >= 0 else inputs.dim() + k): v for k, v in bs.items() if isinstance(k, int)}
        nd = inputs.dim()
        if len(axes) == 2 and nd >= 2 and set(axes) == {nd - 2, nd - 1}:
            # tiles over the LAST TWO axes; leading dims (stacked experts, conv output channels ...) fold into the tile
            # rows once the matrix dims are padded to whole tiles, so the 2-D kernel serves any rank
            br, bc = axes[nd - 2], axes[nd - 1]
            lead = tuple(inputs.shape[:-2])
            rows, cols = inputs.shape[-2:]
            pad_r, pad_c = (-rows) % br, (-cols) % bc
            self._original_shape = inputs.shape
            if pad_r or pad_c:  # right / bottom zero padding to whole tiles, cut off again after the QDQ (:1018-1043)
                self._padding = (0, pad_c, 0, pad_r)
                self._slices = (*(slice(None),) * len(lead), slice(rows), slice(cols))
                self._original_shape = torch.Size((*lead, rows + pad_r, cols + pad_c))
            a, b = (rows + pad_r) // br, (cols + pad_c) // bc
            self._block_reshape_size = torch.Size((math.prod(lead) * a, br, b, bc))
            if lead:  # the amax buffer keeps the reference's shape (one entry per leading index and tile)
                self._block_amax_view = torch.Size((*lead, a, 1, b, 1))
            self.axis = (0, 2)
            return
        if any(d != nd - 1 for d in axes):
            # any other set of blocked axes (rows only, a conv weight's input channels, ...; :1018-1043): every blocked
            # axis is padded to whole blocks and split into (blocks, block); the block parts are gathered behind the
            # rest by ONE permuted copy, after which the layout is that of last-axis blocks of size prod(blocks) --
            # the per-row kernels serve it, and the amax buffer keeps the reference's shape (block parts = 1)
            sizes = [axes.get(d) for d in range(nd)]
            pads = [(-inputs.shape[d]) % b if b else 0 for d, b in enumerate(sizes)]
            self._original_shape = inputs.shape
            if any(pads):
                first = next(d for d, p in enumerate(pads) if p)
                self._padding = tuple(v for d in range(nd - 1, first - 1, -1) for v in (0, pads[d]))
                self._slices = tuple(slice(inputs.shape[d]) if pads[d] else slice(None) for d in range(nd))
                self._original_shape = torch.Size(inputs.shape[d] + pads[d] for d in range(nd))
            split, grid_at, tile_at = [], [], []
            for d, b in enumerate(sizes):
                n = self._original_shape[d]
                if b:
                    grid_at.append(len(split)), tile_at.append(len(split) + 1)
                    split += [n // b, b]
                else:
                    grid_at.append(len(split))
                    split.append(n)
            perm = grid_at + tile_at
            self._nd_split, self._nd_perm = torch.Size(split), tuple(perm)
            self._nd_inverse = tuple(perm.index(i) for i in range(len(perm)))
            tile = math.prod(split[i] for i in tile_at)
            self._block_reshape_size = torch.Size((-1, tile))
            self._block_amax_view = torch.Size(1 if i in tile_at else n for i, n in enumerate(split))
            self.axis = (0,)
            return
        g = self._block_size_last(inputs)
        self._original_shape = inputs.shape
        pad = (-inputs.shape[-1]) % g
        if pad:
            self._padding = (0, pad)
            self._slices = (*(slice(None),) * (inputs.dim() - 1), slice(inputs.shape[-1]))
            self._original_shape = torch.Size((*inputs.shape[:-1], inputs.shape[-1] + pad))
        self._block_reshape_size = torch.Size((-1, g))
        self._amax_shape_for_export = (*inputs.shape[:-1], -1)
        self.axis = (0,)

    def _process_for_blockquant(self, inputs):
        if hasattr(self, "_padding"):
            inputs = F.pad(inputs, self._padding, "constant", 0)
        if inputs.shape != self._original_shape:
            raise ValueError(f"Input shape has changed from {self._original_shape} to {inputs.shape}."
                             " Block-quantization requires a fixed input shape.")
        if hasattr(self, "_nd_perm"):
            return inputs.reshape(self._nd_split).permute(self._nd_perm).reshape(self._block_reshape_size)
        return inputs.reshape(self._block_reshape_size)

    def _reset_to_original_shape(self, outputs):
        if hasattr(self, "_nd_perm"):
            gathered = [self._nd_split[i] for i in self._nd_perm]
            outputs = outputs.reshape(gathered).permute(self._nd_inverse)
        outputs = outputs.reshape(self._original_shape)
        if hasattr(self, "_slices"):
            outputs = outputs[self._slices]
        return outputs

    # ------------------------------------------------------------------ forward
    def _block_sizes_to_axis(self, x):
        """tensor_quantizer.py:1056-1085: block_sizes whose sizes are all None name the REDUCED dims (per-token
        activations: {-1: None}); they become the complementary `axis` on the first input and block_sizes is dropped."""
        bs = self._block_sizes
        if bs is None or not all(v is None for k, v in bs.items() if isinstance(k, int)):
            return
        assert self._axis is None, "Axis and block_sizes are both set."
        reduced = tuple(k if k >= 0 else k + x.dim() for k in bs if isinstance(k, int))
        self._axis = tuple(i for i in range(x.dim()) if i not in reduced) or None
        if getattr(self, "_calibrator", None) is not None:
            self._calibrator._axis = self._axis
        self._block_sizes = None

    def _get_amax(self, inputs):
        if self._use_constant_amax:  # (tensor_quantizer.py:738-739)
            return torch.tensor(torch.finfo(torch.float8_e4m3fn).max, device=inputs.device)
        if hasattr(self, "_amax"):
            amax = self._amax.to(inputs.device) if self._amax.device != inputs.device else self._amax
            view = getattr(self, "_block_amax_view", None)
            if view is not None and amax.shape == view:  # the kernels' folded view of an N-D grid's amax
                amax = amax.reshape(-1, 1) if hasattr(self, "_nd_perm") else \
                    amax.reshape(self._block_reshape_size[0], 1, self._block_reshape_size[2], 1)
            return amax
        reduce_axis = convert_quantization_axis_to_reduce_axis(inputs, self._axis)
        return ops.reduce_amax(inputs, axis=reduce_axis, keepdims=True)

    def collect(self, inputs):
        if not self._if_calib or self._dynamic:
            return
        if self._bias is not None and self.bias_type == "static":
            # the offset first, the abs-max of what is left after it (tensor_quantizer.py:1402-1407)
            self.bias_calibrator.collect(inputs)
            inputs = inputs - self.bias_calibrator.compute_bias()
        self._calibrator.collect(inputs)

    def _fake_quantize(self, inputs):
        bias = None if self._bias is None or self._block_dynamic else self._get_bias(inputs)
        if bias is None:  # (dynamic block formats take no offset: tensor_quant.py:539-561)
            return self._fake_quantize_at(inputs, None)
        # the amax is the calibrated one, or that of the input as it came (tensor_quantizer.py:899-901); the grid is laid
        # around the offset: QDQ(x - bias) + bias (tensor_quant.py:364-392, :438-453)
        amax = self._get_amax(inputs)
        return self._fake_quantize_at(inputs - bias, amax) + bias

    def _fake_quantize_at(self, inputs, amax):
        if self._block_dynamic:
            g = self._block_sizes.get(-1, None) or self._block_sizes.get(inputs.dim() - 1, None)
            if g is None:
                raise ValueError("block size for dynamic quantization not found.")
            amax = None if self.is_mx_format else self._get_amax(inputs)
            nb = self._num_bits if not isinstance(self._num_bits, list) else tuple(self._num_bits)
            sb = self._block_sizes.get("scale_bits", None)
            # dynamic blocks need their scale format spelled out; the reference's forward asserts exactly this
            # (tensor_quant.py:482) -- found by tools/quantizer_fuzz.py: a missing scale_bits used to mean E8M0 here
            assert isinstance(sb, (tuple, list)) and len(sb) == 2
            return ops.dynamic_block_quant(inputs, g, amax, nb, tuple(sb))
        if isinstance(self._num_bits, tuple):
            if tuple(self._num_bits) != (4, 3):
                raise MoquantUnsupported(f"float format {self._num_bits} without dynamic blocks")
            return ops.scaled_e4m3(inputs, self._get_amax(inputs) if amax is None else amax)
        if (amax is None and self.is_static_block_quant and not hasattr(self, "_amax") and inputs.dim() == 2
                and _group_kernel_takes(inputs.shape[-1], inputs.dtype)):
            # dynamic per-block amax + QDQ in one pass (what _get_amax + fake_tensor_quant do in two)
            y, _ = ops.amax_qdq_int_group(inputs, inputs.shape[-1], self._num_bits, self._unsigned,
                                          self._narrow_range, return_amax=False)
            return y
        return ops.fake_tensor_quant(inputs, self._get_amax(inputs) if amax is None else amax, self._num_bits,
                                     self._unsigned, self._narrow_range)

    def _fused_input_pass(self, inputs, pqs):
        """pre_quant_scale * x -> collect -> fake-quantize of a PER-TENSOR quantizer as ONE kernel pass over the
        activation (ops.input_quant; the reference runs a multiply, an amax + amin pair and ~8 elementwise kernels,
        tensor_quantizer.py:1143-1212).  Returns the output, or None when this call is not of that shape."""
        if (self._disabled or self._bias is not None or inputs.dtype not in (torch.float32, torch.float16, torch.bfloat16)
                or self._axis is not None or self._block_sizes is not None or self._dynamic
                or not (self._if_quant or self._if_calib) or pqs.numel() != inputs.shape[-1] or pqs.numel() < 2
                or inputs.shape[-1] % (4 if inputs.dtype == torch.float32 else 8)):
            return None
        nb = self._num_bits
        if not (isinstance(nb, int) or tuple(nb) == (4, 3)):
            return None
        amax_q = None
        if self._if_quant:
            amax_q = getattr(self, "_amax", None)
            if amax_q is None or amax_q.numel() != 1:
                return None  # dynamic amax of this very input: the reduction has to finish before the QDQ can start
        running = None
        if self._if_calib:
            cal = self._calibrator
            if type(cal) is not MaxCalibrator or cal._track_amax:
                return None
            if cal._buf is None:
                cal._buf = torch.zeros(1, dtype=torch.float32, device=inputs.device)
                cal._shape, cal._dtype = (), inputs.dtype
            elif cal._shape != ():
                raise RuntimeError("amax shape changed!")
            running = cal._buf
        return ops.input_quant(inputs, pqs, amax_running=running, qdq_amax=amax_q, num_bits=nb if amax_q is not None else None,
                               unsigned=self._unsigned, narrow_range=self._narrow_range)

    # -- shortcuts for the callers that own a quantizer (QuantLinear, the attention wrapper): a calibration loop of
    # Llama-3-8B makes 30 000 calls of quantizers that hand their input straight back (disabled output / query quantizers;
    # weight quantizers whose statistics weight_only_quantize already took), each through nn.Module.__call__ and the chain of
    # tests in forward() -- ~0.1 s of host time in a loop the host barely keeps ahead of the GPU (profiles/r05g_fp8_flow_overhead.md)
    def hands_back(self) -> bool:
        """True when calling this quantizer cannot do anything to its input: disabled, no smoothing scale, and nobody
        hooked its call.  (The reference's forward returns the input of a disabled quantizer after the pre_quant_scale
        step, tensor_quantizer.py:1119-1150.)"""
        return (self._disabled and "_pre_quant_scale" not in self._buffers and not self._forward_hooks
                and not self._forward_pre_hooks and not _GLOBAL_FORWARD_HOOKS and not _GLOBAL_FORWARD_PRE_HOOKS)

    def weight_already_counted(self, w) -> bool:
        """True inside max_calibrate's forward loop for the weight whose statistics this calibration already holds (the
        second test of forward(), without the call)."""
        done = self._weight_stats_done
        return (done is not None and self._if_calib and not self._if_quant and not self._disabled and done[0] == w.data_ptr()
                and not w.is_inference() and done[1] == w._version and done[2] == tuple(w.shape) and not self._forward_hooks
                and not self._forward_pre_hooks and not _GLOBAL_FORWARD_HOOKS and not _GLOBAL_FORWARD_PRE_HOOKS)

    def forward(self, inputs):
        if inputs.numel() == 0:
            return inputs
        if (self._if_calib and not self._if_quant and not self._disabled and not self._dynamic and self._block_sizes is None
                and self._axis is None and self._bias is None and self._weight_stats_done is None and "_pre_quant_scale" not in self._buffers
                and type(self._calibrator) is MaxCalibrator and self._calibrator.collect_per_tensor_fast(inputs)):
            # a per-tensor max-calibrated activation quantizer inside the calibration loop: statistics only, the input
            # passes through (the general path below does exactly this, through a dozen more host-side steps)
            return inputs
        if self._weight_stats_done is not None and self._if_calib and not self._if_quant and not self._disabled:
            # a weight quantizer inside max_calibrate's forward loop: its statistics were taken by weight_only_quantize
            # (one multi-tensor launch) from this very tensor; the reference collects them again on every forward
            # (model_calib.py:351-362), which for a running abs-max of an unchanged weight changes nothing
            done = self._weight_stats_done
            if (done[0] == inputs.data_ptr() and not inputs.is_inference() and done[1] == inputs._version
                    and done[2] == tuple(inputs.shape)):
                return inputs
        pqs = self.pre_quant_scale
        fused_pqs = False
        if pqs is not None:
            out = self._fused_input_pass(inputs, pqs)
            if out is not None:
                return out
            can_fuse = (not self._disabled and self._bias is None and self._if_quant and not self._if_calib
                        and self.is_static_block_quant and not hasattr(self, "_amax") and isinstance(self._num_bits, int) and not self._unsigned
                        and not self._narrow_range and inputs.dim() == 2 and pqs.numel() == inputs.shape[-1])
            if can_fuse:
                g = self._block_size_last(inputs)
                can_fuse = inputs.shape[-1] % g == 0
            if can_fuse:
                fused_pqs = True
            elif pqs.numel() == inputs.shape[-1] and pqs.numel() > 1:
                inputs = ops.scale_cols(inputs, pqs)  # x * s rounded once to x.dtype == inputs * pre_quant_scale
            else:
                inputs = inputs * pqs
        if self._disabled:
            return inputs
        if self._block_sizes is not None and self._fake_quant:
            self._block_sizes_to_axis(inputs)
        if fused_pqs:
            # AWQ search inner op: QDQ_g((W * s).to(dtype)) with dynamic group amax, one read + one write
            return ops.awq_scale_qdq(inputs, pqs, self._block_size_last(inputs), self._num_bits)
        if self.is_static_block_quant:
            self._setup_for_blockquant(inputs)
            inputs = self._process_for_blockquant(inputs)
        outputs = inputs
        if self._if_calib and not self._dynamic:
            self.collect(inputs)
        if self._if_quant:
            # no `.contiguous()` here: every op takes what it needs (per-tensor formats walk a permuted dense
            # tensor in place, the others copy)
            outputs = self._fake_quantize(inputs)
        if self.is_static_block_quant:
            outputs = self._reset_to_original_shape(outputs)
        return outputs

    @staticmethod
    def _short_tensor(tensor: torch.Tensor, fmt=".2e") -> str:
        if tensor.numel() == 1:
            return f"{tensor.item():{fmt}}"
        return f"[{tensor.min().item():{fmt}}, {tensor.max().item():{fmt}}]({tensor.numel()})"

    def _get_name(self):
        # a quantizer promoted to static block scales prints under the reference's subclass name (tensor_quantizer.py:1483-1500)
        return "StaticBlockScaleQuantizer" if getattr(self, "_is_static_block_scale_quantizer", False) else super()._get_name()

    def _short_amax(self, fmt=None) -> str:
        fmt = fmt or (".4f" if getattr(self, "_is_static_block_scale_quantizer", False) else ".2e")  # (:1623)
        if self.is_mx_format:
            return "None"
        if self._use_constant_amax:
            return f"{torch.finfo(torch.float8_e4m3fn).max:{fmt}}(const)"
        held = getattr(self, "_amax", None)
        if not hasattr(self, "_amax"):
            return "dynamic"
        return "None" if held is None else "meta" if held.is_meta else self._short_tensor(held, fmt)

    def extra_repr(self):
        """The line print_quant_summary shows per quantizer, in the reference's words and order (tensor_quantizer.py:1223-1297):
        `disabled`, or sign / bits / narrow / fake, the block sizes or the axis (`per-tensor`), `amax=` as a value, a
        `[min, max](count)` range, `dynamic`, `None` (MX) or the constant, the smoothing scale, the calibrator, the offset, and
        the quant / calib switches."""
        pqs = self.pre_quant_scale
        smoothing = f" pre_quant_scale={self._short_tensor(pqs)}" if pqs is not None else ""
        if self._disabled:
            return "disabled" + smoothing
        s = f"{'unsigned ' if self._unsigned else ''}{self._num_bits} bit"
        s += " narrow" if self._narrow_range else ""
        s += " fake" if self._fake_quant else ""
        if self._block_sizes is not None:
            s += f" block_sizes={self._block_sizes},"
        else:
            s += f" axis={self._axis}" if self._axis is not None else " per-tensor"
        s += f" amax={self._short_amax()}" + smoothing
        s += f" calibrator={type(self._calibrator).__name__}" if self._calibrator is not None else ""
        if self._bias:
            s += f" bias={self._bias}"
        s += " quant" if self._if_quant else ""
        s += " calib" if self._if_calib else ""
        return s


class SequentialQuantizer(nn.Sequential):
    """A chain of TensorQuantizers applied one after the other to the same tensor (W4A8: INT4 blocks, then FP8) --
    nn/modules/tensor_quantizer.py:1797-1862.  Methods are broadcast to every member, properties come from the
    first one, like the reference's _QuantizerContainerBase delegation."""

    _BROADCAST = ("disable", "enable", "enable_calib", "disable_calib", "enable_quant", "disable_quant",
                  "reset_amax", "load_calib_amax")

    def __init__(self, *quantizers: TensorQuantizer):
        super().__init__(*quantizers)
        assert all(isinstance(q, TensorQuantizer) for q in self), "All quantizers must be a TensorQuantizer."

    def __getattr__(self, name):
        if name in SequentialQuantizer._BROADCAST:
            def call(*args, **kwargs):
                out = None
                for q in self:
                    out = getattr(q, name)(*args, **kwargs)
                return out  # the last member's result (_format_delegated_method_outputs, :1825)
            return call
        try:
            return super().__getattr__(name)
        except AttributeError:
            # public properties come from the first member (the reference delegates `fake_quant`, `is_enabled`, `amax`,
            # :1740-1769; this package's own callers also read `num_bits`, `block_sizes`, ...).  PRIVATE state is not
            # delegated: `hasattr(chain, "_amax")` is False as in the reference (found by tools/flow_fuzz.py)
            if len(self) and not name.startswith("_"):
                return getattr(self[0], name)  # first member wins
            raise

    @staticmethod
    def convert_to_single_quantizer(model, indx: int = 0):
        """Context manager: every SequentialQuantizer in `model` is replaced by its member `indx` (used to calibrate
        the members individually, e.g. the AWQ search on the INT4 stage) -- :1836-1862."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            saved = []
            for name, module in list(model.named_modules()):
                if isinstance(module, SequentialQuantizer):
                    assert len(module) > indx
                    parent = model.get_submodule(name.rpartition(".")[0])
                    attr = name.rpartition(".")[-1]
                    saved.append((parent, attr, module))
                    setattr(parent, attr, module[indx])
            try:
                yield
            finally:
                for parent, attr, module in saved:
                    setattr(parent, attr, module)

        return ctx()


class GroupedQuantizer(nn.ModuleList):
    """Per-group quantizers for one module that holds several independently quantized weights
    (nn/modules/tensor_quantizer.py:1865-1893: the fused experts of a grouped linear).  Unlike SequentialQuantizer the
    members act on DIFFERENT tensors: index in with `grouped[i](weight_i)`; `forward` applies the first member (the
    single-weight compatibility path), property reads come from the first member, life-cycle methods are broadcast
    and return the list of the members' results."""

    def __init__(self, *quantizers):
        super().__init__(quantizers)
        assert all(isinstance(q, (TensorQuantizer, SequentialQuantizer)) for q in self), \
            "All quantizers must be a TensorQuantizer or SequentialQuantizer."

    def forward(self, inputs):
        return self[0](inputs)

    def __getattr__(self, name):
        if name in SequentialQuantizer._BROADCAST:
            return lambda *args, **kwargs: [getattr(q, name)(*args, **kwargs) for q in self]
        try:
            return super().__getattr__(name)
        except AttributeError:
            if len(self) and not name.startswith("_"):  # (public properties only, as for SequentialQuantizer)
                return getattr(self[0], name)
            raise
