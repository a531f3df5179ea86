"""S7 -- the ALGORITHM seam into modelopt.torch.quantization (INTEGRATION.md "S7").

The kernel seams of modelopt_plugin (S1 / S2 / S3 / S5 / S6) put our kernels underneath the reference's own Python loops: one
launch per `fake_tensor_quant` / `reduce_amax` call.  What makes a PTQ run fast on this chip lives one level up -- the
multi-tensor weight pass, the per-decoder-layer deferred statistics, the Gram screen + batched MFMA error GEMM of the AWQ
search, the fused fold / export packers -- in this package's model_calib / model_quant / export.  The reference HAS a seam
at that level: every calibration algorithm is a class attribute `_calib_func` of its mode descriptor, read at call time
(modelopt/torch/quantization/mode.py:343-360; max :431, mse :460, local_hessian :476, smoothquant :488, awq_lite / awq_clip /
awq_full :500-524, gptq :555), and `model_calib.weight_only_quantize` (:187), `model_quant.fold_weight` (:728),
`export/quant_utils.pack_int4_in_uint8` / `to_quantized_weight` (:792, :836) are module-level functions.

`install_algorithms()` re-points those at adapters.  An adapter never builds a second model: for the duration of ONE algorithm
call it ADOPTS the reference's quantized model --

  * every reference `TensorQuantizer` child (and every stage of a `SequentialQuantizer`) is swapped in its parent's `_modules`
    for a twin of this package's `TensorQuantizer` holding the SAME attribute values and the SAME buffer tensors (`_amax`,
    `_pre_quant_scale`, `_bias_value`) and the running maximum of its calibrator;
  * every reference `QuantLinear` made from a plain `nn.Linear` takes this package's `QuantLinear` class for the call (the
    reference promotes quantizers by the same in-place class assignment: nn/modules/tensor_quantizer.py:1533-1546);
    parameters stay where they are, so weights are updated in place;
  * every other quantized module of the reference (attention with its *_bmm_quantizers, LayerNorm, MoE blocks) keeps its class:
    its forward calls `self.<name>_quantizer(x)`, which now reaches the twin;

runs this package's algorithm on it, and RELEASES it: attribute values, buffers, calibrator maxima and the
StaticBlockScale promotion are written back to the reference's own quantizer objects, which return to their places.  What the
caller holds afterwards is the reference's model in the state the reference's algorithm would have left (same buffer names,
same enable flags, same `_amax_for_smoothing`), and everything downstream (its export, its save / restore) proceeds
unchanged.  A model the adoption cannot represent faithfully (a quantized module type outside this path with an enabled
weight quantizer, rotation, custom backends, real-quantized weights, tensor / expert parallel groups, offloaded weights, CPU
tensors) is handed back to the reference's own function, counted in modelopt_plugin.STATS as `S7:<algorithm>:fallback:<why>`.

Nothing here imports modelopt at module import time.
"""

from __future__ import annotations

import contextlib

import torch
from torch import nn

from . import model_calib as _mc
from . import model_quant as _mq
from . import numerics
from .nn import QuantLinear
from .tensor_quantizer import QuantizerAttributeConfig, SequentialQuantizer, TensorQuantizer

_BUFFERS = ("_amax", "_pre_quant_scale", "_bias_value")
# plain attributes an ALGORITHM changes on a quantizer (enable / calibration switches, the smoothing switch, dynamic-ness); written
# back on release together with `_axis` and `_block_sizes` (same names on both sides).  The format itself (`_num_bits`,
# `_unsigned`, `_narrow_range`) is configuration, not calibration: no algorithm touches it, and it is left as the reference holds it.
_FLAGS = ("_disabled", "_if_quant", "_if_calib", "_dynamic", "_enable_pre_quant_scale")
# attributes algorithms hang on a quantizer beside its state (model_calib.py:1646)
_EXTRAS = ("_amax_for_smoothing",)


class CannotAdopt(Exception):
    """The model holds something this path does not represent; the caller falls back to the reference's function."""


def _ref():
    """The reference's classes, imported on first use."""
    import modelopt.torch.quantization.calib as rcalib
    from modelopt.torch.quantization.nn import QuantModule, SequentialQuantizer as RSequential, TensorQuantizer as RTensorQuantizer
    from modelopt.torch.quantization.nn.modules import tensor_quantizer as rtq

    return {"TQ": RTensorQuantizer, "Seq": RSequential, "QuantModule": QuantModule, "calib": rcalib,
            "Static": getattr(rtq, "StaticBlockScaleQuantizer", None)}


def _calibrator_spec(r, R):
    cal = getattr(r, "_calibrator", None)
    if cal is None:
        return None
    if type(cal) is R["calib"].MaxCalibrator:
        if getattr(cal, "_track_amax", False):
            raise CannotAdopt("MaxCalibrator(track_amax=True)")
        return "max"
    if type(cal) is R["calib"].HistogramCalibrator:
        if getattr(cal, "_calib_hist", None) is not None:
            raise CannotAdopt("a histogram calibrator that already holds counts")
        return "histogram"
    raise CannotAdopt(f"calibrator {type(cal).__name__}")


def _twin(r, R) -> TensorQuantizer:
    """This package's TensorQuantizer in the state of the reference's quantizer `r`."""
    d = r.__dict__
    if not d.get("_fake_quant", True):
        raise CannotAdopt("real quantization (fake_quant=False)")
    if d.get("_rotate"):
        raise CannotAdopt("rotation")
    if d.get("backend") is not None:
        raise CannotAdopt(f"quant backend {d.get('backend')!r}")
    if d.get("_learn_amax") or d.get("_dequantize"):
        raise CannotAdopt("learn_amax / dequantize mode")
    if d.get("_shared_quant_tied_attrs"):
        raise CannotAdopt("tied shared quantizer state")
    promoted = R["Static"] is not None and isinstance(r, R["Static"])
    if type(r) is not R["TQ"] and not promoted:
        raise CannotAdopt(f"quantizer class {type(r).__name__}")
    if promoted and (not isinstance(d.get("_num_bits"), int) or getattr(r, "_lsq", False)
                     or getattr(r, "_global_amax", None) is not None):
        raise CannotAdopt("a promoted static-block quantizer outside the INT formats")
    spec = _calibrator_spec(r, R)
    bs = d.get("_block_sizes")
    cfg = QuantizerAttributeConfig(
        num_bits=tuple(d["_num_bits"]) if isinstance(d["_num_bits"], (list, tuple)) else d["_num_bits"], axis=d.get("_axis"),
        block_sizes=dict(bs) if bs else None, unsigned=bool(d.get("_unsigned", False)), narrow_range=bool(d.get("_narrow_range", False)),
        calibrator=spec or "max", enable=not d.get("_disabled", False), type="dynamic" if d.get("_dynamic") else "static",
        bias=dict(d["_bias"]) if d.get("_bias") else None, use_constant_amax=bool(d.get("_use_constant_amax", False)))
    t = TensorQuantizer(cfg, if_quant=d.get("_if_quant", True), if_calib=d.get("_if_calib", False))
    td = t.__dict__
    td["_constant_amax"] = d.get("_constant_amax")  # (the pinned value already sits in `_amax`: taken with the buffers)
    td["_enable_pre_quant_scale"] = d.get("_enable_pre_quant_scale", True)
    if spec is None:
        td["_calibrator"] = None
    elif spec == "max" and getattr(r._calibrator, "_calib_amax", None) is not None:
        # the running maximum of earlier calibrations (the reference never resets it: calib/max.py:66-84)
        cal, held = t._calibrator, r._calibrator._calib_amax
        cal._buf = held.detach().reshape(-1).float().clone()
        cal._shape, cal._dtype = tuple(held.shape), held.dtype
    for name in _BUFFERS:
        buf = r._buffers.get(name)
        if buf is not None:
            t.register_buffer(name, buf, persistent=name not in r._non_persistent_buffers_set)
    for name in _EXTRAS:
        if name in d:
            td[name] = d[name]
    if promoted:
        t.promote_static_block()
    t.train(r.training)
    return t


def _write_back(r, t, R):
    """release: the twin's state onto the reference's quantizer."""
    d, td = r.__dict__, t.__dict__
    for name in _FLAGS:
        if name in td:
            d[name] = td[name]
    if d.get("_axis") != td.get("_axis"):
        d["_axis"] = td["_axis"]
    if getattr(r, "_calibrator", None) is not None:
        r._calibrator._axis = td.get("_axis")  # (tensor_quantizer.py:246-249)
    if (d.get("_block_sizes") or None) != (td.get("_block_sizes") or None):
        d["_block_sizes"] = dict(td["_block_sizes"]) if td.get("_block_sizes") else None
    for name in _BUFFERS:
        mine = t._buffers.get(name)
        if mine is None:
            if name in r._buffers:
                del r._buffers[name]
                r._non_persistent_buffers_set.discard(name)
            continue
        r._buffers[name] = mine
    for name in _EXTRAS:
        if name in td:
            d[name] = td[name]
        else:
            d.pop(name, None)
    cal = getattr(r, "_calibrator", None)
    if cal is not None and type(cal) is R["calib"].MaxCalibrator and type(t._calibrator).__name__ == "MaxCalibrator":
        buf = t._calibrator._buf
        if buf is None:
            cal._calib_amax = None
        else:
            out = buf.detach().to(t._calibrator._dtype or buf.dtype)
            cal._calib_amax = out.reshape(t._calibrator._shape) if t._calibrator._shape is not None else out
    if "_block_reshape_size" in td and "_block_reshape_size" not in d and d.get("_block_sizes"):
        # The twin met its first tensor during the call and fixed its block layout; the reference's quantizer would have done
        # the same in the reference's own run (`_setup_for_blockquant`, tensor_quantizer.py:975-1043: `_original_shape`,
        # `_block_reshape_size`, `_padding`, `_slices`, and `_amax_shape_for_export`, which `export_amax` reads WITHOUT a
        # forward in between, :1095-1098).  The two sides lay N-D blocks out differently, so the values are not copied: the
        # reference computes its own from the shape the twin saw (a meta tensor: shapes only).
        shape = list(td["_original_shape"])
        for i, sl in enumerate(td.get("_slices") or ()):
            if isinstance(sl, slice) and sl.stop is not None:
                shape[i] = sl.stop  # (the unpadded extent)
        with contextlib.suppress(Exception):
            if getattr(r, "is_static_block_quant", False):
                r._setup_for_blockquant(torch.empty(shape, device="meta"))
    if getattr(t, "_is_static_block_scale_quantizer", False) and R["Static"] is not None and not isinstance(r, R["Static"]):
        R["Static"].from_tensor_quantizer(r)


class Adoption:
    """with Adoption(model): the reference's quantized `model` is this package's for the duration (module docstring)."""

    def __init__(self, model: nn.Module):
        self.model = model
        self.R = _ref()
        self.pairs: dict[int, tuple] = {}  # id(reference leaf quantizer) -> (reference, twin)
        self.swapped: list = []  # (parent, key, reference child)
        self.linears: list = []  # (module, reference class)
        self.forwards: list = []  # (module, the instance-level forward the reference's own awq_lite left on it)
        self.foreign_weighted: list = []  # names of modules that keep their class and hold weight quantizers of their own kind

    # -- what the model holds ---------------------------------------------------------------------------------------
    def _leaf(self, r):
        if id(r) not in self.pairs:
            self.pairs[id(r)] = (r, _twin(r, self.R))
        return self.pairs[id(r)][1]

    def _twin_child(self, child):
        R = self.R
        if isinstance(child, R["Seq"]):
            if not all(isinstance(q, R["TQ"]) for q in child):
                raise CannotAdopt("a SequentialQuantizer of something else than TensorQuantizers")
            return SequentialQuantizer(*[self._leaf(q) for q in child])
        if isinstance(child, R["TQ"]):
            special = type(child) is not R["TQ"] and not (R["Static"] is not None and type(child) is R["Static"])
            if special and child.__dict__.get("_disabled", False):
                return None  # a disabled quantizer of a special class (HardDisabledTensorQuantizer, ...): stays, it does nothing
            return self._leaf(child)
        return None

    def _check_module(self, name, m):
        R = self.R
        if not isinstance(m, R["QuantModule"]):
            return
        ps = m.__dict__.get("_parallel_state")
        if ps is not None:
            for g in ("tensor_parallel_group", "expert_model_parallel_group"):
                grp = getattr(ps, g, None)
                if grp is not None and getattr(grp, "is_initialized", lambda: False)():
                    raise CannotAdopt(f"{g} of {name or type(m).__name__}")
            # this package reduces statistics over the world (or the replicas declared to it); a reference module whose
            # data-parallel group is a proper SUBGROUP keeps the reference's own per-group reductions
            dp = getattr(ps, "data_parallel_group", None)
            if dp is not None and getattr(dp, "group", None) not in (None, -1):
                raise CannotAdopt(f"a data-parallel subgroup on {name or type(m).__name__}")
        weighted = [k for k, c in m._modules.items() if k.endswith("weight_quantizer") and c is not None]
        if not weighted:
            # a module that enumerates weights of its own kind (a fused MoE expert container: one quantizer per expert in a
            # ModuleList, plugins/huggingface.py:1085-1101) keeps its class; max / mse calibrate its weights through
            # model_calib._foreign_weight_pairs, a fold is the reference's (its per-module `fold_weight` overrides)
            if any(k.endswith("weight_quantizers") and c is not None and len(c) for k, c in m._modules.items()):
                self.foreign_weighted.append(name or type(m).__name__)
            return
        enabled = any(getattr(q, "is_enabled", True) for k in weighted
                      for q in (m._modules[k] if isinstance(m._modules[k], R["Seq"]) else [m._modules[k]]))
        # the class whose forward runs underneath the reference's quantized bases: nn.Linear itself (transformers' Conv1D is
        # converted to that, plugins/huggingface.py:559-571) or FalconLinear (input @ W.T + b, :1574-1590)
        base = next((c for c in type(m).__mro__ if not c.__module__.startswith("modelopt.")), None)
        linear_forward = base is nn.Linear or (base is not None and base.__name__ == "FalconLinear" and issubclass(base, nn.Linear))
        plain_linear = (linear_forward and weighted == ["weight_quantizer"]
                        and isinstance(m._modules.get("input_quantizer"), R["TQ"])
                        and isinstance(m._parameters.get("weight"), torch.Tensor) and m._parameters["weight"].dim() == 2)
        if plain_linear:
            # ... and nothing but the reference's two quantized bases between the module's class and that forward: a subclass
            # with a forward of its own (SVDQuantLinear's low-rank branch, RealQuantLinear's GEMM dispatch) computes something
            # this package's QuantLinear does not
            own = [c.__name__ for c in type(m).__mro__[:type(m).__mro__.index(base)]
                   if "forward" in c.__dict__ and c.__name__ not in ("QuantLinearConvBase", "QuantInputBase")]
            if own:
                raise CannotAdopt(f"{own[0]} ({name}) has a forward of its own")
            if m._parameters["weight"].is_meta or hasattr(m, "_hf_hook") and getattr(m._hf_hook, "offload", False):
                raise CannotAdopt("offloaded weights")
            if "_forward_pre_dm" in m.__dict__:
                # a forward the caller patched onto the instance BEFORE the conversion: the reference's quantized forward
                # calls it in place of the class's (quant_module.py:204-221, opt/dynamic.py:626-631)
                raise CannotAdopt(f"a forward patched before the conversion on {name or type(m).__name__}")
            patched = m.__dict__.get("forward")
            if patched is not None:
                # An instance-level `forward` shadows the class's -- swapping the class would change nothing.  The one kind
                # taken along: what the reference's own awq_lite leaves behind (`unpatch_forward_method`, utils/network.py:
                # 671-675: a bound method of this very module whose function is a forward of its class hierarchy); it is set
                # aside for the call and put back on release.  Anything else (an accelerate hook, a user's patch) is the caller's.
                fn = getattr(patched, "__func__", None)
                if getattr(patched, "__self__", None) is not m or not any(c.__dict__.get("forward") is fn for c in type(m).__mro__):
                    raise CannotAdopt(f"a patched forward on {name or type(m).__name__}")
            self.linears.append((m, type(m)))
        elif enabled:
            raise CannotAdopt(f"{type(m).__name__} ({name}) has an enabled weight quantizer and is not a plain quantized nn.Linear")

    def probe(self):
        """Every refusal (CannotAdopt) is raised here, before anything is touched; builds the twins."""
        R = self.R
        if isinstance(self.model, (R["TQ"], R["Seq"])):
            raise CannotAdopt("the root is a quantizer")
        self.linears, self._twins, self.foreign_weighted = [], [], []
        for name, m in self.model.named_modules():
            self._check_module(name, m)
        for m in list(self.model.modules()):
            if isinstance(m, (R["TQ"], R["Seq"])):
                continue
            for key, child in list(m._modules.items()):
                twin = self._twin_child(child) if child is not None else None
                if twin is not None:
                    self._twins.append((m, key, child, twin))
        return self

    def __enter__(self):
        if getattr(self, "_twins", None) is None:
            self.probe()
        for m, key, child, twin in self._twins:
            m._modules[key] = twin
            self.swapped.append((m, key, child))
        for m, _ in self.linears:
            m.__class__ = QuantLinear
            if "forward" in m.__dict__:
                self.forwards.append((m, m.__dict__.pop("forward")))
        return self

    def _restore_structure(self):
        for m, cls in self.linears:
            if m.__class__ is QuantLinear:
                m.__class__ = cls
        for m, f in self.forwards:
            m.__dict__.setdefault("forward", f)
        self.forwards = []
        for m, key, child in self.swapped:
            m._modules[key] = child

    def __exit__(self, et, ev, tb):
        self._restore_structure()
        for r, t in self.pairs.values():
            _write_back(r, t, self.R)
        return False


# -------------------------------------------------------------------------------------------------------------------
def _stats():
    from . import modelopt_plugin

    return modelopt_plugin.STATS


def _reference_numerics():
    """Under the reference, "the same result" is the reference's run on the SAME device: the flows' small fp32 scale formulas
    are evaluated as torch evaluates the reference's expressions on the tensors' device (numerics.py), not on the host."""
    return numerics.scale_math("device")


def _device_ok(model) -> str | None:
    from . import modelopt_plugin

    p = next((p for p in model.parameters()), None)
    if p is None:
        return "no parameters"
    if not modelopt_plugin._takes(p):
        return "not on the GPU"
    return None


def _adapter(name, original, run, rmc=None, precheck=None):
    """`run(model, forward_loop, **kwargs)` on the adopted model, or `original` when the model is not adoptable.
    rmc: the reference's model_calib module for an algorithm whose reference implementation starts with max_calibrate --
    that call's structural bookkeeping (shared-state children on the parents, :353-354) and its closing step (:143-157) then
    run on the reference's objects around the adopted section, as they would inside the reference's own function."""

    def algorithm(model, forward_loop=None, **kwargs):
        why = _device_ok(model) if isinstance(model, nn.Module) else "not a module"
        if why is None and precheck is not None:
            why = precheck(kwargs, model)
        adoption = patterns = None
        if why is None:
            try:
                adoption = Adoption(model)
                adoption.probe()
                if rmc is not None:
                    patterns = rmc.SharedWeightGlobalAmaxState.resolve_patterns(shared_states=kwargs.get("shared_states"))
                    rmc.SharedWeightGlobalAmaxState.attach(model, patterns=patterns)
                adoption.__enter__()
            except CannotAdopt as e:
                why, adoption = str(e), None
        if adoption is None:
            _stats()[f"S7:{name}:fallback:{why[:80]}"] += 1
            return original(model, forward_loop, **kwargs)
        _stats()[f"S7:{name}"] += 1
        try:
            with _reference_numerics():
                out = run(model, forward_loop, adoption=adoption, **kwargs)
        finally:
            adoption.__exit__(None, None, None)
        if rmc is not None:
            rmc._finalize_with_shared_state(model, patterns)
        return out

    algorithm._moq_seam = True
    algorithm.__name__ = getattr(original, "__name__", name)
    algorithm.__wrapped__ = original
    return algorithm


def _max_calibrate_adapter(rmc):
    original = rmc.max_calibrate

    def max_calibrate(model, forward_loop=None, distributed_sync=True, sync_expert_weight_amax=False, shared_states=None,
                      skip_forward_without_activation_calib=False):
        """model_calib.py:310-498 on the adopted model: this package's max_calibrate (one multi-tensor weight pass, one
        statistics launch per decoder layer, one bucketed MAX over the replicas); the reference's own shared-state
        bookkeeping and static-block promotion (:143-157) run on its objects around it."""
        why = _device_ok(model) if isinstance(model, nn.Module) else "not a module"
        adoption = None
        if why is None:
            try:
                adoption = Adoption(model).probe()
                # (the reference's structural bookkeeping first: it adds `_shared_quant_states` children, no quantizer state)
                patterns = rmc.SharedWeightGlobalAmaxState.resolve_patterns(shared_states=shared_states)
                rmc.SharedWeightGlobalAmaxState.attach(model, patterns=patterns)
                if (forward_loop is not None and skip_forward_without_activation_calib
                        and not rmc._needs_activation_forward_for_max_calib(model)):
                    forward_loop = None
                adoption.__enter__()
            except CannotAdopt as e:
                why, adoption = str(e), None
        if adoption is None:
            _stats()[f"S7:max_calibrate:fallback:{why[:80]}"] += 1
            return original(model, forward_loop, distributed_sync=distributed_sync, sync_expert_weight_amax=sync_expert_weight_amax,
                            shared_states=shared_states, skip_forward_without_activation_calib=skip_forward_without_activation_calib)
        _stats()["S7:max_calibrate"] += 1
        try:
            with _reference_numerics():
                _mc.max_calibrate(model, forward_loop, distributed_sync=bool(distributed_sync),
                                  sync_expert_weight_amax=bool(sync_expert_weight_amax))
        finally:
            adoption.__exit__(None, None, None)
        for _, module in model.named_modules():  # (:366-368)
            if hasattr(module, "layer_sync_moe_local_experts_amax"):
                module.layer_sync_moe_local_experts_amax(sync_weight_amax=sync_expert_weight_amax)
        rmc._finalize_with_shared_state(model, patterns)

    max_calibrate._moq_seam = True
    max_calibrate.__wrapped__ = original
    return max_calibrate


def _mse_precheck(kwargs, model=None):
    if kwargs.get("fp8_scale_sweep") or kwargs.get("shared_states"):
        return "fp8_scale_sweep / shared_states (NVFP4 static block scales)"
    return None


def _run_mse(model, forward_loop, adoption=None, distributed_sync=True, step_size=0.1, start_multiplier=0.25, stop_multiplier=4.0,
             fp8_scale_sweep=False, shared_states=None):
    return _mc.mse_calibrate(model, forward_loop, distributed_sync=distributed_sync, step_size=step_size,
                             start_multiplier=start_multiplier, stop_multiplier=stop_multiplier,
                             fp8_scale_sweep=fp8_scale_sweep, shared_states=shared_states)


def _run_smoothquant(model, forward_loop, adoption=None, alpha=1.0):
    return _mc.smoothquant(model, forward_loop, alpha=alpha)


_AWQ_OPTIONS = {"algorithm", "alpha_step", "debug", "max_co_batch_size", "max_tokens_per_batch", "min_clip_ratio", "shrink_step"}


def _awq_precheck(kwargs, model=None):
    """An option this package's search does not know (a newer reference) must not be dropped silently, and a weight format its
    search was never compared on must not be searched differently: hand the call back."""
    unknown = sorted(k for k, v in kwargs.items() if k not in _AWQ_OPTIONS and v is not None)
    if unknown:
        return f"awq option(s) {unknown}"
    if model is None:
        return None
    clip = kwargs.get("algorithm", "awq_lite") in ("awq_clip", "awq_full")
    for name, m in model.named_modules():
        wq = m._modules.get("weight_quantizer") if hasattr(m, "_modules") else None
        if wq is None or "input_quantizer" not in m._modules:
            continue
        first = wq[0] if isinstance(wq, nn.Sequential) and len(wq) else wq
        d = getattr(first, "__dict__", {})
        bs = d.get("_block_sizes")
        if d.get("_disabled", False) or not bs:
            continue
        # two-level block formats (NVFP4: E2M1 elements, E4M3 block scales under a tensor-wide amax, dynamic or static blocks) are
        # outside SURVEY section 8: their search runs, but was never held against the reference's (its own GPU test that
        # compares an offloaded run -- handed back -- with a resident one found the two apart)
        if bs.get("type", "static") != "static" or isinstance(bs.get("scale_bits"), (tuple, list)):
            return f"awq over a two-level / dynamic block format ({name})"
        # model_calib.awq_clip: signed static-block INT weight quantizers (the per-tensor NVFP4 branch, model_calib.py:1804-1813,
        # is outside this path)
        if clip and (not isinstance(d.get("_num_bits"), int) or d.get("_unsigned") or d.get("_narrow_range")):
            return f"awq_clip over a block format that is not signed static INT ({name})"
    return None


def _run_awq(model, forward_loop, adoption=None, algorithm="awq_lite", **kwargs):
    debug = bool(kwargs.pop("debug", False))
    out = _mc.awq(model, forward_loop, algorithm=algorithm, **({"debug": debug} if algorithm != "awq_lite" else {}), **kwargs)
    if not debug:  # (:1719-1720, :1937-1939: the helpers stay on the modules only for debugging)
        for m, _ in adoption.linears:
            for name in ("awq_lite", "awq_clip"):
                if name in m.__dict__:
                    delattr(m, name)
    return out


def _weight_only_quantize_adapter(rmc):
    original = rmc.weight_only_quantize

    def weight_only_quantize(model):
        """model_calib.py:187-199 on the adopted model: all per-tensor weight statistics in one multi-tensor launch."""
        why = _device_ok(model) if isinstance(model, nn.Module) else "not a module"
        adoption = None
        if why is None:
            try:
                adoption = Adoption(model).probe().__enter__()
            except CannotAdopt as e:
                why, adoption = str(e), None
        if adoption is None:
            _stats()[f"S7:weight_only_quantize:fallback:{why[:80]}"] += 1
            return original(model)
        _stats()["S7:weight_only_quantize"] += 1
        try:
            with _reference_numerics():
                _mc.weight_only_quantize(model)
        finally:
            adoption.__exit__(None, None, None)

    weight_only_quantize._moq_seam = True
    weight_only_quantize.__wrapped__ = original
    return weight_only_quantize


def _fold_weight_adapter(rmq):
    original = rmq.fold_weight

    def fold_weight(model, keep_attrs: bool = False):
        why = _device_ok(model) if isinstance(model, nn.Module) else "not a module"
        adoption = None
        if why is None:
            try:
                adoption = Adoption(model).probe()
                if adoption.foreign_weighted:
                    raise CannotAdopt(f"{adoption.foreign_weighted[0]} folds its own weights")
                adoption.__enter__()
            except CannotAdopt as e:
                why, adoption = str(e), None
        if adoption is None:
            _stats()[f"S7:fold_weight:fallback:{why[:80]}"] += 1
            return original(model, keep_attrs)
        _stats()["S7:fold_weight"] += 1
        try:
            with _reference_numerics():
                _mq.fold_weight(model, keep_attrs=keep_attrs)
        finally:
            adoption.__exit__(None, None, None)

    fold_weight._moq_seam = True
    fold_weight.__wrapped__ = original
    return fold_weight


# ------------------------------------------------------------------------------------------------------- export packers
def _pack_int4_adapter(original):
    from . import modelopt_plugin, ops

    def pack_int4_in_uint8(weight, weights_scaling_factor):
        """export/quant_utils.py:792-833 in one kernel (scale, round, clamp, transpose, two nibbles per byte)."""
        if (modelopt_plugin._takes(weight) and weight.dim() == 2 and weight.is_contiguous()
                and weight.dtype in (torch.float16, torch.bfloat16, torch.float32) and weight.shape[0] % 2 == 0
                and weights_scaling_factor.dim() == 2 and weight.shape[1] % weights_scaling_factor.shape[1] == 0):
            try:
                with _reference_numerics():
                    out = ops.pack_int4_in_uint8(weight, weights_scaling_factor.to(weight.device))
                _stats()["S7:pack_int4_in_uint8"] += 1
                return out
            except Exception as e:  # an unsupported layout: the reference's own code
                _stats()[f"S7:pack_int4_in_uint8:fallback:{type(e).__name__}"] += 1
        return original(weight, weights_scaling_factor)

    pack_int4_in_uint8._moq_seam = True
    pack_int4_in_uint8.__wrapped__ = original
    return pack_int4_in_uint8


_PACKED_FORMATS = ("fp8", "int8_sq", "int8_wo", "fp8_pc_pt")


def _to_quantized_weight_adapter(rqu):
    from . import export as _export
    from . import modelopt_plugin

    original = rqu.to_quantized_weight

    def to_quantized_weight(weight, weights_scaling_factor, quantization, weights_scaling_factor2=None, block_size=None):
        """export/quant_utils.py:836-931: per-tensor FP8, per-channel INT8 / FP8 of 2-D weights on the fused packers; INT4-AWQ
        / W4A8 reach pack_int4_in_uint8 (module global: the adapter above); everything else is the reference's own code."""
        if (isinstance(weight, torch.Tensor) and type(weight) in (torch.Tensor, nn.Parameter) and modelopt_plugin._takes(weight)
                and quantization in _PACKED_FORMATS and weight.dim() == 2 and weight.is_contiguous()
                and weight.dtype in (torch.float16, torch.bfloat16, torch.float32) and weights_scaling_factor is not None):
            try:
                with _reference_numerics():
                    out = _export.to_quantized_weight(weight, weights_scaling_factor.to(weight.device), quantization)
                _stats()[f"S7:to_quantized_weight:{quantization}"] += 1
                return out
            except Exception as e:
                _stats()[f"S7:to_quantized_weight:fallback:{quantization}:{type(e).__name__}"] += 1
        if (isinstance(weight, torch.Tensor) and type(weight) in (torch.Tensor, nn.Parameter) and modelopt_plugin._takes(weight)
                and quantization in ("mxfp4", "w4a8_mxfp4_fp8") and weight.dim() == 2 and weight.is_contiguous()
                and weight.dtype in (torch.float16, torch.bfloat16, torch.float32)
                and weight.shape[-1] % (block_size or 32) == 0 and weight.shape[-1] % 2 == 0):
            # MXFP4QTensor.quantize(weight, block_size)[0]._quantized_data (qtensor/mxfp4_tensor.py:37-81): E2M1 nibbles,
            # two per byte; the E8M0 block scales are written by the caller's own scale query
            try:
                from . import ops

                out = ops.mxfp4_quantize(weight, block_size or 32)[0]
                _stats()[f"S7:to_quantized_weight:{quantization}"] += 1
                return out
            except Exception as e:
                _stats()[f"S7:to_quantized_weight:fallback:{quantization}:{type(e).__name__}"] += 1
        return original(weight, weights_scaling_factor, quantization, weights_scaling_factor2, block_size)

    to_quantized_weight._moq_seam = True
    to_quantized_weight.__wrapped__ = original
    return to_quantized_weight


# --------------------------------------------------------------------------------------------------------------- install
def install_algorithms(swap, export: bool = True) -> list:
    """Re-point the reference's algorithm hooks (module docstring).  `swap(obj, name, new)` records what it replaces
    (modelopt_plugin._swap), so that modelopt_plugin.uninstall() puts the reference back."""
    import modelopt.torch.quantization.mode as rmode
    import modelopt.torch.quantization.model_calib as rmc
    import modelopt.torch.quantization.model_quant as rmq

    installed = []
    if getattr(rmc.max_calibrate, "_moq_seam", False):
        return ["S7:algorithms"]
    max_adapter = _max_calibrate_adapter(rmc)
    table = {
        "MaxCalibrateModeDescriptor": max_adapter,
        "MseCalibrateModeDescriptor": _adapter("mse_calibrate", rmc.mse_calibrate, _run_mse, rmc=rmc, precheck=_mse_precheck),
        "SmoothQuantModeDescriptor": _adapter("smoothquant", rmc.smoothquant, _run_smoothquant, rmc=rmc),
    }
    awq_adapter = _adapter("awq", rmc.awq, _run_awq, precheck=_awq_precheck)
    for cls in ("AWQLiteModeDescriptor", "AWQClipModeDescriptor", "AWQFullModeDescriptor"):
        table[cls] = awq_adapter
    for cls, fn in table.items():
        desc = getattr(rmode, cls, None)
        if desc is not None:
            swap(desc, "_calib_func", fn)
    installed.append("S7:_calib_func[max,mse,smoothquant,awq_lite,awq_clip,awq_full]")
    # module-level names: the reference's other algorithms (and user code) call these by name
    swap(rmc, "max_calibrate", max_adapter)
    swap(rmc, "weight_only_quantize", _weight_only_quantize_adapter(rmc))
    installed.append("S7:model_calib.max_calibrate,weight_only_quantize")
    fold = _fold_weight_adapter(rmq)
    swap(rmq, "fold_weight", fold)
    import modelopt.torch.quantization as rmtq

    if getattr(rmtq, "fold_weight", None) is fold.__wrapped__:
        swap(rmtq, "fold_weight", fold)
    installed.append("S7:model_quant.fold_weight")
    if export:
        try:
            import modelopt.torch.export.quant_utils as rqu
        except ImportError:
            rqu = None
        if rqu is not None:
            pack = _pack_int4_adapter(rqu.pack_int4_in_uint8)
            swap(rqu, "pack_int4_in_uint8", pack)
            tqw = _to_quantized_weight_adapter(rqu)
            swap(rqu, "to_quantized_weight", tqw)
            # importers that bound the names at import time
            import sys

            for modname in ("modelopt.torch.export.unified_export_hf", "modelopt.torch.export.layer_utils",
                            "modelopt.torch.export.model_config_export", "modelopt.torch.export.moe_utils"):
                mod = sys.modules.get(modname)
                if mod is None:
                    with contextlib.suppress(Exception):
                        mod = __import__(modname, fromlist=["_"])
                if mod is None:
                    continue
                if getattr(mod, "to_quantized_weight", None) is tqw.__wrapped__:
                    swap(mod, "to_quantized_weight", tqw)
                if getattr(mod, "pack_int4_in_uint8", None) is pack.__wrapped__:
                    swap(mod, "pack_int4_in_uint8", pack)
            installed.append("S7:export.to_quantized_weight,pack_int4_in_uint8")
    return installed
