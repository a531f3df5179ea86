"""quantize() -- the mtq.quantize(model, config, forward_loop) entry point for this path
(quantization/model_quant.py:147-250): convert Linears, apply wildcard quantizer configs, calibrate."""

from __future__ import annotations

import copy
import fnmatch
import re
import warnings

from torch import nn

from . import model_calib
from .nn import replace_quant_module
from .tensor_quantizer import QuantizerAttributeConfig, SequentialQuantizer, TensorQuantizer

# modelopt_recipes/configs/ptq/units/default_disabled_quantizers.yaml: module patterns every preset keeps in high
# precision (routers / MoE gates, lm_head and other output layers, conv1d mixers, the vision branch of multimodal
# models).  The reference's presets are [disable all, enable the format's quantizers, THIS list]; so are ours.  Its
# parent_class entries (BatchNorm*, LeakyReLU, Embedding) name modules this path never wraps.
DEFAULT_DISABLED_QUANTIZERS = (
    "*block_sparse_moe.gate*", "*linear_attn.conv1d*", "*linear_attn.in_proj_a*", "*linear_attn.in_proj_b*", "*lm_head*",
    "*mixer.conv1d*", "*mlp.gate.*", "*mlp.shared_expert_gate.*", "*output_layer*", "*proj_out.*", "*router*", "mtp.*",
    "output.*", "*embed_vision*", "*vision_tower*", "*visual*", "*vision_model*", "*multi_modal_projector*")


def _preset(quantizers: dict, algorithm) -> dict:
    return {"quant_cfg": {**quantizers, **{pat: {"enable": False} for pat in DEFAULT_DISABLED_QUANTIZERS}},
            "algorithm": algorithm}


# presets mirroring modelopt_recipes/configs/ptq/presets/model/{int8,fp8,int4_awq,mxfp4,int8_smoothquant}.yaml
INT8_DEFAULT_CFG = _preset({"*weight_quantizer": {"num_bits": 8, "axis": 0},
                            "*input_quantizer": {"num_bits": 8, "axis": None}}, "max")
# presets/model/int8_weight_only.yaml: per-channel INT8 weights, inputs untouched
INT8_WEIGHT_ONLY_CFG = _preset({"*weight_quantizer": {"num_bits": 8, "axis": 0},
                                "*input_quantizer": {"enable": False}}, "max")
FP8_DEFAULT_CFG = _preset({"*weight_quantizer": {"num_bits": (4, 3), "axis": None},
                           "*input_quantizer": {"num_bits": (4, 3), "axis": None}}, "max")
INT4_AWQ_CFG = _preset({"*weight_quantizer": {"num_bits": 4, "block_sizes": {-1: 128, "type": "static"}},
                        "*input_quantizer": {"enable": False}}, {"method": "awq_lite", "alpha_step": 0.1})
# presets/model/int4_blockwise_weight_only.yaml (numerics/int4_per_block.yaml: num_bits 4, block_sizes {-1: 128})
INT4_BLOCKWISE_WEIGHT_ONLY_CFG = _preset({"*weight_quantizer": {"num_bits": 4, "block_sizes": {-1: 128, "type": "static"}},
                                          "*input_quantizer": {"enable": False}}, "max")
# presets/model/fp8_2d_blockwise_weight_only.yaml: FP8 weights with one scale per 128 x 128 tile
FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG = _preset({"*weight_quantizer": {"num_bits": (4, 3), "block_sizes": {-1: 128, -2: 128}},
                                            "*input_quantizer": {"enable": False}}, "max")
_MXFP4_Q = {"num_bits": (2, 1), "block_sizes": {-1: 32, "type": "dynamic", "scale_bits": (8, 0)}}
MXFP4_DEFAULT_CFG = _preset({"*weight_quantizer": dict(_MXFP4_Q), "*input_quantizer": dict(_MXFP4_Q)}, None)


# presets/model/w4a8_mxfp4_fp8.yaml: MXFP4 block weights under per-tensor FP8 inputs (max-calibrated) -- the 4-bit weight
# format MI355X multiplies natively with an 8-bit activation; exported as w4a8_mxfp4_fp8
W4A8_MXFP4_FP8_CFG = _preset({"*weight_quantizer": dict(_MXFP4_Q), "*input_quantizer": {"num_bits": (4, 3), "axis": None}}, None)
# presets/model/mxfp4_mlp_weight_only.yaml: MXFP4 weights on the MLP / MoE projections only (everything else stays off:
# base_disable_all first, then the two enabling patterns)
MXFP4_MLP_WEIGHT_ONLY_CFG = {"quant_cfg": {"*": {"enable": False},
                                           "*mlp*weight_quantizer": dict(_MXFP4_Q),
                                           "*block_sparse_moe*weight_quantizer": dict(_MXFP4_Q),
                                           **{pat: {"enable": False} for pat in DEFAULT_DISABLED_QUANTIZERS}},
                             "algorithm": None}


def _mx_cfg(num_bits):  # numerics/mx*.yaml: blocks of 32 along the last dim, E8M0 block scales, weights and inputs
    q = {"num_bits": num_bits, "block_sizes": {-1: 32, "type": "dynamic", "scale_bits": (8, 0)}}
    return _preset({"*weight_quantizer": dict(q), "*input_quantizer": dict(q)}, None)


# presets/model/fp8_per_channel_per_token.yaml: per-output-channel FP8 weights, FP8 inputs with a dynamic abs-max per
# token ({-1: None}: the last dim is reduced; becomes `axis` on the first input).  Exported as fp8_pc_pt (E4M3 weights
# with an fp32 scale per output channel, no input scale: export.QUANTIZATION_FP8_PC_PT)
FP8_PER_CHANNEL_PER_TOKEN_CFG = _preset({"*weight_quantizer": {"num_bits": (4, 3), "axis": 0},
                                         "*input_quantizer": {"num_bits": (4, 3), "axis": None, "type": "dynamic",
                                                              "block_sizes": {-1: None}}}, "max")
MXFP8_DEFAULT_CFG = _mx_cfg((4, 3))  # presets/model/mxfp8.yaml
MXFP6_DEFAULT_CFG = _mx_cfg((3, 2))  # presets/model/mxfp6.yaml
MXINT8_DEFAULT_CFG = _mx_cfg(8)      # presets/model/mxint8.yaml
# presets/model/nvfp4.yaml quantizer layout (numerics/nvfp4.yaml): E2M1 elements in blocks of 16 with E4M3 block scales
# relative to a max-calibrated tensor-wide amax (two-level scaling, tensor_quant_mx.cu:154-183).  The format is NVIDIA's;
# what runs here is its fake quantization and calibration (no packed export)
_NVFP4_Q = {"num_bits": (2, 1), "axis": None, "block_sizes": {-1: 16, "type": "dynamic", "scale_bits": (4, 3)}}
NVFP4_DEFAULT_CFG = _preset({"*weight_quantizer": dict(_NVFP4_Q), "*input_quantizer": dict(_NVFP4_Q)}, "max")
# presets/model/w4a8_awq_beta.yaml: INT4 blocks then FP8 on the weights, FP8 inputs; AWQ-lite searches on the INT4
# stage with the input quantizers bypassed and max-calibrated per channel (model_calib.awq_lite)
_W4A8_Q = {"*weight_quantizer": [{"num_bits": 4, "block_sizes": {-1: 128, "type": "static"}},
                                 {"num_bits": (4, 3), "axis": None}],
           "*input_quantizer": {"num_bits": (4, 3), "axis": None}}
W4A8_AWQ_BETA_CFG = _preset(copy.deepcopy(_W4A8_Q), "awq_lite")
W4A8_MAX_CFG = _preset(copy.deepcopy(_W4A8_Q), "max")  # the same layout, max calibration only
INT8_SMOOTHQUANT_CFG = _preset({"*weight_quantizer": {"num_bits": 8, "axis": 0},
                                "*input_quantizer": {"num_bits": 8, "axis": None}},
                               "smoothquant")  # (the reference's literal; alpha = 1.0 is SmoothQuantCalibConfig's default)


# BASELINE configs[4]: MXFP4 (g = 32, E8M0 block scales) weights and inputs with SmoothQuant's per-channel scaling folded
# in first (model_calib.smoothquant(formats="all")).  The reference has no such preset -- its smoothquant skips every
# non-INT8 linear -- the composition is the one SURVEY 9.1 spells out: its INT8 scale math + its MX quantization
MXFP4_SMOOTHQUANT_CFG = _preset({"*weight_quantizer": dict(_MXFP4_Q), "*input_quantizer": dict(_MXFP4_Q)},
                                {"method": "smoothquant", "alpha": 0.5, "formats": "all"})


# presets/kv/fp8.yaml (units/kv_fp8.yaml): FP8 E4M3 per-tensor key / value quantizers, merged into a model preset
FP8_KV_CFG = {"quant_cfg": {"*[kv]_bmm_quantizer": {"num_bits": (4, 3), "axis": None, "enable": True}},
              "algorithm": "max"}


# presets/kv/fp8_affine.yaml (units/kv_fp8_affine.yaml): the same with an offset per head and channel (mean over batch
# and tokens of the [batch, heads, tokens, head_dim] states), exported as k_proj.k_bias / v_proj.v_bias
FP8_AFFINE_KV_CFG = {"quant_cfg": {"*[kv]_bmm_quantizer": {"num_bits": (4, 3), "axis": None, "enable": True,
                                                           "bias": {-2: None, -4: None, "type": "static"}}},
                     "algorithm": "max"}


# presets/kv/fp8_cast.yaml (units/kv_fp8_cast.yaml; hf_ptq.py's default `--kv_cache_qformat fp8_cast`): a plain E4M3 cast --
# amax fixed at the format's range (448: scale 1), nothing calibrated, no k_scale / v_scale in the checkpoint
FP8_CAST_KV_CFG = {"quant_cfg": {"*[kv]_bmm_quantizer": {"num_bits": (4, 3), "axis": None, "enable": True,
                                                         "use_constant_amax": True}},
                   "algorithm": "max"}


def update_quant_cfg_with_kv_cache_quant(quant_cfg: dict, kv_cache_quant_cfg: dict) -> dict:
    """utils/core_utils.py:1049-1075: a copy of `quant_cfg` with the KV-cache entries appended (later entries win);
    a config without an algorithm gets "max" so that the KV quantizers are calibrated."""
    import copy

    out = copy.deepcopy(quant_cfg)
    base = out.get("quant_cfg", {})
    if isinstance(base, dict) and isinstance(kv_cache_quant_cfg, dict):
        out["quant_cfg"] = {**base, **copy.deepcopy(kv_cache_quant_cfg)}
    else:  # list form: entries are simply appended
        out["quant_cfg"] = normalize_quant_cfg_list(base) + normalize_quant_cfg_list(copy.deepcopy(kv_cache_quant_cfg))
    if out.get("algorithm") is None:
        out["algorithm"] = "max"
    return out


def _apply_attrs(mod: TensorQuantizer, attrs: dict):
    attrs = dict(attrs)
    enable = attrs.pop("enable", True)
    if attrs:
        mod.set_from_attribute_config(QuantizerAttributeConfig(**{"narrow_range": False, **attrs, "enable": enable}))
    if enable:
        mod.enable()
    else:
        mod.disable()


_FUSED_EXPERTS_QUANTIZER_LIST_RE = re.compile(r"(weight_quantizers?|input_quantizers?)\.\d+(?=$|\.)")


def _normalize_fused_experts_quantizer_name(name: str) -> str:
    """conversion.py:317-341: `...gate_up_proj_weight_quantizers.3` also answers to `*weight_quantizer`."""
    return _FUSED_EXPERTS_QUANTIZER_LIST_RE.sub(lambda m: m.group(1).removesuffix("s"), name)


def normalize_quant_cfg_list(v) -> list[dict]:
    """config.py:1447-1569: the canonical quant_cfg is an ORDERED LIST of entries
    {"quantizer_name": wildcard, "cfg": attributes | [attributes, ...] | None, "enable": bool, "parent_class": str | None};
    also accepted: the legacy flat dict {wildcard: attributes} (its "default" key means "*"), single-key dict entries
    inside a list, and the legacy {"nn.<Class>": {wildcard: attributes}} scoping.  `enable` defaults to True when a cfg
    is given; an entry needs a cfg, an enable flag, or both."""
    if isinstance(v, dict):
        v = [{k: val} for k, val in v.items()]
    elif not isinstance(v, (list, tuple)):
        raise ValueError(f"quant_cfg must be a sequence of entries (or a legacy flat mapping), got {type(v).__name__}")

    def from_pair(key, value):
        if key == "default":
            key = "*"
        if isinstance(key, str) and key.startswith("nn."):
            if not isinstance(value, dict):
                raise ValueError(f"For 'nn.*' scoped format, value must be a mapping, got {value!r}")
            out = []
            for q_path, sub in value.items():
                sub = dict(sub)
                enable = sub.pop("enable", None)
                out.append({"parent_class": key, "quantizer_name": q_path, "cfg": sub or None, "enable": enable})
            return out
        if isinstance(value, dict):
            cfg = {k: val for k, val in value.items() if k != "enable"} or None
            return [{"quantizer_name": key, "cfg": cfg, "enable": value.get("enable")}]
        return [{"quantizer_name": key, "cfg": value, "enable": None}]

    out = []
    for raw in v:
        if isinstance(raw, dict) and "quantizer_name" in raw:
            entries = [dict(raw)]
        elif isinstance(raw, dict) and (len(raw) == 1 or any(str(k).startswith("nn.") for k in raw)):
            entries = [e for k, val in raw.items() for e in from_pair(k, val)]
        else:
            raise ValueError(f"Invalid quant_cfg entry: {raw!r}.")
        for e in entries:
            cfg, enable = e.get("cfg"), e.get("enable")
            if cfg is None and enable is None:
                raise ValueError(f"quant_cfg entry {e!r} must specify 'cfg', 'enable', or both.")
            if isinstance(cfg, (list, tuple)):
                cfg = [dict(c) for c in cfg]
            elif cfg is not None:
                cfg = dict(cfg)
            out.append({"quantizer_name": e["quantizer_name"], "cfg": cfg, "enable": True if enable is None else bool(enable),
                        "parent_class": e.get("parent_class")})
    return out


def _resolve_parent_class(name: str):
    """The module class a `parent_class` entry names ("nn.Linear", "nn.BatchNorm2d", ...): matched with isinstance
    against the quantizer's immediate parent module (conversion.py:284-294 looks the name up in its registry of
    quantized classes, which are subclasses of the originals)."""
    if name.startswith("nn.") and hasattr(nn, name[3:]):
        return getattr(nn, name[3:])
    raise ValueError(f"parent_class {name!r} not found (expected a torch.nn class name like 'nn.Linear')")


def set_quantizer_by_cfg(model: nn.Module, quant_cfg):
    """conversion.py:245-314 set_quantizer_by_cfg: the entries of `quant_cfg` (normalize_quant_cfg_list) are applied in
    order, later entries override earlier ones for any quantizer they match.  An entry with a cfg REPLACES the matched
    quantizer's attributes (unspecified ones go back to their defaults) and enables it unless it says otherwise; an
    entry without one only toggles `enable`.  A LIST of attribute dicts turns the quantizer into a SequentialQuantizer
    with one member per entry (conversion.py:296-321)."""
    # every configurable quantizer once: (name, fused-experts alias, parent module, attribute, owner for parent_class).
    # The index survives the loop: an entry may REPLACE a quantizer object (setattr on the parent), never move it, so the
    # current object is fetched from the parent each time.  (Walking named_modules and get_submodule per entry and
    # quantizer cost 0.11 s for a 32-layer Llama: 22 entries x 1 500 modules.)
    mods = dict(model.named_modules())
    index = []
    for name, mod in mods.items():
        if not isinstance(mod, (TensorQuantizer, SequentialQuantizer)):
            continue
        parent = mods[name.rpartition(".")[0]] if "." in name else model
        if isinstance(parent, SequentialQuantizer):
            continue  # members are configured through their container
        owner = parent
        if isinstance(owner, (nn.ModuleList, nn.ModuleDict)) and name.count(".") >= 2:
            owner = mods[name.rsplit(".", 2)[0]]  # per-expert quantizer lists hang off the experts
        index.append((name, _normalize_fused_experts_quantizer_name(name), parent, name.rpartition(".")[-1], owner))
    for entry in normalize_quant_cfg_list(quant_cfg):
        pattern, cfg, enable = entry["quantizer_name"], entry["cfg"], entry["enable"]
        parent_class = _resolve_parent_class(entry["parent_class"]) if entry["parent_class"] else None
        attrs = {"enable": enable} if cfg is None else cfg
        matches = re.compile(fnmatch.translate(pattern)).match  # (fnmatch.fnmatch on POSIX: case-sensitive, same regex)
        for name, normalized, parent, attr, owner in index:
            if not (matches(name) or (normalized != name and matches(normalized))):
                continue
            if parent_class is not None and not isinstance(owner, parent_class):
                continue
            cur = getattr(parent, attr)
            if isinstance(attrs, (list, tuple)):
                if not isinstance(cur, SequentialQuantizer) or len(cur) != len(attrs):
                    cur = SequentialQuantizer(*[TensorQuantizer() for _ in attrs])
                    setattr(parent, attr, cur)
                for q, a in zip(cur, attrs):
                    _apply_attrs(q, {**a, "enable": enable})
            else:
                if isinstance(cur, SequentialQuantizer):
                    if cfg is None:  # enable / disable broadcasts to the members
                        for q in cur:
                            _apply_attrs(q, attrs)
                        continue
                    cur = TensorQuantizer()
                    setattr(parent, attr, cur)
                _apply_attrs(cur, attrs if cfg is None else {**attrs, "enable": enable})


# what the last quantize() call spent where (wall-clock seconds; the device is drained at the three stage boundaries,
# where nothing is in flight that a later stage could have overlapped): convert = nn.Linear -> QuantLinear and the
# on-the-fly attention / expert wrappers, set_quantizers = the wildcard config walk, calibrate = the algorithm (for
# awq_lite its own stages are in model_calib.AWQ_LITE_STATS["stages_s"] and sum to this figure), validate = the
# warning for invalid amax / pre-quant scales mtq.calibrate ends with (one flattened test per device)
QUANTIZE_STATS: dict = {}


def _drain(model):
    import torch

    p = next(model.parameters(), None)
    if p is not None and p.is_cuda:
        torch.cuda.synchronize(p.device)


def quantize(model: nn.Module, config: dict, forward_loop=None) -> nn.Module:
    import time

    QUANTIZE_STATS.clear()
    stages = QUANTIZE_STATS.setdefault("stages_s", {})
    _drain(model)
    t0 = t = time.perf_counter()

    def stage(name):
        nonlocal t
        _drain(model)
        now = time.perf_counter()
        stages[name] = round(now - t, 4)
        QUANTIZE_STATS["total_s"] = round(now - t0, 4)
        t = now

    replace_quant_module(model)
    stage("convert")
    set_quantizer_by_cfg(model, config["quant_cfg"])
    stage("set_quantizers")
    _run_algorithm(model, config.get("algorithm", "max"), forward_loop)
    stage("calibrate")
    _warn_invalid_quantizer_state(model)
    stage("validate")
    return model


# algorithms that only write TensorQuantizer amax (QuantizeAlgorithmConfig._mutates_weights False, config.py:784-843):
# the only ones whose layer-by-layer checkpoints may leave the weights out
_AMAX_ONLY_ALGORITHMS = ("max", "mse", "local_hessian")


def _layerwise_options(method, kwargs: dict) -> dict | None:
    """The `layerwise` entry of an algorithm dict as LayerwiseConfig reads it (config.py:711-843): None / {} = off, a
    bool = {"enable": bool}, a dict with enable / get_qdq_activations_from_prev_layer / checkpoint_dir / save_every /
    calib_mutates_weights.  Same refusals: a checkpoint directory without enable, calib_mutates_weights=False for an
    algorithm that writes weights; GPTQ's next-layer inputs default to the quantized predecessor (:1243-1250)."""
    lw = kwargs.pop("layerwise", None)
    lw = {} if lw is None else {"enable": lw} if isinstance(lw, bool) else dict(lw)
    unknown = set(lw) - {"enable", "get_qdq_activations_from_prev_layer", "checkpoint_dir", "save_every",
                         "calib_mutates_weights", "capture"}
    if unknown:
        raise ValueError(f"unknown layerwise option(s) {sorted(unknown)}")
    if lw.get("checkpoint_dir") is not None and not lw.get("enable", False):
        raise ValueError("layerwise.checkpoint_dir requires layerwise.enable=True. "
                         "Set layerwise.enable=True or remove layerwise.checkpoint_dir.")
    if lw.get("calib_mutates_weights", True) is False and method not in _AMAX_ONLY_ALGORITHMS:
        raise ValueError(f"Algorithm '{method}' mutates layer weights in-place; calib_mutates_weights=False would lose "
                         "those updates on resume. Only max/mse/local_hessian (amax-only) support this flag.")
    if not lw.get("enable", False):
        return None
    return {"checkpoint_dir": lw.get("checkpoint_dir"),
            "get_qdq_activations_from_prev_layer": bool(lw.get("get_qdq_activations_from_prev_layer", method == "gptq")),
            "save_every": int(lw.get("save_every", 1)),
            "calib_mutates_weights": bool(lw.get("calib_mutates_weights", True)),
            "capture": lw.get("capture", "parent")}


def _run_algorithm(model: nn.Module, algo, forward_loop):
    """One calibration algorithm on the model -- or, with `layerwise.enable`, on one decoder layer at a time
    (wrapped_calib_func, mode.py:215-277: every algorithm of this path takes the layer-by-layer wrapper)."""
    method, kwargs = (algo, {}) if not isinstance(algo, dict) else (algo["method"], {k: v for k, v in algo.items() if k != "method"})
    if method is None:
        return model
    lw = _layerwise_options(method, kwargs)
    ratio = kwargs.pop("moe_calib_experts_ratio", None)  # QuantizeAlgorithmConfig.moe_calib_experts_ratio (config.py:791-806)
    if ratio is not None:
        from . import hf_moe

        hf_moe.set_moe_calib_experts_ratio(model, ratio)  # (mode.py:239-247: it stays on the blocks)
    if method == "max":  # MaxCalibConfig.distributed_sync (config.py): off for callers that synchronise by their own rules
        func = model_calib.max_calibrate
        kwargs = {"distributed_sync": bool(kwargs.get("distributed_sync", True)), "shard_weights": kwargs.get("shard_weights"),
                  "sync_expert_weight_amax": bool(kwargs.get("sync_expert_weight_amax", False)),
                  "defer_stats": kwargs.get("defer_stats")}
    elif method == "mse":
        func = model_calib.mse_calibrate
    elif method == "local_hessian":
        func = model_calib.local_hessian_calibrate
    elif method == "smoothquant":
        func = model_calib.smoothquant
    elif method in ("awq_lite", "awq_clip", "awq_full"):
        func, kwargs = model_calib.awq, {**kwargs, "algorithm": method}
    elif method == "gptq":
        # GPTQCalibConfig (config.py:1204-1250): perc_damp, block_size, fused
        from . import gptq as _gptq

        func = _gptq.gptq
        kwargs = {k: kwargs[k] for k in ("perc_damp", "block_size", "fused", "shard_weights", "report_mse") if kwargs.get(k) is not None}
    else:
        raise ValueError(f"algorithm {method!r} is outside this path")
    if lw is None:
        func(model, forward_loop, **kwargs)
        return model
    if forward_loop is None:
        raise ValueError("forward_loop is required for calibration but got None.")
    from . import layerwise as _layerwise

    _layerwise.layerwise_calibrate(model, forward_loop, func, **lw, **kwargs)
    return model


def _warn_invalid_quantizer_state(model: nn.Module):
    """The check mtq.calibrate ends with (quantization/model_quant.py:119-122): a warning for every `_amax` /
    `_pre_quant_scale` that holds a negative, infinite or NaN entry.  The reference asks every quantizer (three reductions
    and three host reads per buffer); here all buffers of a device go through ONE flattened test, and only a failing
    model is walked quantizer by quantizer for the messages."""
    import torch

    found = [(name, attr, getattr(mod, attr)) for name, mod in model.named_modules() if isinstance(mod, TensorQuantizer)
             for attr in ("_amax", "_pre_quant_scale") if isinstance(getattr(mod, attr, None), torch.Tensor)]
    by_device: dict = {}
    for _, _, t in found:
        if not t.is_meta and t.numel():
            by_device.setdefault(t.device, []).append(t.detach().reshape(-1).float())

    def clean(flat) -> bool:
        # Per-tensor / per-channel state is a few thousand numbers: one concatenation on the device (a kernel the model's own
        # forward has already loaded), one copy, the test on the host.  The device form costs four elementwise / reduction
        # code objects their first use in the process -- 46 ms of a 4.2 s FP8 calibration of Llama-3-8B
        # (profiles/r05g_fp8_flow_overhead.md); it stays for per-group state (hundreds of MB), where the copy would cost more
        if flat.numel() <= (1 << 22):
            host = flat.cpu()
            return bool((torch.isfinite(host) & (host >= 0)).all())
        return bool((torch.isfinite(flat) & (flat >= 0)).all())

    ok = all(clean(torch.cat(ts)) for ts in by_device.values())
    if not ok:
        owners = dict(model.named_modules())
        for name, attr, _ in found:
            owners[name].validate_attr(attr_name=attr, warn_error=True, name=name)


def calibrate(model: nn.Module, algorithm="max", forward_loop=None) -> nn.Module:
    """mtq.calibrate (quantization/model_quant.py:64-129): run a calibration algorithm on a model whose quantizers are in
    place -- "max", "mse", "local_hessian", "smoothquant", "awq_lite" / "awq_clip" / "awq_full", "gptq", or a dict with
    "method" and the algorithm's keyword arguments; None does nothing.  A forward_loop that takes no argument is accepted
    with the reference's deprecation warning; the model is calibrated in eval mode and put back."""
    import inspect
    import warnings

    if forward_loop is not None and len(inspect.signature(forward_loop).parameters) == 0:
        warnings.warn("forward_loop should take model as argument, but got forward_loop without any arguments. This usage "
                      "will be deprecated in future versions.", DeprecationWarning)
        zero_arg = forward_loop
        forward_loop = lambda _model: zero_arg()  # noqa: E731
    training = model.training
    model.eval()
    try:
        _run_algorithm(model, algorithm, forward_loop)
        _warn_invalid_quantizer_state(model)
    finally:
        model.train(training)
    return model


def postprocess_amax(model: nn.Module, key: str, post_process_fn) -> nn.Module:
    """mtq.postprocess_amax (:132-144): amax <- post_process_fn(amax) for every calibrated quantizer whose name matches."""
    import fnmatch

    assert isinstance(key, str), "key should be a string"
    for name, module in model.named_modules():
        if isinstance(module, TensorQuantizer) and hasattr(module, "_amax") and fnmatch.fnmatch(name, key):
            module.amax = post_process_fn(module.amax)
    return model


def _toggle(model: nn.Module, wildcard_or_filter_func, enable: bool):
    import fnmatch

    for name, module in model.named_modules():
        if not isinstance(module, TensorQuantizer):
            continue
        hit = wildcard_or_filter_func(name) if callable(wildcard_or_filter_func) else fnmatch.fnmatch(name, wildcard_or_filter_func)
        if hit:
            module.enable() if enable else module.disable()


def need_calibration(config) -> bool:
    """config.py:1827-1859: does this configuration need calibration DATA?  Yes for every algorithm other than None / "max";
    otherwise when some entry that is not a weight quantizer's is switched on and not dynamic."""
    if config["algorithm"] is not None and config["algorithm"] != "max":
        return True

    def static(cfg):
        return cfg.get("enable", True) and cfg.get("type", "") != "dynamic"

    for entry in normalize_quant_cfg_list(config.get("quant_cfg") or []):
        if "weight_quantizer" in entry["quantizer_name"]:
            continue  # weights are calibrated without data
        raw = entry.get("cfg")
        if isinstance(raw, (list, tuple)):
            if any(static(c) for c in raw):
                return True
            continue
        cfg = dict(raw or {})
        if entry.get("enable") is not None:
            cfg["enable"] = entry["enable"]
        if static(cfg):
            return True
    return False


def _matched_quantizers(model: nn.Module, wildcard_or_filter_func, parent_class):
    """conversion.py:344-370 (_match_quantizer): quantizers and quantizer chains whose name answers to the wildcard (a fused
    expert container's per-expert quantizers also under their singular name) or the filter function, optionally only those whose
    immediate parent is a `parent_class`."""
    mods = dict(model.named_modules())
    for name, mod in list(mods.items()):
        if not isinstance(mod, (TensorQuantizer, SequentialQuantizer)):
            continue
        if isinstance(wildcard_or_filter_func, str):
            normalized = _normalize_fused_experts_quantizer_name(name)
            if not (fnmatch.fnmatch(name, wildcard_or_filter_func)
                    or (normalized != name and fnmatch.fnmatch(normalized, wildcard_or_filter_func))):
                continue
        elif callable(wildcard_or_filter_func):
            if not wildcard_or_filter_func(name):
                continue
        else:
            raise NotImplementedError(f"Unsupported type {type(wildcard_or_filter_func)}")
        parent = mods[name.rpartition(".")[0]] if "." in name else model
        if parent_class is not None and not isinstance(parent, parent_class):
            continue
        yield name, mod, parent


def set_quantizer_attributes_partial(model: nn.Module, wildcard_or_filter_func, partial_attributes, parent_class=None):
    """conversion.py:443-512: MERGE a subset of attributes (a dict, or a list of dicts for a quantizer chain) into every matched
    quantizer; what is not named stays -- the calibrated amax too.  A dict is broadcast over the members of a chain; a list
    needs a chain."""
    if not isinstance(partial_attributes, (dict, list)):
        raise ValueError(f"Invalid type for attributes: {type(partial_attributes)}, expected dictionary or list of dict.")
    if isinstance(partial_attributes, list) and not all(isinstance(a, dict) for a in partial_attributes):
        raise ValueError("All elements in attributes list must be of type dict.")
    for _, mod, _ in list(_matched_quantizers(model, wildcard_or_filter_func, parent_class)):
        if isinstance(partial_attributes, list):
            if not isinstance(mod, SequentialQuantizer):
                raise ValueError(f"Attributes is a list but {mod} is not a SequentialQuantizer.")
            for q, a in zip(mod, partial_attributes):
                q.update_attributes(a)
        elif isinstance(mod, SequentialQuantizer):
            for q in mod:
                q.update_attributes(partial_attributes)
        else:
            mod.update_attributes(partial_attributes)


def set_quantizer_attribute(model: nn.Module, wildcard_or_filter_func, attribute, parent_class=None):
    """mtq.set_quantizer_attribute (conversion.py:602-620): the deprecated name of set_quantizer_attributes_partial."""
    warnings.warn("set_quantizer_attribute is deprecated, use set_quantizer_attributes_partial", DeprecationWarning, stacklevel=2)
    set_quantizer_attributes_partial(model, wildcard_or_filter_func, attribute, parent_class)


def set_quantizer_attributes_full(model: nn.Module, wildcard_or_filter_func, attributes, parent_class=None):
    """conversion.py:373-440: REPLACE the matched quantizers' attributes by a complete QuantizerAttributeConfig (unspecified
    fields go back to their defaults); a list of them turns the quantizer into a chain with one member per entry, a single
    config turns a chain back into one quantizer."""
    if not isinstance(attributes, (QuantizerAttributeConfig, list)):
        raise ValueError(f"Invalid type for attributes: {type(attributes)}, expected QuantizerAttributeConfig or list of "
                         "QuantizerAttributeConfig.")
    if isinstance(attributes, list) and not all(isinstance(a, QuantizerAttributeConfig) for a in attributes):
        raise ValueError("All elements in attributes list must be of type QuantizerAttributeConfig.")
    for name, mod, parent in list(_matched_quantizers(model, wildcard_or_filter_func, parent_class)):
        attr = name.rpartition(".")[-1]
        if isinstance(attributes, list):
            if not isinstance(mod, SequentialQuantizer):
                mod = SequentialQuantizer(*[TensorQuantizer() for _ in attributes])
                setattr(parent, attr, mod)
            elif len(attributes) != len(mod):
                warnings.warn(f"The number of attributes ({len(attributes)}) does not match the number of quantizers of {mod} "
                              "leading to partial assignment.")
            for q, a in zip(mod, attributes):
                q.set_from_attribute_config(a)
        else:
            if isinstance(mod, SequentialQuantizer):
                mod = TensorQuantizer()
                setattr(parent, attr, mod)
            mod.set_from_attribute_config(attributes)


class set_quantizer_by_cfg_context:
    """conversion.py:568-599: apply `quant_cfg` for the length of a `with` block and put every quantizer's ATTRIBUTES back
    afterwards (buffers -- the amax calibrated inside the block included -- are not part of what is saved).  Entries that would
    turn a quantizer into a chain are refused there, and here."""

    _SAVED = ("_disabled", "_num_bits", "_axis", "_block_sizes", "_dynamic", "_unsigned", "_narrow_range", "_bias",
              "_use_constant_amax", "_constant_amax", "_if_quant", "_if_calib")

    def __init__(self, quant_model: nn.Module, quant_cfg):
        self.model, self.cfg = quant_model, quant_cfg

    def __enter__(self):
        for entry in normalize_quant_cfg_list(self.cfg):
            assert not isinstance(entry["cfg"], (list, tuple)), "list of config not support."
        self.saved = [(q, {k: (dict(v) if isinstance(v, dict) else v) for k, v in q.__dict__.items() if k in self._SAVED},
                       q._calibrator) for q in model_calib._quantizers(self.model)]
        set_quantizer_by_cfg(self.model, self.cfg)
        return self

    def __exit__(self, *exc):
        for q, attrs, calibrator in self.saved:
            q.__dict__.update(attrs)
            q.__dict__["_calibrator"] = calibrator
            if calibrator is not None:
                calibrator._axis = attrs.get("_axis")
            q.drop_layout_caches()  # (filled inside the block for the temporary layout)
        return False


def disable_quantizer(model: nn.Module, wildcard_or_filter_func):
    """mtq.disable_quantizer (:698-700): by wildcard or by a filter function of the quantizer's name."""
    _toggle(model, wildcard_or_filter_func, False)


def enable_quantizer(model: nn.Module, wildcard_or_filter_func):
    """mtq.enable_quantizer (:703-705)."""
    _toggle(model, wildcard_or_filter_func, True)


def print_quant_summary(model: nn.Module, output_dir: str | None = None):
    """mtq.print_quant_summary (:709-725): one line per TensorQuantizer (name padded to 80 columns, then the quantizer's
    repr) and the count, printed -- or written to <output_dir>/.quant_summary.txt."""
    import os

    rows = []
    for name, mod in model.named_modules():
        if isinstance(mod, TensorQuantizer):
            rows.append(name.ljust(80) + " " + repr(mod))
    text = "\n".join(rows + [f"{len(rows)} TensorQuantizers found in model"])
    if not output_dir:
        print(text)
        return
    os.makedirs(output_dir, exist_ok=True)
    path = os.path.join(output_dir, ".quant_summary.txt")
    with open(path, "w", encoding="utf-8") as f:
        f.write(text + "\n")
    print(f"Quant summary saved to {path}")


# ------------------------------------------------------------------------------------------------ fold_weight
def _fold_kind(w, wq):
    """Which whole-model launch can fold this (weight, quantizer) pair; None: the quantizer's own forward."""
    import torch

    if (not w.is_cuda or not w.is_contiguous() or w.dtype not in (torch.float32, torch.float16, torch.bfloat16)
            or wq.pre_quant_scale is not None or w.data_ptr() % 16 or w.numel() == 0):
        return None
    nb = wq._num_bits if not isinstance(wq._num_bits, list) else tuple(wq._num_bits)
    amax = getattr(wq, "_amax", None)
    if wq._block_sizes is None and wq._axis is None and amax is not None and amax.numel() == 1:
        if nb == (4, 3):
            return ("fp8",)
        if isinstance(nb, int):
            return ("int", nb, bool(wq._unsigned), bool(wq._narrow_range))
    if wq.is_mx_format and isinstance(nb, (tuple, int)):
        g = wq._block_sizes.get(-1, None) or wq._block_sizes.get(w.dim() - 1, None)
        fmt = {(2, 1): "E2M1", (4, 3): "E4M3", (5, 2): "E5M2", (3, 2): "E3M2", (2, 3): "E2M3", 8: "INT8"}.get(nb)
        if g and fmt and w.shape[-1] % g == 0 and set(wq._block_sizes) <= {-1, w.dim() - 1, "type", "scale_bits"}:
            return ("mx", int(g), fmt)
    if (wq.is_static_block_quant and amax is None and isinstance(nb, int) and w.dim() == 2
            and set(wq._block_sizes) <= {-1, 1, "type"}):
        g = wq._block_sizes.get(-1, None) or wq._block_sizes.get(1, None)
        if g and w.shape[-1] % g == 0:
            return ("group", int(g), nb, bool(wq._unsigned), bool(wq._narrow_range))
    return None


def fold_weight(model: nn.Module, keep_attrs: bool = False, shard_weights: bool | None = None):
    """mtq.fold_weight (quantization/model_quant.py:728-736 -> QuantModule.fold_weight / _fold_weight_quantizer,
    nn/modules/quant_module.py:132-186): every fake-quant weight quantizer is baked into its weight in place and
    disabled; `_amax` / `_pre_quant_scale` are dropped unless `keep_attrs` (a kept pre-quant scale is made inactive).
    Sequential (W4A8) weight quantizers are left alone, as in the reference.

    The reference folds tensor by tensor through the quantizer's forward; here all weights that share a format are
    quantize-dequantized by ONE multi-tensor launch (per-tensor FP8 / INT-k with their calibrated amax, dynamic MX
    blocks, dynamic per-group INT-k), the rest go through their quantizer.

    shard_weights (data-parallel replicas, distributed.declare_data_parallel): the (weight, quantizer) units are dealt
    over the ranks, each rank folds its share and the folded weights are broadcast from their owners, so every replica
    ends up with every folded weight."""
    import torch

    from . import distributed as mdist
    from .hf_experts import is_quant_fused_experts
    from .multi_tensor import SegmentTable
    from .nn import is_quantized_linear

    units = []
    for m in model.modules():
        if is_quantized_linear(m) and isinstance(m.weight_quantizer, TensorQuantizer) and m.weight_quantizer.fake_quant:
            units.append((m.weight.data, m.weight_quantizer))
        elif is_quant_fused_experts(m):
            units += [(w.data if hasattr(w, "data") else w, q) for w, q in m.iter_weights_for_calibration()
                      if isinstance(q, TensorQuantizer) and q.fake_quant]
    shard = mdist.resolve_shard(shard_weights)
    # only contiguous weights can be received in place from their owner; the (rare) others are folded by every rank --
    # and so are weights whose storage appears in more than one unit (tied / shared tensors): dealt to different owners,
    # the second broadcast would overwrite the first fold, where a single rank (and the reference) folds them in sequence
    ptr_count = {}
    for w, _ in units:
        ptr_count[w.data_ptr()] = ptr_count.get(w.data_ptr(), 0) + 1

    def dealable(u):
        return u[0].is_contiguous() and ptr_count[u[0].data_ptr()] == 1

    dealt = [u for u in units if dealable(u)] if shard else []
    mine = mdist.shard_list(dealt) + [u for u in units if not dealable(u)] if shard else units
    with torch.no_grad():
        groups, single = {}, []
        for w, wq in mine:
            if wq._disabled and wq.pre_quant_scale is None:
                continue  # disabled and no pre-quant scale on this path: the forward is the identity
            # (a disabled quantizer that still carries an active pre-quant scale IS folded -- its forward multiplies by
            # the scale before the disabled early return, quant_module.py:144-158 and its docstring :163-165)
            kind = _fold_kind(w, wq)
            if kind is None:
                single.append((w, wq))
            else:
                groups.setdefault((kind, w.dtype, w.device), []).append((w, wq))
        for (kind, _, _), pairs in groups.items():
            ws = [w for w, _ in pairs]
            if kind[0] in ("fp8", "int"):
                tab = SegmentTable(ws, outputs=ws)
                tab.amax_flat.copy_(torch.cat([wq._amax.detach().reshape(1).float() for _, wq in pairs]))
                if kind[0] == "fp8":
                    tab.fake_quant_e4m3()
                else:
                    tab.fake_quant_int(kind[1], kind[2], kind[3])
            elif kind[0] == "mx":
                SegmentTable(ws, outputs=ws).mx_fused_amax_convert(kind[1], kind[2])
            else:
                SegmentTable(ws, outputs=ws, group_size=kind[1]).amax_qdq_int_group(kind[2], kind[3], kind[4])
        for w, wq in single:
            w.copy_(wq(w.float().contiguous()).to(w.dtype))  # quant_module.py:150: the quantizer sees the fp32 weight
        if shard:
            mdist.broadcast_from_owners([w for w, _ in dealt], group=mdist.replica_group())
        seen = set()
        for _, wq in units:
            if id(wq) in seen:
                continue
            seen.add(id(wq))
            wq.disable()
            if keep_attrs and hasattr(wq, "_pre_quant_scale"):
                wq._enable_pre_quant_scale = False
            elif not keep_attrs:
                for attr in ("_pre_quant_scale", "_amax"):
                    if hasattr(wq, attr):
                        delattr(wq, attr)
    return model
