"""Checkpoint export for the formats of this path -- mirror of the INT4-AWQ / FP8 / INT8 branch of
modelopt.torch.export (unified_export_hf.py:280-810, export/quant_utils.py:225-345, :792-938, :1285-1302, :1442-1547).

What runs where:
  * every pass over a weight tensor is a HIP kernel: the resmooth rescale W * old / new (moq_rescale_cols), the
    per-block amax recalibration (fused group amax), the nibble packer (moq_int4_pack_export), FP8 / INT8 casts;
  * the per-channel / per-block vectors (pre_quant_scale averages, weight scales = amax / maxbound) are tiny and are
    computed on the HOST in the reference's exact torch expressions, so the exported bytes do not depend on a GPU
    reduction order (the reference is pinned on its CPU run).
Byte-identity with the reference's exported tensors, given the same calibrated state, is tested against a
reference-generated tiny-Llama checkpoint (tests/golden/export_llama.npz).
"""

from __future__ import annotations

import json
import os
import warnings
from collections import defaultdict

import torch
from torch import nn

from . import numerics, ops
from .hf_experts import is_quant_fused_experts
from .nn import is_quantized_linear
from .tensor_quantizer import SequentialQuantizer

QUANTIZATION_NONE = None
QUANTIZATION_FP8 = "fp8"
QUANTIZATION_INT8_SQ = "int8_sq"
QUANTIZATION_INT8_WO = "int8_wo"
QUANTIZATION_INT4_AWQ = "int4_awq"
QUANTIZATION_W4A8_AWQ = "w4a8_awq"
QUANTIZATION_MXFP4 = "mxfp4"
QUANTIZATION_W4A8_MXFP4_FP8 = "w4a8_mxfp4_fp8"
QUANTIZATION_FP8_PB_WO = "fp8_pb_wo"
QUANTIZATION_MXFP8 = "mxfp8"
QUANTIZATION_FP8_PC_PT = "fp8_pc_pt"


def get_quantization_format(module) -> str | None:
    """export/quant_utils.py:506-700 for the quantizer settings of this path."""
    if not (is_quantized_linear(module) or isinstance(module, _ExpertProjection)):
        return QUANTIZATION_NONE
    wq, iq = module.weight_quantizer, module.input_quantizer
    if not wq.is_enabled:
        return QUANTIZATION_NONE
    if isinstance(wq, SequentialQuantizer):
        # export/quant_utils.py:503-516: INT4 blocks chained with FP8 is W4A8_AWQ -- nibbles packed on the block
        # scale of stage 0, the per-tensor scale of stage 1 travels as weight_scale_2
        nbs = [q._num_bits for q in wq]
        if not (len(wq) == 2 and nbs[0] == 4 and isinstance(nbs[1], (tuple, list)) and tuple(nbs[1]) == (4, 3)):
            raise NotImplementedError(f"export of the SequentialQuantizer weight format {nbs} is outside this path")
        assert wq[0].is_static_block_quant and wq[1].block_sizes is None, "Invalid block_sizes for SequentialQuantizer"
        return QUANTIZATION_W4A8_AWQ
    nb = wq._num_bits
    if nb == 4 and wq.is_static_block_quant:
        return QUANTIZATION_INT4_AWQ
    if nb == 8:
        return QUANTIZATION_INT8_SQ if iq.is_enabled else QUANTIZATION_INT8_WO
    if isinstance(nb, (tuple, list)) and tuple(nb) == (4, 3) and wq.block_sizes is None:
        axis = wq._axis
        if axis is not None:  # export/quant_utils.py:545-546: `weight_quantizer.axis == 0` -> per-channel weight scales
            if axis == 0 or (isinstance(axis, (tuple, list)) and tuple(axis) == (0,)):
                return QUANTIZATION_FP8_PC_PT
            raise NotImplementedError(f"export of FP8 weights quantized along axis {axis} is outside this path")
        return QUANTIZATION_FP8
    if (isinstance(nb, (tuple, list)) and tuple(nb) == (4, 3) and wq.block_sizes is not None
            and wq.block_sizes.get("type", "static") == "dynamic"
            and tuple(wq.block_sizes.get("scale_bits") or ()) == (8, 0)):
        return QUANTIZATION_MXFP8  # export/quant_utils.py:534-541
    if (isinstance(nb, (tuple, list)) and tuple(nb) == (4, 3) and wq.block_sizes is not None
            and wq.block_sizes.get("type", "static") != "dynamic"):
        return QUANTIZATION_FP8_PB_WO  # export/quant_utils.py:531-544 (fake-quant static blocks)
    if (isinstance(nb, (tuple, list)) and tuple(nb) == (2, 1) and wq.block_sizes is not None
            and tuple(wq.block_sizes.get("scale_bits", ())) == (8, 0)):
        inb = iq._num_bits if not isinstance(iq._num_bits, list) else tuple(iq._num_bits)
        if (wq.block_sizes.get("type", "static") == "dynamic" and iq.is_enabled and inb == (4, 3) and iq.block_sizes is None):
            return QUANTIZATION_W4A8_MXFP4_FP8  # export/quant_utils.py:567-574: MXFP4 weights under per-tensor FP8 inputs
        return QUANTIZATION_MXFP4  # export/quant_utils.py:585-586
    raise NotImplementedError(f"export of weight format num_bits={nb} block_sizes={wq.block_sizes} is outside this path")


def get_scaling_factor(quantizer) -> torch.Tensor | None:
    """export/quant_utils.py:225-243: export_amax().float() / maxbound."""
    if not quantizer.is_enabled:
        return None
    amax = quantizer.export_amax()
    if amax is None:
        return None
    # numerics "host" (default): IEEE fp32 division on the host -- torch's GPU kernel turns `tensor / python_scalar` into a
    # multiplication by the reciprocal (last-bit differences), the CPU kernel the reference fixtures come from divides;
    # "device": torch's own kernel on amax's device, i.e. what the reference's run on that device writes
    scaling_factor = numerics.div_scalar(amax, quantizer.maxbound)
    assert torch.all(scaling_factor > 0), f"scaling factor {scaling_factor} not positive."
    return scaling_factor


def get_weight_scaling_factor(module) -> torch.Tensor | None:
    wq = module.weight_quantizer
    return get_scaling_factor(wq[0] if isinstance(wq, SequentialQuantizer) else wq)  # export/quant_utils.py:277-278


def get_weight_scaling_factor_2(module) -> torch.Tensor | None:
    """export/quant_utils.py:338-345: the scale of the LAST stage of a two-stage sequential weight quantizer."""
    wq = module.weight_quantizer
    if not isinstance(wq, SequentialQuantizer) or not wq[-1].is_enabled:
        return None
    assert len(wq) == 2, "modelopt only supports 2 sequential quantization layers for now"
    return get_scaling_factor(wq[-1])


def _recollect_weight_amax(module):
    """"Redo weights collection" (export/quant_utils.py:1297-1301) on the weight quantizer alone; every stage of a
    sequential quantizer sees the unquantized weight (calibration passes the input through)."""
    wq = module.weight_quantizer
    stages = list(wq) if isinstance(wq, SequentialQuantizer) else [wq]
    state = [q._if_quant for q in stages]
    for q in stages:
        q.reset_amax()
        q.disable_quant()
        q.enable_calib()
    wq(module.weight)
    for q, was_quant in zip(stages, state):
        q.load_calib_amax()
        q.disable_calib()
        if was_quant:
            q.enable_quant()


@torch.no_grad()
def _update_pre_quant_scale(module, new_pre_quant_scale: torch.Tensor):
    """export/quant_utils.py:1285-1302: W <- (W.f32 * old.f32 / new.f32).to(dtype); input quantizer gets the new
    scale; the weight amax is re-collected."""
    old = module.input_quantizer._pre_quant_scale
    ops.rescale_cols(module.weight.data, old, new_pre_quant_scale, out=module.weight.data)
    module.input_quantizer.pre_quant_scale = new_pre_quant_scale
    _recollect_weight_amax(module)


@torch.no_grad()
def preprocess_linear_fusion(modules, resmooth_only: bool = False):
    """export/quant_utils.py:1476-1547: linears that share an input get one pre_quant_scale (the mean), one input
    amax (the max) and, for per-tensor weight formats, one weight amax (the max)."""
    fmts = [get_quantization_format(m) for m in modules]
    assert all(f == fmts[0] for f in fmts), "Modules have different quantization formats"
    iq0 = modules[0].input_quantizer
    if iq0.pre_quant_scale is not None:
        dev = modules[0].weight.device
        # tiny [Cin] vectors, the reference's expression `torch.mean(torch.stack(scales), dim=0)` where numerics.mode() puts
        # it: on the host (default: the reference's CPU run), or on the scales' own device -- torch's GPU mean multiplies the
        # fp32 sum by a rounded 1/N where its CPU mean divides, so a 16-bit mean can land one step apart (OPT fp16 on the
        # MI355X: 13 of 54 INT4-AWQ checkpoint tensors, all of them q / k / v and the LayerNorm their scale is folded
        # into, until this ran where the reference's does: tools/diag/opt_awq_device_diff.py)
        place = (lambda t: t.detach().cpu()) if numerics.on_host() else (lambda t: t.detach())
        avg = torch.mean(torch.stack([place(m.input_quantizer.pre_quant_scale) for m in modules]), dim=0)
        for m in modules:
            if not torch.equal(place(m.input_quantizer.pre_quant_scale), avg):
                _update_pre_quant_scale(m, avg.to(dev))
    if resmooth_only:
        return
    if iq0.is_enabled and iq0.amax is not None:
        assert iq0.amax.numel() == 1, "Only support scalar input quant amax"
        input_amax = torch.max(torch.stack([m.input_quantizer.amax for m in modules]))
        for m in modules:
            m.input_quantizer.amax = input_amax
    wq0 = modules[0].weight_quantizer
    if isinstance(wq0, SequentialQuantizer):  # :1515-1524: the per-tensor FP8 stage is unified, the blocks are not
        if wq0[-1].is_enabled:
            assert len(wq0) == 2
            weight_amax = torch.max(torch.stack([m.weight_quantizer[-1].amax for m in modules]))
            for m in modules:
                m.weight_quantizer[-1].amax = weight_amax
    elif wq0.is_enabled and wq0.amax is not None and wq0.amax.numel() == 1:
        weight_amax = torch.max(torch.stack([m.weight_quantizer.amax for m in modules]))
        for m in modules:
            m.weight_quantizer.amax = weight_amax


def _layernorm_uses_weight_plus_one(module) -> bool:
    """export/quant_utils.py:1432-1439: Gemma-style norms and zero-centred gammas scale by (1 + weight)."""
    n = type(module).__name__
    if any(k in n for k in ("LayerNorm1P", "GemmaRMSNorm", "Gemma2RMSNorm", "Gemma3RMSNorm")):
        return True
    return bool(getattr(module, "zero_centered_gamma", False))


@torch.no_grad()
def fuse_prequant_layernorm(layernorm_module, modules):
    """export/quant_utils.py:1442-1473: fold the (shared) pre_quant_scale into the preceding norm's weight."""
    if not hasattr(modules[0].input_quantizer, "_pre_quant_scale"):
        return
    pqs = modules[0].input_quantizer._pre_quant_scale.to(layernorm_module.weight.device)
    if _layernorm_uses_weight_plus_one(layernorm_module):
        fused = (layernorm_module.weight + 1.0) * pqs - 1.0  # the norm multiplies by (1 + weight)
    else:
        fused = layernorm_module.weight * pqs
    layernorm_module.weight = nn.Parameter(fused.to(layernorm_module.weight.dtype))
    if getattr(layernorm_module, "bias", None) is not None:
        layernorm_module.bias = nn.Parameter(layernorm_module.bias * pqs)
    for m in modules:
        delattr(m.input_quantizer, "_pre_quant_scale")
        m.fused_with_prequant = True


def is_layernorm(module) -> bool:
    """export/layer_utils.py is_layernorm: LayerNorm / *RMSNorm by class name."""
    n = type(module).__name__
    return any(k in n for k in ("LayerNorm", "RMSNorm", "RmsNorm"))


@torch.no_grad()
def collect_shared_input_modules(model, dummy_forward_fn):
    """unified_export_hf.py:280-348: run one probe forward with all quantizers off and record, by tensor identity,
    which quantized linears consume the same input and which norm produced it."""
    from .tensor_quantizer import TensorQuantizer

    input_to_linear, output_to_layernorm = defaultdict(list), {}
    handles = []
    for name, m in model.named_modules():
        if is_layernorm(m):
            m.name = name
            handles.append(m.register_forward_hook(lambda mod, i, o: output_to_layernorm.__setitem__(o, mod)
                                                   if isinstance(o, torch.Tensor) else None))
        elif is_quantized_linear(m) and (m.input_quantizer.is_enabled or m.weight_quantizer.is_enabled):
            m.name = name
            handles.append(m.register_forward_hook(lambda mod, i, o: input_to_linear[i[0]].append(mod)
                                                   if len(i) and isinstance(i[0], torch.Tensor) else None))
    quantizers = [q for q in model.modules() if isinstance(q, TensorQuantizer)]
    saved = [q._disabled for q in quantizers]
    try:
        for q in quantizers:
            q._disabled = True
        dummy_forward_fn()
    finally:
        for q, d in zip(quantizers, saved):
            q._disabled = d
        for h in handles:
            h.remove()
    return input_to_linear, output_to_layernorm


@torch.no_grad()
def requantize_resmooth_fused_llm_layers(model, dummy_forward_fn):
    """unified_export_hf.py:433-543 (dense LLM part): resmooth + unify every group of linears that share an input
    and, for AWQ formats, fold the shared pre_quant_scale into the norm that feeds them."""
    input_to_linear, output_to_layernorm = collect_shared_input_modules(model, dummy_forward_fn)
    fused = {}
    for tensor, modules in input_to_linear.items():
        fmt = get_quantization_format(modules[0])
        if len(modules) > 1 and fmt not in (QUANTIZATION_FP8, QUANTIZATION_NONE):
            preprocess_linear_fusion(modules)
            fused[modules[0].name] = [m.name for m in modules]
            if fmt is not None and "awq" in fmt and tensor in output_to_layernorm:
                fuse_prequant_layernorm(output_to_layernorm[tensor], modules)
    return fused


@torch.no_grad()
def to_quantized_weight(weight, weights_scaling_factor, quantization: str):
    """export/quant_utils.py:836-938 for the formats of this path."""
    if quantization in (QUANTIZATION_INT4_AWQ, QUANTIZATION_W4A8_AWQ):  # :911-912
        return ops.pack_int4_in_uint8(weight, weights_scaling_factor)
    wsf = weights_scaling_factor.to(weight.device)
    if quantization == QUANTIZATION_FP8:
        # (weight / wsf).to(float8_e4m3fn) with a 0-dim fp32 scaling factor: torch keeps the WEIGHT dtype for the
        # quotient (a 0-dim operand does not promote), so a 16-bit quotient is rounded to 16 bits before the cast --
        # one kernel: divide, round to dtype, cast
        if wsf.numel() != 1:
            raise NotImplementedError("FP8 export with a dimensioned weight scale (per-channel FP8) is outside this path")
        if not numerics.on_host() and weight.is_cuda and weight.dtype != torch.float32:
            # numerics "device": torch's GPU division functor runs in the COMMON dtype -- the 0-dim fp32 divisor is cast
            # to the 16-bit weight dtype before the fp32 quotient is formed (BinaryDivTrueKernel.cu: DivFunctor<scalar_t>),
            # where the CPU kernel keeps a 0-dim operand at full precision.  11 % of a bf16 weight's quotients differ
            # between the reference's two runs (tools/torch_cpu_vs_gpu_ops.py); the same kernel, the divisor rounded first
            wsf = wsf.to(weight.dtype).float()
        return ops.fp8_quantize(weight, wsf, fp32_scales=weight.dtype != torch.float32)
    if quantization in (QUANTIZATION_MXFP4, QUANTIZATION_W4A8_MXFP4_FP8):
        raise AssertionError("MXFP4 weights are packed together with their scales (export_quantized_weight)")
    if quantization in (QUANTIZATION_INT8_SQ, QUANTIZATION_INT8_WO):
        return ops.int8_pack_rows(weight, wsf)
    if quantization == QUANTIZATION_FP8_PC_PT:
        # (weight / wsf[:, None]).to(float8_e4m3fn) (export/quant_utils.py:879-909, 2-D weights): the fp32 [Cout, 1]
        # scaling factor is a dimensioned operand, so torch promotes the quotient to fp32 -- no rounding to the weight
        # dtype before the cast; one scale per row = 1 x Cin tiles of the tile packer
        if weight.dim() != 2 or wsf.numel() != weight.shape[0]:
            raise NotImplementedError("fp8_pc_pt export takes 2-D weights with one scale per output channel")
        return ops.fp8_quantize_tile(weight, wsf.float().reshape(-1, 1), 1, weight.shape[1])
    raise NotImplementedError(f"quantization format {quantization} not supported")


class _ExpertProjection:
    """One expert's 2-D projection cut out of a fused 3-D expert weight, dressed like a quantized linear for
    export_quantized_weight (the reference builds an nn.Module wrapper for the same purpose, moe_utils.py:196-201)."""

    def __init__(self, weight, weight_quantizer, input_quantizer):
        self.weight, self.weight_quantizer, self.input_quantizer, self.bias = weight, weight_quantizer, input_quantizer, None


def _needs_amax_fallback(q) -> bool:
    return q.is_enabled and (getattr(q, "_amax", None) is None or bool(torch.all(q._amax == 0)))


@torch.no_grad()
def export_fused_experts(module, dtype: torch.dtype) -> dict:
    """moe_utils.py:48-215: split the fused 3-D expert weights into per-expert 2-D projections (gated: gate_proj,
    up_proj, down_proj; non-gated: up_proj, down_proj) and export each like a quantized linear.  Keys:
    `<E>.<proj>.weight`, `.weight_scale`, `.input_scale`.  gate and up share the fused tensor's weight quantizer:
    a per-tensor amax is shared (runtimes re-fuse W1 / W3 under ONE scale), a per-row / per-block amax is sliced along
    dim 0 with the weight; quantizers that never got an amax fall back to the weight's own abs-max with a warning."""
    import copy
    import warnings

    first_attr = module._first_proj_attr
    first, down = getattr(module, first_attr).data, module.down_proj.data
    first_wqs = getattr(module, module._first_proj_weight_quantizers_attr)
    first_iq, down_iq = getattr(module, module._first_proj_input_quantizer_attr), module.down_proj_input_quantizer
    fused_rows = first.shape[1]
    inter = fused_rows // 2 if module._is_gated else None
    out = {}
    for idx in range(module.num_experts):
        fq = first_wqs[idx]
        if module._is_gated and _needs_amax_fallback(fq):
            if hasattr(fq, "_amax"):
                delattr(fq, "_amax")
            fq.amax = ops.reduce_amax(first[idx]).to(torch.float32)
            warnings.warn(f"Expert {idx} gate_up_proj weight quantizer was not calibrated (amax missing or zero). "
                          "Using fused-tensor amax as fallback (shared by gate and up).", stacklevel=2)
        if module._is_gated:
            projections = [("gate_proj", first[idx, :inter], 0, True), ("up_proj", first[idx, inter:], inter, True),
                           ("down_proj", down[idx], 0, False)]
        else:
            projections = [("up_proj", first[idx], 0, True), ("down_proj", down[idx], 0, False)]
        for proj, w, row0, uses_first in projections:
            src = fq if uses_first else module.down_proj_weight_quantizers[idx]
            wq = copy.deepcopy(src) if uses_first else src
            total = fused_rows if uses_first else down.shape[1]
            amax = getattr(wq, "_amax", None)
            if amax is not None and amax.dim() >= 1 and amax.numel() > 1:
                if amax.numel() != total and amax.numel() % total == 0:
                    amax = amax.contiguous().view(total, amax.numel() // total)
                rows = amax.shape[0]
                if total % rows == 0:
                    sliced = amax[row0 * rows // total:(row0 + w.shape[0]) * rows // total].contiguous()
                    keep_col = wq._amax.dim() == 2 and wq._amax.shape[-1] == 1  # [rows, 1] per-channel layout
                    delattr(wq, "_amax")
                    wq.amax = sliced.reshape(-1, 1) if keep_col else sliced
                else:
                    warnings.warn(f"Expert {idx} {proj}: fused amax dim0 ({rows}) does not evenly divide the fused "
                                  f"rows ({total}). Skipping amax slicing.", stacklevel=2)
            if _needs_amax_fallback(wq):
                if hasattr(wq, "_amax"):
                    delattr(wq, "_amax")
                wq.amax = ops.reduce_amax(w.contiguous()).to(torch.float32)
                warnings.warn(f"Expert {idx} {proj} weight quantizer was not calibrated (amax missing or zero). "
                              "Using weight-derived amax as fallback.", stacklevel=2)
            wrapper = _ExpertProjection(w.contiguous(), wq, first_iq if uses_first else down_iq)
            for k, v in export_quantized_weight(wrapper, dtype).items():
                out[f"{idx}.{proj}.{k}"] = v
    return out


# Checkpoint names that differ from the module tree: transformers >= 5 loads / saves some architectures through a key
# conversion (`transformers.conversion_mapping`), and its save_pretrained applies the REVERSE mapping to whatever state
# dict it is given -- which is how the reference's exported tensors get the checkpoint's names.  The same reverse
# mapping is derived here from transformers' own table: plain renamings (Mixtral: `.mlp.` -> `.block_sparse_moe.`) and
# the per-expert projections a fused-experts converter was built from (`experts.*.w1 / w3` -> `experts.gate_up_proj`
# tells that our `experts.N.gate_proj / up_proj` are `w1 / w3`).
def _checkpoint_rename_rules(model):
    """The reverse of transformers' checkpoint conversion for THIS model (quant_aware_conversion.py:392-470): the mapping is
    read per model -- a class-specific entry (GPT-NeoX's head: `^embed_out.` -> `lm_head.`) comes before the model type's --
    without the legacy renames, and every entry is reversed by transformers itself, so anchored patterns come out right.
    ("regex", pattern, replacement, scope prefixes) for a WeightRenaming; ("expert", ours, theirs) for the per-expert leaf names
    of a fused-experts converter (the export expands those into per-expert linears already)."""
    import re

    try:
        from transformers import conversion_mapping as cm

        mapping = getattr(model, "_weight_conversions", None)
        if mapping is None:
            mapping = cm.get_model_conversion_mapping(model, add_legacy=False)
    except Exception:  # noqa: BLE001 -- older transformers: module trees and checkpoints share their names
        mapping = None
    rules = []
    for item in mapping or []:
        src = list(getattr(item, "source_patterns", []) or [])
        tgt = list(getattr(item, "target_patterns", []) or [])
        if type(item).__name__ == "WeightRenaming":
            try:
                rev = item.reverse_transform()
            except Exception as e:  # noqa: BLE001
                # all or nothing (the reference's _build_reverse_rules, export/quant_aware_conversion.py: a rule it cannot
                # reverse keeps the module-tree names for tensors AND tables): a checkpoint with only some keys renamed
                # would mix two namespaces
                warnings.warn(f"checkpoint names: a renaming rule of {type(model).__name__} cannot be reversed ({e}); the "
                              "checkpoint keeps the module names")
                return []
            scope = getattr(rev, "scope_prefix", None)
            prefixes = ()
            if scope is not None:  # a sub-model's rule applies under its own path only (:323-345)
                dot = f"{scope}." if scope != "" else ""
                base = getattr(rev, "base_model_prefix", None) or ""
                prefixes = tuple(dict.fromkeys(([f"{base}.{dot}"] if base else []) + [dot]))
            pats, repls = getattr(rev, "source_patterns", []), getattr(rev, "target_patterns", [])
            pats, repls = ([pats] if isinstance(pats, str) else list(pats or [])), ([repls] if isinstance(repls, str) else list(repls or []))
            for pattern, repl in zip(pats, repls):
                rules.append(("regex", re.compile(pattern), repl, prefixes))
        elif type(item).__name__ == "WeightConverter" and len(tgt) == 1 and tgt[0].endswith(("experts.gate_up_proj", "experts.down_proj")):
            ours = ["gate_proj", "up_proj"] if tgt[0].endswith("gate_up_proj") else ["down_proj"]
            for our_name, pat in zip(ours, src):
                m = re.search(r"experts\.\*\.([A-Za-z0-9_]+)\.weight$", pat)
                if m and m.group(1) != our_name:
                    rules.append(("expert", our_name, m.group(1)))
        elif type(item).__name__ == "WeightConverter":
            # a converter other than the two fused-experts shapes (a dense fused tensor the hub stores in pieces, a scoped
            # converter): un-fusing it is outside this path -- module names for everything, as above
            warnings.warn(f"checkpoint names: {type(model).__name__} has a weight converter outside this path "
                          f"({src} -> {tgt}); the checkpoint keeps the module names")
            return []
    return rules


def _keeps_module_names(model) -> bool:
    """True when the model holds a fused-experts container whose exported scales have 3 or more dims (2-D FP8 blocks):
    the model-level form of rename_to_checkpoint_keys' guard, for callers without the tensors at hand (the quantization
    tables; a rank whose shard of the checkpoint happens to hold no expert)."""
    return any(is_quant_fused_experts(m) and getattr(m, m._first_proj_weight_quantizers_attr)[0].is_enabled
               and get_quantization_format(_ExpertProjection(None, getattr(m, m._first_proj_weight_quantizers_attr)[0],
                                                             getattr(m, m._first_proj_input_quantizer_attr)))
               == QUANTIZATION_FP8_PB_WO for m in model.modules())


def rename_to_checkpoint_keys(state: dict, model) -> dict:
    import re

    rules = _checkpoint_rename_rules(model)
    if not rules:
        return state
    if any(rule[0] == "expert" for rule in rules):
        # The reference's reversal is all-or-nothing (unified_export_hf.py:1594-1613): its guard against experts that
        # were not expanded (quant_aware_conversion.py:298-320) takes ANY tensor of 3 or more dims under `.experts.` for a
        # stacked expert weight -- which the [R/br, 1, C/bc, 1] scales of 2-D FP8 blocks are -- and the whole checkpoint
        # (and the module names in the quantization tables) then keeps the module tree's names.  Mirrored, so that both
        # libraries write the same file for the same model.
        bad = next((k for k, v in state.items() if ".experts." in k and getattr(v, "ndim", 0) >= 3), None)
        if bad is not None or _keeps_module_names(model):
            warnings.warn("checkpoint names not restored (a tensor under `.experts.` has 3 or more dims"
                          + (f": '{bad}'" if bad else "") + ", which the reference's reversal refuses): tensors and "
                          "quantization tables keep the module tree's names")
            return state
    return {_rename_key(k, rules): v for k, v in state.items()}


def _rename_key(k: str, rules) -> str:
    import re

    for rule in rules:
        if rule[0] == "expert":
            k = re.sub(rf"(\.experts\.\d+\.){re.escape(rule[1])}\.", rf"\g<1>{rule[2]}.", k)
    for rule in rules:
        if rule[0] == "regex":  # in order, each under its scope (quant_aware_conversion.py:173-194)
            _, pattern, repl, prefixes = rule
            if not prefixes:
                k = pattern.sub(repl, k)
            else:
                hit = next((p for p in prefixes if k.startswith(p)), None)
                if hit is not None:
                    k = hit + pattern.sub(repl, k[len(hit):])
    return k


def _module_name_mapper(model):
    """build_reverse_name_mapper (quant_aware_conversion.py:236-277): the same rules on the module names and wildcard patterns of
    the quantization tables, which are summarised and sorted under the module tree's names FIRST and mapped entry by entry
    afterwards (revert_quant_config_names, :280-295) -- a sentinel path segment lets rules that end in a separator match a
    bare module name, a trailing wildcard is set aside and put back."""
    rules = [] if _keeps_module_names(model) else _checkpoint_rename_rules(model)
    if not rules:
        return lambda name: name
    sentinel = ".\x00name_sentinel"

    def mapped(name: str) -> str:
        base, suffix = name, ""
        if name.endswith(".*"):
            base, suffix = name[:-2], ".*"
        elif name.endswith("*"):
            base, suffix = name[:-1], "*"
        out = _rename_key(base + sentinel, rules)
        return (out[:-len(sentinel)] if out.endswith(sentinel) else out) + suffix

    return mapped


def _with_pre_quant_scale(out: dict, iq) -> dict:
    """unified_export_hf.py:1121-1138: a smoothing scale left on the input quantizer is promoted to <module>.pre_quant_scale,
    whatever the linear's format (an AWQ search over a model with per-layer format overrides smooths those layers too)."""
    pqs = getattr(iq, "_pre_quant_scale", None)
    if pqs is not None:
        out["pre_quant_scale"] = pqs.detach().clone()
    return out


@torch.no_grad()
def export_quantized_weight(module, dtype: torch.dtype):
    """unified_export_hf.py:569-810 for one quantized linear: returns the tensors the checkpoint stores for it
    ({'weight', 'weight_scale'[, 'input_scale'][, 'pre_quant_scale']})."""
    fmt = get_quantization_format(module)
    if fmt is QUANTIZATION_NONE:
        return {"weight": module.weight.detach()}
    wq, iq = module.weight_quantizer, module.input_quantizer
    out = {}
    if fmt in (QUANTIZATION_MXFP4, QUANTIZATION_W4A8_MXFP4_FP8):
        # export/quant_utils.py:304-307, :935-936: MXFP4QTensor.quantize gives the packed nibbles and the E8M0 scale bytes
        # in one pass (moq_mxfp4_pack); the scales are stored as [..., Cin / block].  w4a8_mxfp4_fp8 stores the same two
        # tensors plus, once its per-tensor FP8 input quantizer is calibrated, `input_scale` (unified_export_hf.py:685-696:
        # every format's enabled input quantizer with an amax; MX inputs are dynamic and have none)
        block = wq.block_sizes.get(-1) or wq.block_sizes.get(module.weight.dim() - 1)
        w = module.weight.detach().to(dtype)
        packed, e8m0 = ops.mxfp4_quantize(w, block)
        out = {"weight": packed, "weight_scale": e8m0.reshape(*w.shape[:-1], -1)}
        if iq.is_enabled and iq.amax is not None:
            out["input_scale"] = get_scaling_factor(iq).squeeze()
        pqs = getattr(iq, "_pre_quant_scale", None)
        if pqs is not None:  # SmoothQuant scaling composed with MXFP4: promoted like every other format's (:1121-1138)
            out["pre_quant_scale"] = pqs.detach().clone()
        return out
    if fmt == QUANTIZATION_MXFP8:
        # unified_export_hf.py:671-679, export/quant_utils.py:871-872: E8M0 scale bytes [Cout, Cin / 32] from the block
        # abs-max, E4M3 elements from the tile pack kernel
        from .qtensor import MXFP8QTensor

        w = module.weight.detach().to(dtype)
        e8m0 = MXFP8QTensor.get_weights_scaling_factor_from_quantizer(w, wq)
        return _with_pre_quant_scale({"weight": MXFP8QTensor.quantize_with_scale(w, e8m0), "weight_scale": e8m0}, iq)
    if fmt == QUANTIZATION_FP8_PB_WO:
        # export/quant_utils.py:874-877: FP8QTensor.quantize(weight, scale.squeeze(), block_sizes on both axes); the
        # scale keeps the quantizer's amax shape [R/br, 1, C/bc, 1]
        from .qtensor import FP8QTensor

        blocks = {d: b for d, b in wq.block_sizes.items() if isinstance(d, int)}
        weight_scale = get_weight_scaling_factor(module)
        w = module.weight.detach().to(dtype)
        qt, _ = FP8QTensor.quantize(w, weight_scale.squeeze(), block_sizes=blocks)
        return _with_pre_quant_scale({"weight": qt._quantized_data, "weight_scale": weight_scale}, iq)
    if fmt == QUANTIZATION_FP8:
        amax = wq._amax.to(torch.float32)
        # unified_export_hf.py:635-643 decides by the amax buffer's RANK, not its size: a [1] buffer takes python float
        # division of amax.item(), everything else -- a 0-dim per-tensor amax too -- `amax / maxbound` as a tensor op (on a
        # GPU a multiplication by the reciprocal: the two forms differ there, which only a run beside the reference on the
        # device shows -- on the host both are IEEE division)
        weight_scale = (torch.tensor(amax.item() / wq.maxbound) if amax.dim() == 1
                        else numerics.div_scalar(amax, wq.maxbound))
    else:
        weight_scale = get_weight_scaling_factor(module)
        if fmt in (QUANTIZATION_INT8_SQ, QUANTIZATION_INT8_WO, QUANTIZATION_FP8_PC_PT) and weight_scale.dim() > 1:
            # per-channel amax is kept as [Cout, 1]; the checkpoint stores [Cout] (export_amax squeezes the kept-dims
            # shape, tensor_quantizer.py:1087-1117) and to_quantized_weight divides by wsf[:, None]
            weight_scale = weight_scale.reshape(-1)
    if iq.is_enabled and iq.amax is not None:
        out["input_scale"] = get_scaling_factor(iq).squeeze()
    out["weight"] = to_quantized_weight(module.weight.detach().to(dtype), weight_scale, fmt)
    out["weight_scale"] = weight_scale
    if fmt == QUANTIZATION_W4A8_AWQ:  # unified_export_hf.py:696-709
        out["weight_scale_2"] = get_weight_scaling_factor_2(module).squeeze()
    pqs = getattr(iq, "_pre_quant_scale", None)
    if pqs is not None:  # unified_export_hf.py:1121-1138: promoted to <module>.pre_quant_scale
        out["pre_quant_scale"] = pqs.detach().clone()
    return out


@torch.no_grad()
def export_state_dict(model, dtype: torch.dtype, dummy_forward_fn=None, shard_weights: bool | None = None) -> dict:
    """Checkpoint tensors of a quantized model: resmooth / fuse (when a probe forward is given), then pack every
    quantized linear; everything else is copied through.

    shard_weights (data-parallel replicas; None follows distributed.declare_data_parallel): the packing -- the pass
    over every weight that produces the checkpoint bytes -- is dealt over the ranks (quantized linears / expert
    containers in module order, distributed.shard_list) and NOT gathered: the returned dict holds this rank's packed
    modules only (rank 0 also everything that is copied through), for `save_checkpoint` to write as this rank's
    safetensors shard.  The union over the ranks is the single-rank state dict, byte for byte."""
    from . import distributed as mdist

    if dummy_forward_fn is not None:
        requantize_resmooth_fused_llm_layers(model, dummy_forward_fn)
    shard = mdist.resolve_shard(shard_weights)
    world, me = 1, 0
    if shard:
        import torch.distributed as dist

        world, me = dist.get_world_size(mdist.replica_group()), dist.get_rank(mdist.replica_group())
    state = {}
    handled = set()
    unit = 0
    for name, m in model.named_modules():
        prefix = name + "." if name else ""
        if is_quantized_linear(m):
            # linears whose quantizers are off (lm_head, routers) are copied through: rank 0, like every other
            # plain tensor (tied-weight aliases are resolved against the tensors of the same shard)
            packed = get_quantization_format(m) is not QUANTIZATION_NONE
            if (unit % world if packed else 0) == me:
                for k, v in export_quantized_weight(m, dtype).items():
                    state[prefix + k] = v
                if m.bias is not None:
                    state[prefix + "bias"] = m.bias.detach()
            unit += int(packed)
            handled.add(name)
        elif is_quant_fused_experts(m):
            if unit % world == me:
                for k, v in export_fused_experts(m, dtype).items():
                    state[prefix + k] = v
            unit += 1
            handled.add(name)
    kv_format = get_kv_cache_format(model)
    for k, v in model.state_dict().items():
        owner = k.rsplit(".", 1)[0] if "." in k else ""
        if any(owner == h or owner.startswith(h + ".") for h in handled):
            continue  # quantizer buffers (_amax, _pre_quant_scale) and raw weights of exported linears
        if me != 0:
            continue  # the tensors that are copied through travel in rank 0's shard
        new_key, value = _postprocess_kv_key(k, v, kv_format)
        if new_key is not None:
            state[new_key] = value
    for alias in _tied_alias_keys(model, state):
        del state[alias]
    if not shard:
        state = _in_module_tree_order(state, model)
    # the dict remembers HOW it was produced: save_checkpoint writes a per-rank shard exactly when this is one
    return ExportedState(rename_to_checkpoint_keys(state, model), sharded=bool(shard))


_SCALE_RANK = {"weight": 0, "bias": 0, "weight_scale": 1, "input_scale": 2, "weight_scale_2": 3, "pre_quant_scale": 4}
_EXPERT_PROJ_RANK = {"gate_proj": 0, "up_proj": 1, "down_proj": 2}


def _in_module_tree_order(state: dict, model) -> dict:
    """The reference's checkpoint dict is the model's own state_dict() -- walked module by module, parameters before buffers --
    with the exporter's buffers registered on the linears (`weight_scale`, then `input_scale`, `weight_scale_2`, and
    `pre_quant_scale` promoted last: unified_export_hf.py:630-720, :1121-1138), every expert container expanded in place
    (expert by expert; gate, up, down) and the KV-cache amax renamed where it stood (export/quant_utils.py:1000-1060).  The ORDER
    of that dict decides which tensor lands in which file once `max_shard_size` splits the checkpoint, so the same order is
    produced here: a single file holds its tensors sorted either way, a 47 GB Mixtral checkpoint gets the same five files."""
    import re

    place = {k: i for i, k in enumerate(model.state_dict().keys())}
    kv_source = {new: old for old, new in _KV_CACHE_REPLACEMENTS.items()}
    # where every `...experts` container's first tensor stood -- one pass over the model's keys (asking per exported key would
    # be (exported keys) x (state-dict keys) prefix tests: hours for 58 layers x 256 per-expert linears)
    anchors: dict = {}
    for k, pos in place.items():
        head = k
        while "." in head:
            head = head.rpartition(".")[0]
            if head.endswith("experts") and pos < anchors.get(head, len(place) + 1):
                anchors[head] = pos

    def rank(item):
        i, key = item
        owner, _, leaf = key.rpartition(".")
        m = re.match(r"(.*\bexperts)\.(\d+)\.([A-Za-z0-9_]+)\.([A-Za-z0-9_]+)$", key)
        if m and m.group(3) in _EXPERT_PROJ_RANK:  # an expanded container: where its first fused parameter stood
            anchor = anchors.get(m.group(1))
            if anchor is not None:
                return (anchor, 0, int(m.group(2)), _EXPERT_PROJ_RANK[m.group(3)], _SCALE_RANK.get(m.group(4), 5), i)
        if key in place:
            return (place[key], 0, 0, 0, 0, i)
        if leaf in _SCALE_RANK:  # an exporter buffer: after the parameters of its linear
            last = max((place[k] for k in (f"{owner}.weight", f"{owner}.bias") if k in place), default=None)
            if last is not None:
                return (last, 1, 0, 0, _SCALE_RANK[leaf], i)
        for new, old in kv_source.items():  # <attn>.k_proj.k_scale stood at <attn>.k_bmm_quantizer._amax
            if key.endswith("." + new) and key[: -len(new)] + old in place:
                return (place[key[: -len(new)] + old], 0, 0, 0, 0, i)
        return (len(place), 0, 0, 0, 0, i)  # anything else keeps its place at the end

    order = sorted(enumerate(state), key=rank)
    return {k: state[k] for _, k in order}


class ExportedState(dict):
    """export_state_dict's result: a plain dict of checkpoint tensors that also knows whether it holds this rank's
    shard only (`sharded`) -- save_checkpoint must not re-derive that from the process-wide declare_data_parallel state,
    which may have changed (or been overridden per call) since the dict was made."""

    def __init__(self, *args, sharded: bool = False, **kw):
        super().__init__(*args, **kw)
        self.sharded = sharded


def _tied_alias_keys(model, state: dict) -> list:
    """Tied weights are stored once (postprocess_state_dict's tied-weight dedup, export/quant_utils.py:1062-1110): a key
    the model DECLARES as an alias (`_tied_weights_keys`: alias -> canonical, transformers >= 5; a list of aliases
    before) is dropped when its canonical counterpart is in the state and the two really share memory."""
    tied = getattr(model, "_tied_weights_keys", None)
    if not tied:
        return []
    pairs = tied.items() if isinstance(tied, dict) else [(a, None) for a in tied]
    out = []
    for alias, canonical in pairs:
        if alias not in state:
            continue
        ptr = state[alias].data_ptr()
        if canonical is not None:
            same = canonical in state and state[canonical].data_ptr() == ptr
        else:
            same = any(k != alias and v.data_ptr() == ptr for k, v in state.items())
        if same:
            out.append(alias)
    return out


KV_CACHE_FP8 = "FP8"
# export/quant_utils.py:964-970: where the KV-cache quantizer buffers land in the checkpoint
_KV_CACHE_REPLACEMENTS = {"k_bmm_quantizer._amax": "k_proj.k_scale", "v_bmm_quantizer._amax": "v_proj.v_scale",
                          "k_bmm_quantizer._bias_value": "k_proj.k_bias", "v_bmm_quantizer._bias_value": "v_proj.v_bias"}


def get_kv_cache_format(model) -> str | None:
    """get_kv_cache_dtype / _compute_kv_cache_dtype (export/quant_utils.py:408-461) over the modules of the model, the way
    get_quant_config walks them (:1675-1690): a module counts when its k / v bmm quantizer OR ITS OUTPUT QUANTIZER is enabled
    (the output quantizers of k_proj / v_proj were the KV-cache quantizers of the Megatron export path, and the rule still
    reads any module's); "FP8" when one of them is E4M3, "INT8" for 8 bits, "NVFP4" / "NVFP4_AFFINE" for E2M1 (all with
    offsets), None otherwise; every counted module must agree.  What the checkpoint writer packs is FP8 (incl. the affine and
    cast-style presets): an INT8 KV cache stops it with the reference's assertion, NVFP4 is outside this path."""
    fmt = None
    for m in model.modules():
        live = [q for q in (getattr(m, name, None) for name in ("k_bmm_quantizer", "v_bmm_quantizer", "output_quantizer"))
                if q is not None and q.is_enabled]
        if not live:
            continue
        bits = [tuple(q.num_bits) if isinstance(q.num_bits, (tuple, list)) else q.num_bits for q in live]
        if (4, 3) in bits:
            this = KV_CACHE_FP8
        elif 8 in bits:
            this = "INT8"
        elif (2, 1) in bits:
            this = "NVFP4_AFFINE" if all(hasattr(q, "_bias_value") for q in live) else "NVFP4"
        else:
            this = None
        if fmt is None:
            fmt = this
        else:
            assert fmt == this, "Do not support mixed precision kv cache quantization"
    return fmt


def _postprocess_kv_key(key: str, value: torch.Tensor, kv_format: str | None):
    """The KV-cache part of postprocess_state_dict (export/quant_utils.py:1000-1060): `<attn>.k_bmm_quantizer._amax`
    becomes `<attn>.k_proj.k_scale = amax.float() / maxbound`; every other quantizer buffer is dropped; the rest is
    copied through."""
    if not any(s in key for s in ("_bmm_quantizer", "output_quantizer", "_amax", "_bias_value")):
        return key, value.detach()
    for old, new in _KV_CACHE_REPLACEMENTS.items():
        if key.endswith(old):
            if "_amax" in key:
                if kv_format in ("NVFP4", "NVFP4_AFFINE"):
                    raise NotImplementedError("NVFP4 KV-cache scales are outside this path (FP8 E4M3 only)")
                assert kv_format == KV_CACHE_FP8, "Invalid KV cache quantization format."  # (quant_utils.py:1038-1040)
                value = numerics.div_scalar(value, 448.0)  # (see get_scaling_factor)
            return key[: -len(old)] + new, value
    return None, None


def _summarize_excluded(unquantized, quantized) -> set:
    """_prefix_wildcard_summarize_exclude_modules (export/quant_utils.py:607-677): the unquantized module names as the
    SHORTEST prefix wildcards cut at a dot that match no quantized module -- `name*` first, then the pair {name, name.*}
    (emitted as `name.*`), finally the full name; a wildcard already emitted covers later layers."""
    forbidden = set()
    for q in quantized:
        forbidden.add(q)
        forbidden.update(q[:i] + "*" for i in range(len(q) + 1))
    picked: set = set()
    for layer in unquantized:
        options = []
        for i, ch in enumerate(layer):
            if ch == ".":
                options.append([layer[:i] + "*"])
                options.append([layer[:i], layer[:i] + ".*"])
        options.append([layer])
        choice = []
        for cand in options:
            if any(c in forbidden for c in cand):
                continue  # would swallow a quantized layer: be more specific
            if all(c in picked for c in cand):
                choice = []  # covered already
                break
            choice = cand
            break
        if len(choice) == 2:
            a, b = sorted(choice, key=len)
            picked.update([b] if b == a + ".*" else choice)
        else:
            picked.update(choice)
    return picked


_QUANT_ALGO = {QUANTIZATION_FP8: "FP8", QUANTIZATION_FP8_PC_PT: "FP8_PER_CHANNEL_PER_TOKEN", QUANTIZATION_INT8_WO: "W8A16",
               QUANTIZATION_INT8_SQ: "W8A8_SQ_PER_CHANNEL"}


def _layer_quant_config(fmt: str, block: int) -> dict:
    """One layer's entry of process_layer_quant_config (export/quant_utils.py:703-768)."""
    if fmt in (QUANTIZATION_INT4_AWQ, QUANTIZATION_W4A8_AWQ):
        return {"quant_algo": "W4A16_AWQ" if fmt == QUANTIZATION_INT4_AWQ else "W4A8_AWQ", "group_size": block,
                "has_zero_point": False, "pre_quant_scale": True}
    if fmt == QUANTIZATION_W4A8_MXFP4_FP8:
        return {"quant_algo": "W4A8_MXFP4_FP8", "group_size": block}
    if fmt == QUANTIZATION_MXFP8:
        return {"quant_algo": "MXFP8", "group_size": block}
    return {"quant_algo": _QUANT_ALGO.get(fmt, fmt)}  # (mxfp4, fp8_pb_wo: the format's own name)


def hf_quant_config(model, group_size: int | None = None) -> dict:
    """hf_quant_config.json content: get_quant_config + process_layer_quant_config (export/quant_utils.py:1583-1702,
    :680-790) and the renaming of module references to checkpoint names (unified_export_hf.py:1594-1610).  Every module
    that carries quantizers in the reference is listed -- quantized linears, fused expert containers, embeddings (the
    reference wraps them, disabled) and MoE routers (:1550-1580) --; one format over the model gives that format's entry
    plus `exclude_modules` (shortest prefix wildcards of the unquantized modules), several give MIXED_PRECISION with the
    per-layer table."""
    from torch import nn as _nn

    layers: dict = {}
    for name, m in model.named_modules():
        if is_quantized_linear(m):
            wq = m.weight_quantizer
            fmt = get_quantization_format(m)
            stage = wq[0] if isinstance(wq, SequentialQuantizer) else wq
            bsz = getattr(stage, "block_sizes", None) or {}
            block = bsz.get(-1, None) or bsz.get(m.weight.dim() - 1, None) or 0
            layers[name] = (fmt, int(block) if fmt is not None else 0)
        elif is_quant_fused_experts(m):
            # get_quantization_format on the container (quant_utils.py:485-601): per weight attribute the FIRST expert's
            # weight quantizer stands for all of them, next to the projection's shared input quantizer; the first
            # attribute with a format decides.  Block size: the first attribute whose quantizer has one (:1640-1650).
            fmt, block = None, 0
            for wqs_attr, iq_attr in ((m._first_proj_weight_quantizers_attr, m._first_proj_input_quantizer_attr),
                                      ("down_proj_weight_quantizers", "down_proj_input_quantizer")):
                wq = getattr(m, wqs_attr)[0]
                if fmt is None:
                    fmt = get_quantization_format(_ExpertProjection(None, wq, getattr(m, iq_attr)))
                stage = wq[0] if isinstance(wq, SequentialQuantizer) else wq
                bsz = getattr(stage, "block_sizes", None) or {}
                block = block or int(bsz.get(-1, None) or 0)
            layers[name] = (fmt, block if fmt is not None else 0)
        elif type(m) is _nn.Embedding:  # the reference's registry wraps the exact class only (OPT's learned positional
            layers[name] = (None, 0)  # embedding, a subclass, carries no quantizers there and is not listed)
    for name, m in model.named_modules():  # MoE routers kept in original precision (not nn.Linear under transformers >= 5)
        if not hasattr(m, "experts"):
            continue
        for attr in ("gate", "router", "shared_expert_gate"):
            r = getattr(m, attr, None)
            if isinstance(r, _nn.Module) and isinstance(getattr(r, "weight", None), torch.Tensor) and not is_quantized_linear(r):
                layers.setdefault(f"{name + '.' if name else ''}{attr}", (None, 0))
    quantized = {n: _layer_quant_config(f, group_size if (group_size and f in (QUANTIZATION_INT4_AWQ, QUANTIZATION_W4A8_AWQ)) else b)
                 for n, (f, b) in layers.items() if f is not None}
    excluded = [n for n, (f, _) in layers.items() if f is None]
    q: dict = {"quant_algo": None, "kv_cache_quant_algo": None}
    kinds = {json.dumps(v, sort_keys=True) for v in quantized.values()}
    rename = _module_name_mapper(model)  # (identity when the tensors keep the module tree's names too; warned about there)
    if len(kinds) > 1:
        q["quant_algo"] = "MIXED_PRECISION"
        q["quantized_layers"] = {rename(n): v for n, v in quantized.items()}
    elif len(kinds) == 1:
        q.update(next(iter(quantized.values())))
        q["exclude_modules"] = [rename(e) for e in sorted(_summarize_excluded(excluded, list(quantized)))]
    else:
        q["quantized_layers"] = {}
    q["kv_cache_quant_algo"] = get_kv_cache_format(model)
    return {"producer": {"name": "model_optimizer_amd", "version": "0.1"}, "quantization": q}


# weights / input_activations entries of a compressed-tensors style config group, per quant_algo
# (export/convert_hf_config.py:23-130, _quant_algo_to_group_config): (activation spec or None, weight spec, extras)
def _group_of(algo: str, group_size: int | None) -> dict:
    f8 = {"dynamic": False, "num_bits": 8, "type": "float"}
    table = {
        "FP8": (dict(f8), dict(f8), None),
        "FP8_PER_CHANNEL_PER_TOKEN": (dict(f8), dict(f8, strategy="channel"), None),
        "W4A16_AWQ": (None, {"dynamic": False, "num_bits": 4, "type": "int", "group_size": group_size or 128}, None),
        "W4A8_AWQ": (dict(f8, group_size=group_size or 128),
                     {"dynamic": False, "num_bits": 4, "type": "float", "group_size": group_size or 128}, None),
        "W8A16": (None, {"dynamic": False, "num_bits": 8, "type": "int"}, None),
        "W8A8_SQ_PER_CHANNEL": ({"dynamic": False, "num_bits": 8, "type": "int"},
                                {"dynamic": False, "num_bits": 8, "type": "int", "strategy": "channel"}, None),
        "W4A8_MXFP4_FP8": (dict(f8), {"dynamic": False, "num_bits": 4, "type": "float", "group_size": group_size or 16}, None),
        "MXFP8": (dict(f8, group_size=group_size or 32), dict(f8, group_size=group_size or 32), None),
    }
    if algo not in table:
        warnings.warn(f"Unsupported quantization algorithm '{algo}' in the config-group table: the group carries the name only")
        return {"quant_algo": algo}
    act, w, _ = table[algo]
    return {**({"input_activations": act} if act is not None else {}), "weights": w}


def convert_hf_quant_config_format(input_config: dict) -> dict:
    """export/convert_hf_config.py:133-278: the `quantization_config` embedded into config.json (the llm-compressor /
    compressed-tensors layout deployment frameworks read): config groups for FP8 and for MIXED_PRECISION (one group per
    distinct per-layer entry, targets = its layers), `ignore` = exclude_modules, the kv-cache scheme, the producer, and
    quant_method "modelopt"."""
    q = input_config.get("quantization", {})
    algo = q.get("quant_algo")
    out: dict = {}
    if algo == "FP8":
        out["config_groups"] = {"group_0": {**_group_of("FP8", None), "targets": ["Linear"]}}
    elif algo == "MIXED_PRECISION":
        layers = q.get("quantized_layers", {})
        by_cfg = defaultdict(list)
        for name, cfg in layers.items():
            by_cfg[tuple(sorted(cfg.items()))].append(name)
        groups = {}
        for idx, (key, names) in enumerate(by_cfg.items()):
            cfg = dict(key)
            groups[f"group_{idx}"] = {**_group_of(cfg.get("quant_algo", ""), cfg.get("group_size")), "targets": sorted(names)}
        out["config_groups"] = groups
        out["quantized_layers"] = layers
    excl = q.get("exclude_modules")
    out["ignore"] = excl if excl is not None else []
    if algo:
        out["quant_algo"] = algo
    kv = q.get("kv_cache_quant_algo")
    if kv:
        out["kv_cache_scheme"] = {"dynamic": False, "num_bits": 8, "type": "float"} if kv == "FP8" else kv
    if input_config.get("producer"):
        out["producer"] = input_config["producer"]
    out["quant_method"] = "modelopt"
    return out


def export_hf_checkpoint(model, dtype: torch.dtype | None = None, export_dir: str | None = None, dummy_forward_fn=None,
                         max_shard_size: int | str = "10GB") -> dict:
    """export_hf_checkpoint (export/unified_export_hf.py:1491-1650) for this path: the packed state dict under the
    checkpoint's tensor names, hf_quant_config.json, and -- for a Hugging Face model -- config.json (with the embedded
    `quantization_config`), generation_config.json and the (possibly sharded) safetensors through the model's own
    save_pretrained, like the reference.  Other modules get model.safetensors + hf_quant_config.json (save_checkpoint).
    `dummy_forward_fn`: the probe forward of the resmooth / fusion step (default for a causal LM: two token ids).
    Returns the hf_quant_config dict."""
    import tempfile

    export_dir = export_dir or tempfile.gettempdir()
    os.makedirs(export_dir, exist_ok=True)
    param = next(model.parameters())
    dtype = dtype or param.dtype
    is_hf = hasattr(model, "save_pretrained") and hasattr(model, "config")
    if dummy_forward_fn is None and is_hf:
        dummy_forward_fn = lambda: model(torch.ones([1, 2], dtype=torch.long, device=param.device))  # noqa: E731
    state = export_state_dict(model, dtype, dummy_forward_fn, shard_weights=False)
    quant = hf_quant_config(model)
    qd = quant["quantization"]
    quantized = qd.get("quant_algo") is not None or qd.get("kv_cache_quant_algo") is not None
    if not is_hf:
        save_checkpoint(state, export_dir, quant if quantized else None, shard_weights=False)
        return quant
    tensors = {k: v.detach().contiguous() for k, v in state.items()}
    # the shard index counts the MODEL's parameters (save_pretrained: num_parameters()); the reference's exporter has put the
    # packed tensors into its modules by then (INT4: half the elements), this one leaves the model as it was.  (Counted now:
    # save_pretrained empties the dict it is given while it writes the shards.)
    rules = [] if _keeps_module_names(model) else _checkpoint_rename_rules(model)
    total = sum((tensors[k] if (k := _rename_key(n, rules)) in tensors else p).numel() for n, p in model.named_parameters())
    patched = []
    try:
        # transformers >= 5 would apply the reverse of its load-time key conversion to the state dict it is given; the keys
        # are the checkpoint's already (rename_to_checkpoint_keys) and its converter cannot walk 0-dim scale tensors
        import importlib

        for mod_name in ("transformers.core_model_loading", "transformers.modeling_utils"):
            try:
                mod = importlib.import_module(mod_name)
            except Exception:  # noqa: BLE001
                continue
            if hasattr(mod, "revert_weight_conversion"):
                patched.append((mod, mod.revert_weight_conversion))
                mod.revert_weight_conversion = lambda model_, state_dict: state_dict
        model.save_pretrained(export_dir, state_dict=tensors, max_shard_size=max_shard_size)
    finally:
        for mod, fn in patched:
            mod.revert_weight_conversion = fn
    index_path = os.path.join(export_dir, "model.safetensors.index.json")
    if os.path.exists(index_path):
        with open(index_path) as f:
            index = json.load(f)
        if index.get("metadata", {}).get("total_parameters") != total:
            index["metadata"]["total_parameters"] = total
            with open(index_path, "w") as f:
                f.write(json.dumps(index, indent=2, sort_keys=True) + "\n")
    if quantized:
        with open(os.path.join(export_dir, "hf_quant_config.json"), "w") as f:
            json.dump(quant, f, indent=4)
    cfg_path = os.path.join(export_dir, "config.json")
    with open(cfg_path) as f:
        cfg = json.load(f)
    if quantized:
        cfg["quantization_config"] = convert_hf_quant_config_format(quant)
    with open(cfg_path, "w") as f:
        json.dump(cfg, f, indent=4)
    return quant


def save_checkpoint(state: dict, export_dir: str, quant_config: dict | None = None, shard_weights: bool | None = None):
    """model.safetensors (+ hf_quant_config.json); tensors are written sorted by key like safetensors does.

    shard_weights (with export_state_dict(shard_weights=...)): every rank writes the tensors it packed as
    model-<rank+1>-of-<world>.safetensors -- device -> host copy and file write run on all ranks at once, no tensor
    crosses xGMI -- and rank 0 adds the Hugging Face shard index (model.safetensors.index.json: key -> file) from one
    object all-gather of the key lists."""
    from safetensors.torch import save_file

    from . import distributed as mdist

    os.makedirs(export_dir, exist_ok=True)
    produced = getattr(state, "sharded", None)  # ExportedState: how export_state_dict made it
    if produced is not None:
        if shard_weights is not None and bool(shard_weights) != produced:
            raise ValueError(f"save_checkpoint(shard_weights={shard_weights}) but the state dict was exported with "
                             f"shard_weights={produced}: a full dict written as shards collides on every key, a shard "
                             "written as model.safetensors loses the other ranks' tensors")
        sharded = produced and mdist.active(mdist.replica_group())
    else:  # a plain dict (hand-made): the caller's word, else the declared replicas
        sharded = mdist.resolve_shard(shard_weights)
    if sharded:
        import torch.distributed as dist

        group = mdist.replica_group()
        world, me = dist.get_world_size(group), dist.get_rank(group)
        fname = f"model-{me + 1:05d}-of-{world:05d}.safetensors"
        tensors = {k: v.detach().cpu().contiguous() for k, v in state.items()}
        save_file(tensors, os.path.join(export_dir, fname), metadata={"format": "pt"})
        mine = (fname, {k: t.numel() * t.element_size() for k, t in tensors.items()})
        everyone = [None] * world
        dist.all_gather_object(everyone, mine, group=group)
        if me == 0:
            weight_map, total = {}, 0
            for f, sizes in everyone:
                for k, n in sizes.items():
                    if k in weight_map:
                        raise RuntimeError(f"sharded export: {k} was packed by two ranks")
                    weight_map[k] = f
                    total += n
            with open(os.path.join(export_dir, "model.safetensors.index.json"), "w") as f:
                json.dump({"metadata": {"total_size": total}, "weight_map": dict(sorted(weight_map.items()))}, f, indent=2)
            if quant_config is not None:
                with open(os.path.join(export_dir, "hf_quant_config.json"), "w") as f:
                    json.dump(quant_config, f, indent=4)
        dist.barrier(group=group)
        return
    save_file({k: v.detach().cpu().contiguous() for k, v in state.items()},
              os.path.join(export_dir, "model.safetensors"), metadata={"format": "pt"})
    if quant_config is not None:
        with open(os.path.join(export_dir, "hf_quant_config.json"), "w") as f:
            json.dump(quant_config, f, indent=4)
