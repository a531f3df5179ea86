"""Fused MoE expert containers of Hugging Face models (transformers >= 5: `gate_up_proj` [E, 2I, H] and `down_proj`
[E, H, I] as 3-D parameters, e.g. MixtralExperts) -- the mirror of `_QuantFusedExperts` /
`_QuantNonGatedFusedExperts` and `register_fused_experts_on_the_fly` (quantization/plugins/huggingface.py:976-1142,
:1650-1730) for this path.

Every expert gets its own weight quantizers (`<proj>_weight_quantizers`, an nn.ModuleList), the input quantizers are
shared by all experts (`<proj>_input_quantizer`) -- the layout downstream runtimes expect.  The container's forward
calls `F.linear` exactly twice per expert (first projection, then down projection); during that forward `F.linear`
is replaced by a function that runs the quantizers around it and recovers the expert index from the weight slice's
storage offset, and is restored afterwards.  Weight statistics never depend on routing: `iter_weights_for_calibration`
hands every expert's slice to `weight_only_quantize`, which puts all of them into the one multi-tensor abs-max launch
(the slices are contiguous views into the 3-D parameter).
"""

from __future__ import annotations

import torch
from torch import nn

from .tensor_quantizer import QuantizerAttributeConfig, TensorQuantizer

_quant_classes: dict[type, type] = {}


def _first_proj_attr_of(module: nn.Module) -> str | None:
    """huggingface.py:1650-1686: gated containers fuse gate+up into `gate_up_proj`; non-gated ones (NemotronH) have a
    single 3-D `up_proj`.  Both need a 3-D `down_proj`, `num_experts` and an activation."""
    down = getattr(module, "down_proj", None)
    if not (isinstance(down, nn.Parameter) and down.dim() == 3 and hasattr(module, "num_experts")
            and hasattr(module, "act_fn")):
        return None
    for attr in ("gate_up_proj", "up_proj"):
        p = getattr(module, attr, None)
        if isinstance(p, nn.Parameter) and p.dim() == 3:
            return attr
    return None


class _QuantFusedExpertsMixin:
    @property
    def _first_proj_input_quantizer_attr(self) -> str:
        return f"{self._first_proj_attr}_input_quantizer"

    @property
    def _first_proj_weight_quantizers_attr(self) -> str:
        return f"{self._first_proj_attr}_weight_quantizers"

    @property
    def _is_gated(self) -> bool:
        return self._first_proj_attr == "gate_up_proj"

    def _setup_expert_quantizers(self):
        n = self.num_experts
        default_in = QuantizerAttributeConfig(num_bits=8, axis=None)
        default_w = QuantizerAttributeConfig(num_bits=8, axis=None)
        setattr(self, self._first_proj_input_quantizer_attr, TensorQuantizer(default_in))
        setattr(self, self._first_proj_weight_quantizers_attr,
                nn.ModuleList([TensorQuantizer(default_w) for _ in range(n)]))
        self.down_proj_input_quantizer = TensorQuantizer(default_in)
        self.down_proj_weight_quantizers = nn.ModuleList([TensorQuantizer(default_w) for _ in range(n)])

    def _expert_idx_from_first_proj(self, weight: torch.Tensor) -> int:
        """huggingface.py:1019-1040: `<first_proj>[idx]` is a view into the 3-D parameter."""
        first = getattr(self, self._first_proj_attr)
        stride = first.stride(0)
        if stride == 0:
            return 0
        idx = (weight.storage_offset() - first.storage_offset()) // stride
        assert 0 <= idx < self.num_experts, (
            f"Computed expert index {idx} out of range [0, {self.num_experts}). "
            "This can happen if the weight was .contiguous()-copied or redistributed.")
        return idx

    def forward(self, *args, **kwargs):
        functional = torch.nn.functional
        original = functional.linear
        state = {"down": False, "idx": 0}

        def quantized_linear(input, weight, bias=None):
            # huggingface.py:1060-1075: strict alternation first projection / down projection per expert
            if state["down"]:
                idx = state["idx"]
                input = self.down_proj_input_quantizer(input)
                weight = self.down_proj_weight_quantizers[idx](weight)
            else:
                idx = state["idx"] = self._expert_idx_from_first_proj(weight)
                input = getattr(self, self._first_proj_input_quantizer_attr)(input)
                weight = getattr(self, self._first_proj_weight_quantizers_attr)[idx](weight)
            state["down"] = not state["down"]
            return original(input, weight, bias)

        functional.linear = quantized_linear
        try:
            return super().forward(*args, **kwargs)
        finally:
            functional.linear = original

    def iter_weights_for_calibration(self):
        """(weight slice, quantizer) per expert and projection (huggingface.py:1084-1100)."""
        for weight_name, quantizers_name in ((self._first_proj_attr, self._first_proj_weight_quantizers_attr),
                                             ("down_proj", "down_proj_weight_quantizers")):
            weight = getattr(self, weight_name)
            for idx, q in enumerate(getattr(self, quantizers_name)):
                yield weight[idx], q

    def iter_projections(self):
        """(weight name, weight quantizers, shared input quantizer) of the two projections."""
        yield (self._first_proj_attr, getattr(self, self._first_proj_weight_quantizers_attr),
               getattr(self, self._first_proj_input_quantizer_attr))
        yield "down_proj", self.down_proj_weight_quantizers, self.down_proj_input_quantizer


def is_quant_fused_experts(m) -> bool:
    return isinstance(m, _QuantFusedExpertsMixin)


def convert_fused_experts(module: nn.Module) -> nn.Module:
    cls = type(module)
    if issubclass(cls, _QuantFusedExpertsMixin):
        return module
    first = _first_proj_attr_of(module)
    if first is None:
        raise TypeError(f"{cls.__name__} is not a fused-experts container (3-D gate_up_proj / up_proj + down_proj)")
    key = (cls, first)
    qcls = _quant_classes.get(key)
    if qcls is None:
        qcls = type(f"Quant{cls.__name__}", (_QuantFusedExpertsMixin, cls), {"_first_proj_attr": first,
                                                                           "_moq_original_cls": cls})
        _quant_classes[key] = qcls
    module.__class__ = qcls
    module._setup_expert_quantizers()
    return module


def force_eager_experts_impl(model: nn.Module) -> None:
    """huggingface.py:1725-1754: transformers >= 5 dispatches the container's forward on
    `config._experts_implementation`; the `grouped_mm` / `batched_mm` backends bypass F.linear and with it every
    quantizer -- calibration would silently collect nothing.  Any model with a fused-experts container is therefore
    switched to the eager (F.linear) forward, on the model config and its text / vision / audio sub-configs."""
    def force(cfg):
        if cfg is None:
            return
        if hasattr(cfg, "_experts_implementation"):
            cfg._experts_implementation = "eager"
        for sub in ("text_config", "vision_config", "audio_config", "speech_config"):
            if hasattr(cfg, sub):
                force(getattr(cfg, sub))

    force(getattr(model, "config", None))
    for m in model.modules():
        if _first_proj_attr_of(m) is not None:
            force(getattr(m, "config", None))


# Expert containers the reference quantizes with classes of their own (transposed [E, H, 2I] layouts with biases, one
# quantizer over the stacked 3-D tensor, DBRX's GLU: plugins/huggingface.py:753-774, :776-875, :877-963, :1467-1557), registered
# there before the generic rule is tried.  The generic per-expert rule below would accept some of them and quantize them
# DIFFERENTLY (other quantizer granularity, other checkpoint layout), so they are refused by name instead.
_OWN_CLASS_IN_THE_REFERENCE = ("Llama4TextExperts", "GptOssExperts", "DbrxExperts", "DbrxExpertGLU", "Qwen3VLMoeTextExperts")


def register_fused_experts_on_the_fly(model: nn.Module) -> int:
    """Convert every fused-experts container of the model (huggingface.py:1697-1722) and put the model on the eager
    experts forward."""
    from ._lib import MoquantUnsupported

    special = sorted({type(m).__name__ for m in model.modules() if type(m).__name__ in _OWN_CLASS_IN_THE_REFERENCE})
    if special:
        raise MoquantUnsupported(f"expert containers {special}: the reference quantizes these with container classes of its own "
                                 "(plugins/huggingface.py:753-963, :1467-1557), which are outside this path")
    found = [m for m in model.modules() if not is_quant_fused_experts(m) and _first_proj_attr_of(m) is not None]
    if not found:
        return 0
    force_eager_experts_impl(model)
    for m in found:
        convert_fused_experts(m)
    return len(found)
