"""QuantLinear -- mirror of the reference's _QuantLinear (nn/modules/quant_linear.py:38-53,
quant_module.py:227-278): an nn.Linear with input / weight / output TensorQuantizers."""

from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn

from .tensor_quantizer import QuantizerAttributeConfig, TensorQuantizer


class QuantLinear(nn.Linear):
    """forward = output_quantizer(F.linear(input_quantizer(x), weight_quantizer(W), b))."""

    default_quant_desc_input = QuantizerAttributeConfig(num_bits=8, axis=None)
    default_quant_desc_weight = QuantizerAttributeConfig(num_bits=8, axis=0)

    def _setup(self):
        # (input, output, weight: the reference's registration order -- the order of named_modules(), hence of the lines of
        # print_quant_summary and of the quantizer buffers in state_dict())
        self.input_quantizer = TensorQuantizer(self.default_quant_desc_input)
        self.output_quantizer = TensorQuantizer(QuantizerAttributeConfig(enable=False))
        self.weight_quantizer = TensorQuantizer(self.default_quant_desc_weight)

    @classmethod
    def convert(cls, linear: nn.Linear) -> "QuantLinear":
        """In-place class swap (the reference's DynamicModule conversion keeps parameters in place)."""
        linear.__class__ = cls
        linear._setup()
        return linear

    def forward(self, input):
        x = self.input_quantizer(input)
        wq, oq, w = self.weight_quantizer, self.output_quantizer, self.weight
        if not (type(wq) is TensorQuantizer and wq.weight_already_counted(w)):
            w = wq(w)
        y = F.linear(x, w, self.bias)
        return y if type(oq) is TensorQuantizer and oq.hands_back() else oq(y)


class QuantLayerNorm(nn.LayerNorm):
    """nn.LayerNorm with an input (and output) quantizer -- the reference registers nn.LayerNorm with QuantInputBase
    (nn/modules/quant_layernorm.py:28, quant_module.py:192-240), so the `*input_quantizer` entries of the presets also
    fake-quantize what goes INTO every LayerNorm of OPT / GPT-style models.  RMSNorm classes of the Llama family are
    model code, not nn.LayerNorm, and stay untouched on both sides."""

    def _setup(self):
        self.input_quantizer = TensorQuantizer(QuantLinear.default_quant_desc_input)
        self.output_quantizer = TensorQuantizer(QuantizerAttributeConfig(enable=False))

    @classmethod
    def convert(cls, norm: nn.LayerNorm) -> "QuantLayerNorm":
        norm.__class__ = cls
        norm._setup()
        return norm

    def forward(self, input):
        return self.output_quantizer(super().forward(self.input_quantizer(input)))


def is_quantized_linear(m) -> bool:
    return isinstance(m, QuantLinear)


def replace_quant_module(model: nn.Module) -> nn.Module:
    """nn.Linear -> QuantLinear everywhere (conversion.py:214 replace_quant_module for the Linear entry); attention
    modules of a Hugging Face model get their KV-cache quantizers (plugins/huggingface.py:371-415, run by the
    reference from the same place through its on-the-fly plugin registry); fused 3-D expert containers get per-expert
    weight quantizers (:1697-1730)."""
    from .hf_attention import register_hf_attentions_on_the_fly
    from .hf_experts import register_fused_experts_on_the_fly

    register_hf_attentions_on_the_fly(model)
    register_fused_experts_on_the_fly(model)
    for mod in list(model.modules()):
        # nn.Linear itself, and FalconLinear (an nn.Linear subclass computing input @ W.T + b), which the reference
        # registers explicitly (plugins/huggingface.py:1417-1420, :1574-1590); other subclasses are left alone
        if type(mod) is nn.Linear or (type(mod).__name__ == "FalconLinear" and isinstance(mod, nn.Linear)):
            QuantLinear.convert(mod)
        elif type(mod) is nn.LayerNorm:
            QuantLayerNorm.convert(mod)
        elif type(mod).__name__ == "Conv1D" and type(mod).__module__.startswith("transformers."):
            # transformers' Conv1D (GPT-2) is a linear layer with a transposed [in, out] weight and addmm: the weight
            # is transposed once and the module becomes a quantized nn.Linear (plugins/huggingface.py:559-571)
            with torch.no_grad():
                mod.weight = nn.Parameter(mod.weight.T.contiguous())
            mod.out_features, mod.in_features = mod.weight.shape
            QuantLinear.convert(mod)
    return model
