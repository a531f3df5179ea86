"""quantize() -- the mtq.quantize(model, config, forward_loop) entry point for this path
(quantization/model_quant.py:147-250): convert Linears, apply wildcard quantizer configs, calibrate."""

from __future__ import annotations

import fnmatch

from torch import nn

from . import model_calib
from .nn import replace_quant_module
from .tensor_quantizer import QuantizerAttributeConfig, TensorQuantizer

# presets mirroring modelopt_recipes/configs/ptq/presets/model/{int8,fp8,int4_awq,mxfp4,int8_smoothquant}.yaml
INT8_DEFAULT_CFG = {"quant_cfg": {"*weight_quantizer": {"num_bits": 8, "axis": 0},
                                  "*input_quantizer": {"num_bits": 8, "axis": None},
                                  "*lm_head*": {"enable": False}}, "algorithm": "max"}
FP8_DEFAULT_CFG = {"quant_cfg": {"*weight_quantizer": {"num_bits": (4, 3), "axis": None},
                                 "*input_quantizer": {"num_bits": (4, 3), "axis": None},
                                 "*lm_head*": {"enable": False}}, "algorithm": "max"}
INT4_AWQ_CFG = {"quant_cfg": {"*weight_quantizer": {"num_bits": 4, "block_sizes": {-1: 128, "type": "static"}},
                              "*input_quantizer": {"enable": False},
                              "*lm_head*": {"enable": False}},
                "algorithm": {"method": "awq_lite", "alpha_step": 0.1}}
# presets/model/int4_blockwise_weight_only.yaml (numerics/int4_per_block.yaml: num_bits 4, block_sizes {-1: 128})
INT4_BLOCKWISE_WEIGHT_ONLY_CFG = {"quant_cfg": {"*weight_quantizer": {"num_bits": 4, "block_sizes": {-1: 128, "type": "static"}},
                                                "*input_quantizer": {"enable": False},
                                                "*lm_head*": {"enable": False}}, "algorithm": "max"}
MXFP4_DEFAULT_CFG = {"quant_cfg": {"*weight_quantizer": {"num_bits": (2, 1), "block_sizes": {-1: 32, "type": "dynamic", "scale_bits": (8, 0)}},
                                   "*input_quantizer": {"num_bits": (2, 1), "block_sizes": {-1: 32, "type": "dynamic", "scale_bits": (8, 0)}},
                                   "*lm_head*": {"enable": False}}, "algorithm": None}
INT8_SMOOTHQUANT_CFG = {"quant_cfg": {"*weight_quantizer": {"num_bits": 8, "axis": 0},
                                      "*input_quantizer": {"num_bits": 8, "axis": None},
                                      "*lm_head*": {"enable": False}},
                        "algorithm": {"method": "smoothquant", "alpha": 1.0}}


def set_quantizer_by_cfg(model: nn.Module, quant_cfg: dict):
    """conversion.py:245 set_quantizer_by_cfg: later wildcard entries override earlier ones."""
    for name, mod in model.named_modules():
        if not isinstance(mod, TensorQuantizer):
            continue
        for pattern, attrs in quant_cfg.items():
            if not fnmatch.fnmatch(name, pattern):
                continue
            attrs = dict(attrs)
            enable = attrs.pop("enable", True)
            if attrs:
                mod.set_from_attribute_config(QuantizerAttributeConfig(**{"narrow_range": False, **attrs,
                                                                          "enable": enable}))
            if enable:
                mod.enable()
            else:
                mod.disable()


def quantize(model: nn.Module, config: dict, forward_loop=None) -> nn.Module:
    replace_quant_module(model)
    set_quantizer_by_cfg(model, config["quant_cfg"])
    algo = config.get("algorithm", "max")
    method, kwargs = (algo, {}) if not isinstance(algo, dict) else (algo["method"], {k: v for k, v in algo.items() if k != "method"})
    if method is None:
        return model
    if method == "max":
        model_calib.max_calibrate(model, forward_loop)
    elif method == "mse":
        model_calib.mse_calibrate(model, forward_loop, **kwargs)
    elif method == "smoothquant":
        model_calib.smoothquant(model, forward_loop, **kwargs)
    elif method in ("awq_lite", "awq_clip", "awq_full"):
        model_calib.awq(model, forward_loop, algorithm=method, **kwargs)
    else:
        raise ValueError(f"algorithm {method!r} is outside this path")
    return model
