"""model_optimizer_amd -- MI355X-native PTQ calibration / quantize-dequantize engine.

Python host (torch for memory/streams/distributed) over hand-written gfx950 HIP kernels behind the C-ABI of
libmoquant.so (include/moquant.h).  Mirrors the reference's plugin surface for this path:
  ops              functional ops (reduce_amax, fake_tensor_quant, scaled_e4m3, dynamic_block_quant, ...)
  multi_tensor     whole-model weight passes through a segment table (one launch per pass)
  calib            Max / Histogram / Mse calibrators with device-resident state
  tensor_quantizer TensorQuantizer (same attributes and life cycle as the reference's)
  nn, model_quant, model_calib   QuantLinear, quantize(), max_calibrate / smoothquant / awq_lite
  hf_attention     q / k / v bmm quantizers (FP8 KV cache) on Hugging Face attention modules
  hf_experts       per-expert weight quantizers on fused 3-D MoE expert containers (Mixtral, Qwen-MoE, ...)
  sparsity         create_asp_mask (2:4 magnitude), SparseGPT
  gptq             GPTQ weight update (Hessian on the matrix cores, one kernel per column block)
  qtensor          INT4QTensor / FP8QTensor / MXFP4QTensor real quantisation (pack / unpack kernels)
  layerwise        layer-by-layer calibration with checkpoint / resume
  export           resmooth / layernorm fusion / INT4 nibble packing of the checkpoint export (byte-identical)
  distributed      bucketed all-reduce of amax / histograms / AWQ statistics (RCCL via torch.distributed)
  library_ops      torch.library operators (moquant::quantize_op, moquant::dynamic_block_quantize_op) with fake impls
  modelopt_plugin  install() -- the seams into an unmodified modelopt checkout
No CPU fallback exists: every op raises if the HIP library is missing or a tensor is not on the GPU.
"""

from . import _lib  # noqa: F401
from . import numerics  # noqa: F401
from ._lib import MoquantError, MoquantUnsupported  # noqa: F401
from . import ops  # noqa: F401
from . import multi_tensor  # noqa: F401
from . import calib  # noqa: F401
from . import tensor_quantizer  # noqa: F401
from . import nn  # noqa: F401
from . import hf_attention  # noqa: F401
from . import hf_experts  # noqa: F401
from . import hf_moe  # noqa: F401
from . import distributed  # noqa: F401
from . import model_calib  # noqa: F401
from . import model_quant  # noqa: F401
from . import sparsity  # noqa: F401
from . import gptq  # noqa: F401
from . import export  # noqa: F401
from . import qtensor  # noqa: F401
from . import layerwise  # noqa: F401
from . import forward_loop  # noqa: F401
from . import library_ops  # noqa: F401
from . import modelopt_plugin  # noqa: F401
from .model_quant import (calibrate, disable_quantizer, enable_quantizer, fold_weight, postprocess_amax,  # noqa: F401
                          print_quant_summary, quantize, set_quantizer_attribute, set_quantizer_attributes_full,
                          set_quantizer_attributes_partial, set_quantizer_by_cfg, set_quantizer_by_cfg_context)
from .tensor_quantizer import QuantizerAttributeConfig, TensorQuantizer  # noqa: F401

__all__ = ["ops", "multi_tensor", "calib", "tensor_quantizer", "nn", "hf_attention", "hf_experts", "distributed", "model_calib", "model_quant",
           "sparsity", "gptq", "export", "qtensor", "layerwise", "modelopt_plugin", "quantize", "calibrate", "fold_weight", "postprocess_amax", "disable_quantizer", "enable_quantizer",
           "print_quant_summary", "set_quantizer_by_cfg", "set_quantizer_by_cfg_context", "set_quantizer_attributes_partial",
           "set_quantizer_attributes_full", "set_quantizer_attribute", "TensorQuantizer", "QuantizerAttributeConfig",
           "MoquantError", "MoquantUnsupported"]
