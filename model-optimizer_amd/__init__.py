"""model_optimizer_amd -- MI355X-native PTQ calibration / quantize-dequantize engine.

Python host (torch for memory/streams/distributed) over hand-written gfx950 HIP kernels behind the C-ABI of
libmoquant.so (include/moquant.h).  Mirrors the reference's plugin surface for this path:
  ops            -- functional ops (reduce_amax, fake_tensor_quant, scaled_e4m3, dynamic_block_quant, ...)
  multi_tensor   -- whole-model weight passes through a segment table
No CPU fallback exists: every op raises if the HIP library is missing or the tensor is not on the GPU.
"""

from . import _lib  # noqa: F401
from ._lib import MoquantError, MoquantUnsupported  # noqa: F401
from . import ops  # noqa: F401
from . import multi_tensor  # noqa: F401

__all__ = ["ops", "multi_tensor", "MoquantError", "MoquantUnsupported"]
