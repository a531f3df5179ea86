"""ctypes binding of libmoquant.so (C-ABI declared in include/moquant.h).

There is NO fallback: if the HIP library is missing or a call fails, we raise.  (The CPU oracle under
oracle/ is test infrastructure and is never reachable from here.)
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_uint8, c_ulonglong, c_void_p  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# MOQ_LIB_PATH: A/B a differently built libmoquant.so (e.g. `MOQ_EXPERIMENTS=1 csrc/build.sh` -> libmoquant_exp.so with the
# MOQ_TUNE_* knobs live); the default is the in-tree release build
LIB_PATH = os.environ.get("MOQ_LIB_PATH") or os.path.join(_HERE, "csrc", "libmoquant.so")

MOQ_OK, MOQ_ERR_INVALID, MOQ_ERR_UNSUPPORTED, MOQ_ERR_LAUNCH = 0, -1, -2, -3
F32, F16, BF16 = 0, 1, 2
AMAX_SCALAR, AMAX_AXIS = 0, 1
ROUND_HALF_EVEN, ROUND_HALF_AWAY = 0, 1
MT_CHUNK = 8192
MX_TYPES = {"E4M3": 0, "E5M2": 1, "INT8": 2, "E0M3": 3, "E1M2": 4, "E3M0": 5, "E2M1": 6, "E3M2": 7,
            "E2M3": 8, "E8M0": 9}


class MoqSeg(ctypes.Structure):
    """struct moq_seg (include/moquant.h)."""

    _fields_ = [("x", c_void_p), ("y", c_void_p), ("amax", c_void_p), ("n", c_int64)]


# name -> (restype, argtypes); must list every symbol include/moquant.h declares
SIGNATURES = {
    "moq_abi_version": (c_int, []),
    "moq_last_error": (c_char_p, []),
    "moq_device_cu_count": (c_int, []),
    "moq_amax": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "moq_amax_axis": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int, c_void_p]),
    "moq_fake_quant_int": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_int64,
                                   c_int, c_int, c_int, c_void_p]),
    "moq_amax_qdq_int_group": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int,
                                       c_int, c_void_p]),
    "moq_fake_quant_e4m3": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int, c_int64, c_int64,
                                    c_void_p]),
    "moq_mt_plan": (c_int64, [POINTER(c_int64), c_int, POINTER(c_int64)]),
    "moq_mt_amax": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "moq_fp8_pack_tile": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "moq_fp8_unpack_tile": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_void_p]),
    "moq_amax_mid": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "moq_mx_convert": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "moq_row_hist_np": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "moq_hist_entropy": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_void_p, c_void_p]),
    "moq_hist_percentile": (c_int, [c_void_p, c_int, c_int64, c_int64, c_double, c_void_p, c_void_p]),
    "moq_mt_amax_ws": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p]),
    "moq_mt_amax_running": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "moq_mt_fake_quant_e4m3": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "moq_mt_fake_quant_int": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int,
                                      c_void_p]),
    "moq_mt_amax_qdq_int_group": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_int,
                                          c_int, c_void_p]),
    "moq_mt_mask_2to4": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p]),
    "moq_mt_mask_2to4_apply": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_void_p, c_void_p]),
    "moq_mt_mx_fused_amax_convert": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p]),
    "moq_mt_fold_mx_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_int, c_void_p]),
    "moq_mt_fold_mxfp4_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int, c_int, c_void_p]),
    "moq_mx_fused_amax_convert": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int,
                                          c_void_p, c_void_p]),
    "moq_col_abs_mean_accum": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "moq_input_quant": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, c_int,
                                c_int, c_int, c_void_p, c_int, c_float, c_int, c_void_p]),
    "moq_int8_pack_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "moq_hist_abs": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_int, c_float, c_int, c_void_p]),
    "moq_mask_2to4": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "moq_int4_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_void_p]),
    "moq_int4_unpack": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "moq_int4_pack_export": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int,
                                     c_void_p]),
    "moq_scale_cols": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "moq_awq_scale_qdq": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                  c_void_p]),
    "moq_awq_weight_scale": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "moq_col_stats_workspace": (c_int64, [c_int64, c_int64]),
    "moq_col_abs_stats": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int,
                                  c_void_p]),
    "moq_awq_err_gemm_workspace": (c_int64, [c_int64, c_int64]),
    "moq_awq_err_gemm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int,
                                 c_void_p, c_void_p, c_void_p]),
    "moq_awq_err_gemm_multi": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int,
                                       c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "moq_rescale_cols": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_void_p]),
    "moq_scale_cols_multi": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "moq_mse_sweep_workspace": (c_int64, [c_int64, c_int64, c_int64, c_int]),
    "moq_mse_sweep": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_int, c_void_p, c_void_p,
                              c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "moq_awq_clip_loss": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_int,
                                  c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "moq_fp8_pack": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_void_p]),
    "moq_fp8_unpack": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int64, c_int64, c_void_p]),
    "moq_mxfp4_pack": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "moq_mxfp4_unpack": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "moq_transpose16": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "moq_transpose16_ld": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "moq_hessian_accum": (c_int, [c_void_p, c_int64, c_int64, c_int, c_void_p, c_float, c_float, c_int, c_void_p]),
    "moq_symmetrize": (c_int, [c_void_p, c_int64, c_void_p]),
    "moq_sgpt_block_sweep": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int, c_int,
                                     c_void_p]),
    "moq_awq_err_weight": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int,
                                   c_int, c_void_p]),
    "moq_awq_quadform": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p,
                                 c_double, c_void_p]),
    "moq_block2d": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_int,
                            c_int, c_int, c_int, c_void_p]),
    "moq_sgpt_trailing_update": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "moq_gptq_block_sweep": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int64,
                                     c_int, c_int, c_int, c_int, c_void_p]),
    "moq_gemm_nt": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int, c_void_p]),
}

_lib = None


class MoquantError(RuntimeError):
    """A C-ABI call returned MOQ_ERR_INVALID / MOQ_ERR_LAUNCH (reference: TORCH_CHECK -> RuntimeError)."""


class MoquantUnsupported(ValueError):
    """MOQ_ERR_UNSUPPORTED.  ValueError on purpose: the reference's callers treat ValueError from the
    extension as 'layout not supported by the kernel' (quantization/tensor_quant.py:386-389)."""


def lib() -> ctypes.CDLL:
    """Load libmoquant.so (once).  Raises if it has not been built -- no silent fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MoquantError(
                f"{LIB_PATH} not found: build it with model-optimizer_amd/csrc/build.sh "
                "(or __graft_entry__.build()); there is no CPU/eager fallback")
        # torch must be imported first so that our DT_NEEDED libamdhip64.so.7 resolves to the HIP runtime
        # torch already loaded (one runtime per process: streams and pointers are shared with torch)
        import torch  # noqa: F401

        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        if handle.moq_abi_version() != 1:
            raise MoquantError("libmoquant ABI version mismatch")
        _lib = handle
    return _lib


def check(rc: int) -> None:
    if rc == MOQ_OK:
        return
    msg = lib().moq_last_error().decode(errors="replace")
    if rc == MOQ_ERR_UNSUPPORTED:
        raise MoquantUnsupported(msg)
    raise MoquantError(f"libmoquant error {rc}: {msg}")
