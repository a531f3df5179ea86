"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/moquant.h declares, validates arguments before touching the GPU, and the Python host refuses
CPU tensors loudly (no fallback).  No kernel is launched here."""

import ctypes
import os
import re

import pytest
import torch

import _moa_import
from conftest import ROOT

moa = _moa_import.load()
from model_optimizer_amd import _lib  # noqa: E402


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "moquant.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(moq_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in moquant.h but not exported by libmoquant.so"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes SIGNATURES out of sync with moquant.h"
    assert _lib.lib().moq_abi_version() == 1


def test_argument_validation_without_gpu():
    lib = _lib.lib()
    # NULL output pointer -> MOQ_ERR_INVALID and a message, before any HIP call
    assert lib.moq_amax(None, 16, _lib.BF16, None, 0, None) == _lib.MOQ_ERR_INVALID
    assert b"moq_amax" in lib.moq_last_error()
    assert lib.moq_mask_2to4(ctypes.c_void_p(16), 2, 6, _lib.BF16, ctypes.c_void_p(16), None) == _lib.MOQ_ERR_UNSUPPORTED
    assert lib.moq_int4_pack(ctypes.c_void_p(16), ctypes.c_void_p(16), ctypes.c_void_p(16), 130, 128, 2, 0, None) \
        == _lib.MOQ_ERR_INVALID
    assert lib.moq_mx_fused_amax_convert(ctypes.c_void_p(16), ctypes.c_void_p(16), 4, 32, 32, 2, 6, _lib.MX_TYPES["E8M0"],
                                         ctypes.c_void_p(16), None) \
        == _lib.MOQ_ERR_UNSUPPORTED  # E8M0 block scales take no global amax -> loud
    # the entries added later in the round: argument checks happen before any HIP call
    P = ctypes.c_void_p(16)
    assert lib.moq_row_hist_np(P, 4, 64, _lib.BF16, 0, P, P, P, None) == _lib.MOQ_ERR_INVALID           # bins <= 0
    assert lib.moq_row_hist_np(P, 70000, 64, _lib.BF16, 2048, P, P, P, None) == _lib.MOQ_ERR_UNSUPPORTED  # rows > 65535
    assert lib.moq_row_hist_np(None, 0, 64, _lib.BF16, 2048, None, None, None, None) == _lib.MOQ_OK      # empty: nothing to do
    assert lib.moq_fp8_pack_tile(P, P, _lib.F16, P, 256, 256, 128, 128, _lib.BF16, None) == _lib.MOQ_ERR_INVALID  # scale dtype
    assert lib.moq_fp8_pack_tile(P, P, _lib.BF16, P, 200, 256, 128, 128, _lib.BF16, None) == _lib.MOQ_ERR_UNSUPPORTED  # ragged tiles
    assert lib.moq_fp8_unpack_tile(None, P, P, 256, 256, 128, 128, _lib.BF16, None) == _lib.MOQ_ERR_INVALID
    # (round 3: the tile packers index rows and packets in 32 bits -- wider tensors are refused, not wrapped)
    P16 = ctypes.c_void_p(32)
    assert lib.moq_fp8_pack_tile(P16, P16, _lib.BF16, P16, 1 << 32, 128, 128, 128, _lib.BF16, None) == _lib.MOQ_ERR_UNSUPPORTED
    assert b"2^31" in lib.moq_last_error()
    assert lib.moq_fp8_unpack_tile(P16, P16, P16, 1 << 32, 128, 128, 128, _lib.BF16, None) == _lib.MOQ_ERR_UNSUPPORTED
    assert lib.moq_amax_mid(P, 4, 0, 8, _lib.BF16, P, None) == _lib.MOQ_ERR_INVALID                      # empty reduced dim
    assert lib.moq_amax_mid(None, 0, 4, 8, _lib.BF16, None, None) == _lib.MOQ_OK
    assert lib.moq_mx_convert(P, P, 8, 99, None) == _lib.MOQ_ERR_INVALID                                  # unknown format
    assert lib.moq_mx_convert(None, None, 0, _lib.MX_TYPES["E2M1"], None) == _lib.MOQ_OK
    assert lib.moq_transpose16_ld(P, P, 64, 32, 63, None) == _lib.MOQ_ERR_INVALID                         # y_ld < rows
    assert lib.moq_mt_amax_ws(P, P, 3, 10, _lib.BF16, None, None) == _lib.MOQ_ERR_INVALID                 # no scratch
    assert b"chunk_scratch" in lib.moq_last_error()
    assert lib.moq_mt_amax_running(P, P, 3, 10, _lib.BF16, None, P, 2, None) == _lib.MOQ_ERR_INVALID      # no scratch
    assert lib.moq_mt_amax_running(P, P, 3, 10, _lib.BF16, P, None, 2, None) == _lib.MOQ_ERR_INVALID      # folds missing
    assert lib.moq_mt_amax_running(P, P, 0, 0, _lib.BF16, None, None, 0, None) == _lib.MOQ_OK             # nothing to do
    with pytest.raises(ValueError):
        _lib.check(_lib.MOQ_ERR_UNSUPPORTED)
    with pytest.raises(RuntimeError):
        _lib.check(_lib.MOQ_ERR_INVALID)


def test_mt_plan_host_helper():
    lib = _lib.lib()
    n = (ctypes.c_int64 * 4)(0, 1, 8192, 8193)
    blk = (ctypes.c_int64 * 5)()
    assert lib.moq_mt_plan(n, 4, blk) == 4
    assert list(blk) == [0, 0, 1, 2, 4]


def test_ops_refuse_cpu_tensors():
    x = torch.randn(4, 128)
    for call in (lambda: moa.ops.reduce_amax(x),
                 lambda: moa.ops.fake_tensor_quant(x, torch.tensor(1.0)),
                 lambda: moa.ops.scaled_e4m3(x, torch.tensor(1.0)),
                 lambda: moa.ops.amax_qdq_int_group(x, 128),
                 lambda: moa.ops.mask_2to4(x),
                 lambda: moa.ops.fused_amax_convert(x, 32, "E2M1")):
        with pytest.raises(moa.MoquantError, match="must live on the GPU"):
            call()


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "model-optimizer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                text = open(os.path.join(dirpath, f)).read()
                assert "moq_oracle" not in text and "from oracle" not in text and "import oracle" not in text, \
                    f"{f} references the oracle (test infrastructure must stay out of the product path)"
