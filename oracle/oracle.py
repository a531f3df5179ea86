"""Python face of the CPU oracle (oracle/moq_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package.  Each function takes/returns torch CPU tensors and forwards to the C
restatement of the reference algorithm (see the C file for reference file:line citations).
Parity status: PINNED against reference-generated fixtures and the reference tests' golden vectors
(tests/test_oracle_golden.py).
"""

from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libmoq_oracle.so")
_lib = None

DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}
MX_TYPES = {"E4M3": 0, "E5M2": 1, "INT8": 2, "E0M3": 3, "E1M2": 4, "E3M0": 5, "E2M1": 6, "E3M2": 7,
            "E2M3": 8, "E8M0": 9}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "moq_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(src) > os.path.getmtime(_LIB_PATH):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


def usable_cpus() -> int:
    """Host threads this process may really use: the smaller of the visible CPUs, the affinity mask and the cgroup CPU
    quota (a GPU box shows 256 CPUs under a 16-CPU quota; 256 OpenMP threads there run 10x SLOWER than 16)."""
    n = min(os.cpu_count() or 1, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 1 << 30)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, -(-quota // period)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def set_threads(n: int) -> None:
    try:
        ctypes.CDLL("libgomp.so.1").omp_set_num_threads(int(n))
    except OSError:
        pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        set_threads(usable_cpus())
        _lib.orc_reduce_amax.restype = ctypes.c_float
        _lib.orc_e4m3fn_round.restype = ctypes.c_float
        _lib.orc_e4m3fn_round.argtypes = [ctypes.c_float]
    return _lib


def _np(t: torch.Tensor) -> np.ndarray:
    """Contiguous numpy view of a CPU tensor (16-bit floats as uint16 patterns)."""
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16)
    return t.numpy()


def _from_np(a: np.ndarray, dtype: torch.dtype, shape) -> torch.Tensor:
    if dtype in (torch.bfloat16, torch.float16):
        return torch.from_numpy(a.view(np.int16)).view(dtype).reshape(shape)
    return torch.from_numpy(a).reshape(shape)


def _p(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _empty_like_np(t: torch.Tensor) -> np.ndarray:
    if t.dtype in (torch.bfloat16, torch.float16):
        return np.empty(t.numel(), dtype=np.uint16)
    return np.empty(t.numel(), dtype=np.float32)


I64 = ctypes.c_int64


def reduce_amax(x: torch.Tensor) -> torch.Tensor:
    a = _np(x)
    v = lib().orc_reduce_amax(_p(a), I64(x.numel()), DT[x.dtype])
    return torch.tensor(v, dtype=torch.float32)


def reduce_amax_axis(x: torch.Tensor, outer: int, axis_size: int, inner: int) -> torch.Tensor:
    a = _np(x)
    out = np.empty(axis_size, dtype=np.float32)
    lib().orc_reduce_amax_axis(_p(a), I64(outer), I64(axis_size), I64(inner), DT[x.dtype], _p(out))
    return torch.from_numpy(out)


def _amax_args(amax, mode_axis):
    if amax is None:
        return None, 0
    am = np.ascontiguousarray(amax.detach().cpu().float().reshape(-1).numpy())
    return am, (1 if mode_axis else 0)


def fake_quant_int(x, amax, num_bits=8, unsigned=False, narrow_range=True, axis_size=1, inner=1,
                   per_axis=False):
    a = _np(x)
    y = _empty_like_np(x)
    am, mode = _amax_args(amax, per_axis)
    lib().orc_fake_quant_int(_p(a), _p(y), I64(x.numel()), DT[x.dtype], _p(am), mode, I64(axis_size),
                             I64(inner), int(num_bits), int(unsigned), int(narrow_range))
    return _from_np(y, x.dtype, x.shape)


def fake_quant_e4m3(x, amax=None, axis_size=1, inner=1, per_axis=False):
    a = _np(x)
    y = _empty_like_np(x)
    am, mode = _amax_args(amax, per_axis)
    lib().orc_fake_quant_e4m3(_p(a), _p(y), I64(x.numel()), DT[x.dtype], _p(am), mode, I64(axis_size),
                              I64(inner))
    return _from_np(y, x.dtype, x.shape)


def amax_qdq_int_group(x, g, num_bits=4, unsigned=False, narrow_range=False, qdq=True):
    assert x.numel() % g == 0
    a = _np(x)
    ng = x.numel() // g
    y = _empty_like_np(x) if qdq else None
    am = np.empty(ng, dtype=np.float32)
    lib().orc_amax_qdq_int_group(_p(a), _p(y), _p(am), I64(ng), int(g), DT[x.dtype], int(num_bits),
                                 int(unsigned), int(narrow_range))
    return (_from_np(y, x.dtype, x.shape) if qdq else None), torch.from_numpy(am)


def mx_fused_amax_convert(x, block, fmt="E2M1", scale_fmt="E8M0", global_amax=None):
    cols = x.shape[-1]
    rows = x.numel() // cols
    a = _np(x)
    y = _empty_like_np(x)
    g = None if global_amax is None else np.ascontiguousarray(global_amax.detach().cpu().float().reshape(-1).numpy()[:1])
    lib().orc_mx_fused_amax_convert2(_p(a), _p(y), I64(rows), I64(cols), int(block), DT[x.dtype],
                                     MX_TYPES[fmt], MX_TYPES[scale_fmt], _p(g))
    return _from_np(y, x.dtype, x.shape)


def mx_convert(x, fmt):
    a = np.ascontiguousarray(x.detach().cpu().float().numpy())
    y = np.empty_like(a)
    lib().orc_mx_convert(_p(a), _p(y), I64(a.size), MX_TYPES[fmt] if isinstance(fmt, str) else int(fmt))
    return torch.from_numpy(y)


def hist_abs(x, bins, max_edge, skip_zeros=False, counts=None):
    a = _np(x)
    c = np.zeros(bins, dtype=np.uint64) if counts is None else counts
    lib().orc_hist_abs(_p(a), I64(x.numel()), DT[x.dtype], _p(c), int(bins),
                       ctypes.c_float(float(max_edge)), int(skip_zeros))
    return c


def row_hist_np(w, bins, first, last):
    """Per-row |w| histograms with np.histogram's float32 edges (calibrate_weights, calib/histogram.py:346-433)."""
    rows, cols = w.shape
    a = _np(w)
    counts = np.zeros((rows, bins), dtype=np.int32)
    f = np.ascontiguousarray(first.detach().cpu().float().numpy())
    la = np.ascontiguousarray(last.detach().cpu().float().numpy())
    lib().orc_row_hist_np(_p(a), I64(rows), I64(cols), DT[w.dtype], int(bins), _p(f), _p(la), _p(counts))
    return torch.from_numpy(counts)


def mask_2to4(w):
    cols = w.shape[-1]
    rows = w.numel() // cols
    a = _np(w)
    m = np.empty(w.numel(), dtype=np.uint8)
    lib().orc_mask_2to4(_p(a), I64(rows), I64(cols), DT[w.dtype], _p(m))
    return torch.from_numpy(m).reshape(w.shape).bool()


def int4_pack(x, scales, g, rounding=0):
    a, s = _np(x), _np(scales)
    out = np.empty(x.numel() // 2, dtype=np.uint8)
    lib().orc_int4_pack(_p(a), _p(s), _p(out), I64(x.numel()), int(g), DT[x.dtype], int(rounding))
    return torch.from_numpy(out)


def int4_unpack(q, scales, g):
    qa = np.ascontiguousarray(q.detach().cpu().reshape(-1).numpy())
    s = _np(scales)
    out = _empty_like_np(torch.empty(qa.size * 2, dtype=scales.dtype))
    lib().orc_int4_unpack(_p(qa), _p(s), _p(out), I64(qa.size), int(g), DT[scales.dtype])
    return _from_np(out, scales.dtype, (qa.size * 2,))


def int4_pack_export(w, wsf):
    rows, cols = w.shape
    g = cols // wsf.shape[-1]
    a = _np(w)
    s = np.ascontiguousarray(wsf.detach().cpu().float().numpy())
    out = np.empty(rows // 2 * cols, dtype=np.uint8)
    lib().orc_int4_pack_export(_p(a), _p(s), _p(out), I64(rows), I64(cols), int(g), DT[w.dtype])
    return torch.from_numpy(out).reshape(rows // 2, cols)


def scale_cols(w, s):
    rows, cols = w.shape
    a = _np(w)
    sv = np.ascontiguousarray(s.detach().cpu().float().reshape(-1).numpy())
    y = _empty_like_np(w)
    lib().orc_scale_cols(_p(a), _p(sv), _p(y), I64(rows), I64(cols), DT[w.dtype])
    return _from_np(y, w.dtype, w.shape)


def awq_scale_qdq(w, s, g, num_bits=4):
    rows, cols = w.shape
    a, sv = _np(w), _np(s.to(w.dtype))
    y = _empty_like_np(w)
    lib().orc_awq_scale_qdq(_p(a), _p(sv), _p(y), I64(rows), I64(cols), int(g), DT[w.dtype],
                            int(num_bits))
    return _from_np(y, w.dtype, w.shape)


def col_abs_stats(x):
    tokens, cols = x.shape
    a = _np(x)
    s = np.empty(cols, dtype=np.float64)
    m = np.empty(cols, dtype=np.float32)
    lib().orc_col_abs_stats(_p(a), I64(tokens), I64(cols), DT[x.dtype], _p(s), _p(m))
    return torch.from_numpy(s), torch.from_numpy(m)


def e4m3fn_round(v: float) -> float:
    return float(lib().orc_e4m3fn_round(ctypes.c_float(v)))


def awq_weight_scale(w, g):
    rows, cols = w.shape
    a = _np(w)
    out = np.empty(cols, dtype=np.float32)
    lib().orc_awq_weight_scale(_p(a), I64(rows), I64(cols), int(g), DT[w.dtype], _p(out))
    return torch.from_numpy(out)


def awq_err_gemm(x, w, out_actual=None, bias=None, return_out=False):
    """(mean squared error vs out_actual, out) of out = linear(x, w, bias) in x.dtype (bf16 / f16 / f32)."""
    tokens, cin = x.shape
    cout = w.shape[0]
    a, b = _np(x), _np(w)
    r = None if out_actual is None else _np(out_actual)
    bi = None if bias is None else _np(bias)
    o = _empty_like_np(torch.empty(tokens * cout, dtype=x.dtype)) if return_out else None
    f = lib().orc_awq_err_gemm
    f.restype = ctypes.c_double
    loss = f(_p(a), _p(b), _p(r), _p(bi), _p(o), I64(tokens), I64(cout), I64(cin), DT[x.dtype])
    return (float(loss), _from_np(o, x.dtype, (tokens, cout))) if return_out else float(loss)


def mse_sweep(x, cand_amax, outer, axis_size, inner, fp8=False, num_bits=8, unsigned=False, narrow_range=False):
    """loss[k, a] = sum (x - QDQ(x, cand_amax[k, a]))^2 over the elements of amax entry a; fp64 [K, axis_size]."""
    a = _np(x)
    c = np.ascontiguousarray(cand_amax.detach().cpu().float().numpy())
    k = c.shape[0]
    loss = np.empty((k, axis_size), dtype=np.float64)
    lib().orc_mse_sweep(_p(a), I64(outer), I64(axis_size), I64(inner), DT[x.dtype], _p(c), int(k), _p(loss),
                        int(fp8), int(num_bits), int(unsigned), int(narrow_range))
    return torch.from_numpy(loss)


def rescale_cols(w, mul, div):
    rows, cols = w.shape
    a = _np(w)
    m = np.ascontiguousarray(mul.detach().cpu().float().reshape(-1).numpy())
    d = np.ascontiguousarray(div.detach().cpu().float().reshape(-1).numpy())
    y = _empty_like_np(w)
    lib().orc_rescale_cols(_p(a), _p(m), _p(d), _p(y), I64(rows), I64(cols), DT[w.dtype])
    return _from_np(y, w.dtype, w.shape)


def awq_clip_loss(x, w, amax, shrinks, g, num_bits=4, loss=None):
    """loss[k, r, b] (+)= mean_t (cur_k - org)^2 of awq_clip's block search (model_calib.py:1800-1868).
    x [n_tok <= 256, cin], w [cout, cin] (same dtype); amax [cout, nblk] in the dtype w_amax has in the reference
    (weight dtype or fp32); shrinks: python floats.  fp32 [n_shrink, cout, nblk]."""
    n_tok, cin = x.shape
    cout = w.shape[0]
    nblk = (cin + g - 1) // g
    assert n_tok <= 256 and g <= 1024 and tuple(amax.shape) == (cout, nblk)
    a, b = _np(x), _np(w)
    am = np.ascontiguousarray(amax.detach().cpu().float().numpy())
    sh = np.asarray([float(s) for s in shrinks], dtype=np.float32)
    if loss is None:
        loss = np.zeros((len(sh), cout, nblk), dtype=np.float32)
    else:
        loss = np.ascontiguousarray(loss.detach().cpu().float().numpy())
    lib().orc_awq_clip_loss(_p(a), I64(n_tok), _p(b), I64(cout), I64(cin), int(g), DT[w.dtype], _p(am),
                            DT[amax.dtype], _p(sh), int(len(sh)), int(num_bits), _p(loss))
    return torch.from_numpy(loss)


def _mode(x, scales, axis_size, inner):
    return (0, 1, 1) if scales.numel() == 1 else (1, axis_size, inner)


def fp8_pack(x, scales, axis_size=1, inner=1, fp32_scales=False):
    """(x / scales).to(float8_e4m3fn) bytes; scales in x.dtype (FP8QTensor.quantize, fp8_tensor.py:103-107) or fp32
    (to_quantized_weight of the checkpoint export)."""
    sdt = torch.float32 if fp32_scales else x.dtype
    a, s = _np(x), _np(scales.to(sdt))
    out = np.empty(x.numel(), dtype=np.uint8)
    m, ax, inn = _mode(x, scales, axis_size, inner)
    lib().orc_fp8_pack(_p(a), _p(s), DT[sdt], _p(out), I64(x.numel()), DT[x.dtype], m, I64(ax), I64(inn))
    return torch.from_numpy(out).reshape(x.shape)


def fp8_unpack(q, scales, dtype, axis_size=1, inner=1):
    qa = np.ascontiguousarray(q.detach().cpu().contiguous().view(torch.uint8).numpy()).reshape(-1)
    s = _np(scales.to(dtype))
    out = _empty_like_np(torch.empty(qa.size, dtype=dtype))
    m, ax, inn = _mode(q, scales, axis_size, inner)
    lib().orc_fp8_unpack(_p(qa), _p(s), _p(out), I64(qa.size), DT[dtype], m, I64(ax), I64(inn))
    return _from_np(out, dtype, q.shape)


def fp8_pack_tile(x, scales, br, bc):
    """FP8QTensor.quantize with blocks on both axes (fp8_tensor.py:60-112) as a composition of the pinned per-tensor
    restatement: every br x bc tile is packed with its own scalar scale.  A 0-dim fp32 scale with a 16-bit tensor
    would NOT promote in torch, a dimensioned one does -- so for fp32 scales the tile is upcast first."""
    rows, cols = x.shape
    out = torch.empty(rows, cols, dtype=torch.uint8)
    promote = scales.dtype == torch.float32 and x.dtype != torch.float32
    s2 = scales.reshape(rows // br, cols // bc)
    for i in range(rows // br):
        for j in range(cols // bc):
            tile = x[i * br:(i + 1) * br, j * bc:(j + 1) * bc].contiguous()
            if promote:
                tile = tile.float()
            out[i * br:(i + 1) * br, j * bc:(j + 1) * bc] = fp8_pack(tile, s2[i, j].reshape(1).to(tile.dtype))
    return out


def fp8_unpack_tile(q, scales, dtype, br, bc):
    rows, cols = q.shape
    out = torch.empty(rows, cols, dtype=dtype)
    s2 = scales.reshape(rows // br, cols // bc)
    for i in range(rows // br):
        for j in range(cols // bc):
            out[i * br:(i + 1) * br, j * bc:(j + 1) * bc] = fp8_unpack(
                q[i * br:(i + 1) * br, j * bc:(j + 1) * bc].contiguous(), s2[i, j].reshape(1), dtype)
    return out


def mxfp4_pack(x, block=32):
    """MXFP4QTensor.quantize (mxfp4_tensor.py:37-81): (packed uint8 [..., K/2], e8m0 uint8 [n/block, 1])."""
    a = _np(x)
    nb = x.numel() // block
    packed = np.empty(x.numel() // 2, dtype=np.uint8)
    e8 = np.empty(nb, dtype=np.uint8)
    lib().orc_mxfp4_pack(_p(a), _p(packed), _p(e8), I64(nb), int(block), DT[x.dtype])
    return torch.from_numpy(packed).reshape(*x.shape[:-1], x.shape[-1] // 2), torch.from_numpy(e8).reshape(-1, 1)


def mxfp4_unpack(packed, e8m0, dtype, block=32):
    p = np.ascontiguousarray(packed.detach().cpu().numpy()).reshape(-1)
    e = np.ascontiguousarray(e8m0.detach().cpu().numpy()).reshape(-1)
    out = _empty_like_np(torch.empty(p.size * 2, dtype=dtype))
    lib().orc_mxfp4_unpack(_p(p), _p(e), _p(out), I64(e.size), int(block), DT[dtype])
    return _from_np(out, dtype, (*packed.shape[:-1], packed.shape[-1] * 2))


def sgpt_block_sweep(w, i1, bs, hinv, prune_n=2, prune_m=4):
    """In-place column sweep of create_sgpt_mask over block [i1, i1 + bs) (sparsegpt.py:96-127); returns delta."""
    assert w.dtype == torch.float32 and w.is_contiguous() and hinv.dtype == torch.float32 and bs <= 1024
    rows, ld = w.shape
    wa = w.numpy()
    ha = np.ascontiguousarray(hinv.numpy())
    delta = np.zeros((rows, bs), dtype=np.float32)
    lib().orc_sgpt_block_sweep(_p(wa), I64(rows), I64(ld), I64(i1), int(bs), _p(ha), _p(delta), int(prune_n),
                               int(prune_m))
    return torch.from_numpy(delta)


def gptq_block_sweep(w, i1, bs, hinv, amax, amax_row_stride, g, fmt, num_bits=8, unsigned=False, narrow=False):
    """In-place column sweep of gptq_blockwise_update over block [i1, i1 + bs) (calib_utils.py:241-276) for a static
    amax; fmt 1: INT-num_bits, 2: FP8-E4M3; returns the errors [rows, bs]."""
    assert w.dtype == torch.float32 and w.is_contiguous() and hinv.dtype == torch.float32 and bs <= 1024
    rows, ld = w.shape
    delta = np.zeros((rows, bs), dtype=np.float32)
    am = np.zeros(1, dtype=np.float32) if amax is None else np.ascontiguousarray(amax.detach().float().reshape(-1).numpy())
    if fmt in (3, 4) and isinstance(num_bits, str):
        num_bits = MX_TYPES[num_bits]
    if fmt == 4:
        unsigned = MX_TYPES[unsigned] if isinstance(unsigned, str) else int(unsigned)
    lib().orc_gptq_block_sweep(_p(w.numpy()), I64(rows), I64(ld), I64(i1), int(bs), _p(np.ascontiguousarray(hinv.numpy())),
                               _p(delta), _p(am), I64(amax_row_stride), I64(g), int(fmt), int(num_bits),
                               int(unsigned) if fmt == 4 else int(bool(unsigned)), int(bool(narrow)))
    return torch.from_numpy(delta)


def sgpt_trailing_update(w, i1, delta, hinv):
    """In place: w[:, i2:] -= delta @ hinv[i1:i2, i2:] (sparsegpt.py:124) as the ascending-k fmaf chain -- the summation
    order the product path defines where the reference has its BLAS library's."""
    assert w.dtype == torch.float32 and w.is_contiguous() and delta.dtype == torch.float32 and hinv.dtype == torch.float32
    rows, ld = w.shape
    lib().orc_sgpt_trailing_update(_p(w.numpy()), I64(rows), I64(ld), I64(i1), int(delta.shape[1]),
                                   _p(np.ascontiguousarray(delta.numpy())), _p(np.ascontiguousarray(hinv.numpy())))
    return w


def create_sgpt_mask(weight, hinv, prune_n=2, prune_m=4, col_bs=128, blas_update=False):
    """create_sgpt_mask (sparsegpt.py:72-133) given the prepared inverse-Hessian factor: mask = pruned weight != 0.
    blas_update=True: the trailing update as the reference writes it (torch matmul, the CPU BLAS order)."""
    w = weight.detach().float().clone().contiguous()
    cols = w.shape[1]
    for i1 in range(0, cols, col_bs):
        i2 = min(i1 + col_bs, cols)
        delta = sgpt_block_sweep(w, i1, i2 - i1, hinv, prune_n, prune_m)
        if i2 < cols:
            if blas_update:
                w[:, i2:] -= delta.matmul(hinv[i1:i2, i2:])
            else:
                sgpt_trailing_update(w, i1, delta, hinv)
    return w.to(weight.dtype) != 0


def block2d(x, br, bc, mode=2, amax=None, fp8=True, num_bits=8, unsigned=False, narrow_range=False):
    """2-D block amax / QDQ over br x bc tiles of x [rows, cols]: returns (y or None, amax fp32 [rows/br, cols/bc])."""
    rows, cols = x.shape
    a = _np(x)
    y = _empty_like_np(x) if mode != 0 else None
    am = np.zeros((rows // br, cols // bc), dtype=np.float32) if amax is None else \
        np.ascontiguousarray(amax.detach().cpu().float().numpy()).reshape(rows // br, cols // bc).copy()
    lib().orc_block2d(_p(a), _p(y), _p(am), I64(rows), I64(cols), int(br), int(bc), DT[x.dtype], int(mode), int(fp8),
                      int(num_bits), int(unsigned), int(narrow_range))
    return (None if y is None else _from_np(y, x.dtype, x.shape)), torch.from_numpy(am)
