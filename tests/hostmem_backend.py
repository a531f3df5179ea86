"""TEST INFRASTRUCTURE: a host-memory stand-in for libmoquant.so, backed by the CPU oracle.

The Python host (ops.py wrappers, TensorQuantizer, calibrators, calibration algorithms, export) talks to the kernels
only through the C-ABI of include/moquant.h with raw pointers.  For CPU tensors those pointers are host addresses, so
the same calls can be served by the oracle's C restatement (oracle/moq_oracle.c) -- entry by entry, same argument
order.  `install(monkeypatch)` swaps `_lib.lib()` for this object and lets CPU tensors through the wrappers, which
makes the WHOLE host side runnable in the `-m "not gpu"` tier: calibration flows and checkpoint exports are then
compared with what the reference produced on the same CPU (tests/test_host_flows_cpu.py), without a GPU.

This is a checker, never a product path: nothing under model-optimizer_amd/ imports it, the product still raises on
CPU tensors (tests/test_abi_cpu.py::test_ops_refuse_cpu_tensors).  Entries without an oracle twin raise.
"""

from __future__ import annotations

import ctypes
from contextlib import contextmanager

import numpy as np

from oracle import oracle

I64, F32 = ctypes.c_int64, ctypes.c_float
OK = 0


def _addr(p) -> int:
    if p is None:
        return 0
    if isinstance(p, ctypes.c_void_p):
        return p.value or 0
    return int(p)


def _f32_view(p, n):
    return np.ctypeslib.as_array(ctypes.cast(_addr(p), ctypes.POINTER(ctypes.c_float)), shape=(int(n),))


def _vp(p):
    return ctypes.c_void_p(_addr(p))


class HostMemLib:
    """Same method names and argument order as the ctypes SIGNATURES of model_optimizer_amd._lib."""

    def __init__(self):
        self.o = oracle.lib()
        self._err = b""

    def __getattr__(self, name):
        if name.startswith("moq_"):
            def missing(*a, **k):
                raise NotImplementedError(f"hostmem_backend: {name} has no oracle twin")
            return missing
        raise AttributeError(name)

    # -- misc
    def moq_abi_version(self):
        return 1

    def moq_last_error(self):
        return self._err

    def moq_mt_plan(self, n, n_seg, blk):
        total = 0
        for i in range(n_seg):
            blk[i] = total
            total += (n[i] + 8191) // 8192
        blk[n_seg] = total
        return total

    # -- a1
    def moq_amax(self, x, n, dt, out, accumulate, stream):
        self.o.orc_reduce_amax.restype = ctypes.c_float
        v = self.o.orc_reduce_amax(_vp(x), I64(n), int(dt)) if n else 0.0
        o = _f32_view(out, 1)
        if accumulate:
            v = np.float32(v)
            o[0] = v if (np.isnan(v) or v > o[0]) else o[0]
        else:
            o[0] = v
        return OK

    def moq_amax_axis(self, x, outer, axis_size, inner, dt, out, accumulate, stream):
        tmp = np.zeros(int(axis_size), dtype=np.float32)
        if outer * axis_size * inner:
            self.o.orc_reduce_amax_axis(_vp(x), I64(outer), I64(axis_size), I64(inner), int(dt), oracle._p(tmp))
        o = _f32_view(out, axis_size)
        if accumulate:
            o[:] = np.where(np.isnan(tmp) | (tmp > o), tmp, o)
        else:
            o[:] = tmp
        return OK

    def moq_amax_mid(self, x, outer, mid, inner, dt, out, stream):
        # [outer, mid, inner] reduced over mid == axis reduction of the transposed problem, one outer slab at a time
        elem = 4 if dt == 0 else 2  # MOQ_F32 == 0
        o = _f32_view(out, outer * inner)
        tmp = np.zeros(int(inner), dtype=np.float32)
        for k in range(int(outer)):
            self.o.orc_reduce_amax_axis(ctypes.c_void_p(_addr(x) + k * mid * inner * elem), I64(mid), I64(inner), I64(1),
                                        int(dt), oracle._p(tmp))
            o[k * inner:(k + 1) * inner] = tmp
        return OK

    # -- a6 / a7 / fused
    def moq_fake_quant_int(self, x, y, n, dt, amax, mode, axis_size, inner, num_bits, unsigned, narrow, stream):
        self.o.orc_fake_quant_int(_vp(x), _vp(y), I64(n), int(dt), _vp(amax), int(mode), I64(axis_size), I64(inner),
                                  int(num_bits), int(unsigned), int(narrow))
        return OK

    def moq_fake_quant_e4m3(self, x, y, n, dt, amax, mode, axis_size, inner, stream):
        self.o.orc_fake_quant_e4m3(_vp(x), _vp(y), I64(n), int(dt), _vp(amax), int(mode), I64(axis_size), I64(inner))
        return OK

    def moq_amax_qdq_int_group(self, x, y, amax, n_groups, g, dt, num_bits, unsigned, narrow, stream):
        tmp = np.zeros(int(n_groups), dtype=np.float32)
        self.o.orc_amax_qdq_int_group(_vp(x), _vp(y), oracle._p(tmp), I64(n_groups), int(g), int(dt), int(num_bits),
                                      int(unsigned), int(narrow))
        if _addr(amax):
            _f32_view(amax, n_groups)[:] = tmp
        return OK

    def moq_block2d(self, x, y, amax, rows, cols, br, bc, dt, mode, accumulate, fp8, num_bits, unsigned, narrow, stream):
        nt = (rows // br) * (cols // bc)
        if mode == 1:
            self.o.orc_block2d(_vp(x), _vp(y), _vp(amax), I64(rows), I64(cols), int(br), int(bc), int(dt), 1, int(fp8),
                               int(num_bits), int(unsigned), int(narrow))
            return OK
        tmp = np.zeros(nt, dtype=np.float32)
        self.o.orc_block2d(_vp(x), _vp(y), oracle._p(tmp), I64(rows), I64(cols), int(br), int(bc), int(dt), int(mode),
                           int(fp8), int(num_bits), int(unsigned), int(narrow))
        if _addr(amax):
            o = _f32_view(amax, nt)
            o[:] = np.where(np.isnan(tmp) | (tmp > o), tmp, o) if accumulate else tmp
        return OK

    def moq_mx_fused_amax_convert(self, x, y, rows, cols, block, dt, fmt, scale_fmt, global_amax, stream):
        self.o.orc_mx_fused_amax_convert2(_vp(x), _vp(y), I64(rows), I64(cols), int(block), int(dt), int(fmt),
                                          int(scale_fmt), _vp(global_amax))
        return OK

    def moq_mx_convert(self, x, y, n, fmt, stream):
        self.o.orc_mx_convert(_vp(x), _vp(y), I64(n), int(fmt))
        return OK

    # -- statistics
    def moq_hist_abs(self, x, n, dt, counts, bins, max_edge, skip_zeros, stream):
        self.o.orc_hist_abs(_vp(x), I64(n), int(dt), _vp(counts), int(bins), F32(max_edge), int(skip_zeros))
        return OK

    def moq_row_hist_np(self, x, rows, cols, dt, bins, first, last, counts, stream):
        # the C-ABI call OVERWRITES counts (it may be handed uninitialised memory); the oracle accumulates
        ctypes.memset(_addr(counts), 0, int(rows) * int(bins) * 4)
        self.o.orc_row_hist_np(_vp(x), I64(rows), I64(cols), int(dt), int(bins), _vp(first), _vp(last), _vp(counts))
        return OK

    def moq_col_stats_workspace(self, tokens, cols):
        return 0

    def moq_col_abs_stats(self, x, tokens, cols, dt, sum_out, amax_out, workspace, accumulate, stream):
        s64 = np.zeros(int(cols), dtype=np.float64)
        am = np.zeros(int(cols), dtype=np.float32)
        self.o.orc_col_abs_stats(_vp(x), I64(tokens), I64(cols), int(dt), oracle._p(s64), oracle._p(am))
        if _addr(sum_out):
            so = _f32_view(sum_out, cols)
            so[:] = (so + s64.astype(np.float32)) if accumulate else s64.astype(np.float32)
        if _addr(amax_out):
            ao = _f32_view(amax_out, cols)
            ao[:] = np.where(np.isnan(am) | (am > ao), am, ao) if accumulate else am
        return OK

    def moq_col_abs_mean_accum(self, x, tokens, cols, dt, acc, workspace, stream):
        s64 = np.zeros(int(cols), dtype=np.float64)
        self.o.orc_col_abs_stats(_vp(x), I64(tokens), I64(cols), int(dt), oracle._p(s64), None)
        import torch

        tdt = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}[int(dt)]
        mean = (torch.from_numpy(s64.astype(np.float32)) / float(tokens)).to(tdt).float().numpy()
        a = _f32_view(acc, cols)
        a[:] = a + mean
        return OK

    def moq_int8_pack_rows(self, w, scale, out, rows, cols, dt, stream):
        import torch

        tdt = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}[int(dt)]
        n = int(rows) * int(cols)
        src = (ctypes.c_char * (n * (4 if int(dt) == 0 else 2))).from_address(_addr(w))
        wt = torch.frombuffer(bytearray(src), dtype=tdt).reshape(int(rows), int(cols))
        sc = torch.from_numpy(_f32_view(scale, rows).copy())
        q = (wt / sc[:, None]).round().clamp(-128, 127).to(torch.int8)  # the reference's own expression
        ctypes.memmove(_addr(out), q.contiguous().numpy().tobytes(), n)
        return OK

    def moq_input_quant(self, x, pqs, y, rows, cols, dt, amax_running, qdq_amax, fmt, num_bits, unsigned, narrow,
                        hist_counts, hist_bins, hist_max_edge, hist_skip_zeros, stream):
        """The fused pass as the chain of the oracle's single stages (scale -> amax -> histogram -> QDQ)."""
        import torch

        tdt = {0: torch.float32, 1: torch.float16, 2: torch.bfloat16}[int(dt)]
        n = int(rows) * int(cols)
        es = 4 if int(dt) == 0 else 2
        src = (ctypes.c_char * (n * es)).from_address(_addr(x))
        v = torch.frombuffer(bytearray(src), dtype=tdt).reshape(int(rows), int(cols))
        if _addr(pqs):
            sv = torch.from_numpy(_f32_view(pqs, cols).copy())
            v = oracle.scale_cols(v, sv)
        if _addr(amax_running):
            a = _f32_view(amax_running, 1)
            m = float(oracle.reduce_amax(v).float())
            a[0] = m if (m != m or m > a[0]) else a[0]
        if _addr(hist_counts):
            cnt = np.ctypeslib.as_array(ctypes.cast(_addr(hist_counts), ctypes.POINTER(ctypes.c_int64)), shape=(int(hist_bins),))
            cnt += oracle.hist_abs(v.reshape(-1), int(hist_bins), float(hist_max_edge), bool(hist_skip_zeros)).astype(np.int64)
        if int(fmt) == 1:
            am = torch.from_numpy(_f32_view(qdq_amax, 1).copy())
            v = oracle.fake_quant_int(v, am, int(num_bits), bool(unsigned), bool(narrow))
        elif int(fmt) == 2:
            am = torch.from_numpy(_f32_view(qdq_amax, 1).copy())
            v = oracle.fake_quant_e4m3(v, am)
        if _addr(y) and (int(fmt) or _addr(pqs)):
            raw = v.contiguous().view(torch.uint8).numpy().tobytes()
            ctypes.memmove(_addr(y), raw, len(raw))
        return OK

    def moq_mse_sweep_workspace(self, outer, axis_size, inner, n_cand):
        return 0

    def moq_mse_sweep(self, x, outer, axis_size, inner, dt, cand, n_cand, loss, partial, accumulate, fp8, num_bits,
                      unsigned, narrow, stream):
        l64 = np.zeros(int(n_cand) * int(axis_size), dtype=np.float64)
        self.o.orc_mse_sweep(_vp(x), I64(outer), I64(axis_size), I64(inner), int(dt), _vp(cand), int(n_cand),
                             oracle._p(l64), int(fp8), int(num_bits), int(unsigned), int(narrow))
        o = _f32_view(loss, n_cand * axis_size)
        o[:] = (o + l64.astype(np.float32)) if accumulate else l64.astype(np.float32)
        return OK

    # -- AWQ / smooth building blocks
    def moq_scale_cols(self, w, s, y, rows, cols, dt, stream):
        self.o.orc_scale_cols(_vp(w), _vp(s), _vp(y), I64(rows), I64(cols), int(dt))
        return OK

    def moq_rescale_cols(self, w, mul, div, y, rows, cols, dt, stream):
        self.o.orc_rescale_cols(_vp(w), _vp(mul), _vp(div), _vp(y), I64(rows), I64(cols), int(dt))
        return OK

    def moq_awq_scale_qdq(self, w, s, y, rows, cols, g, dt, num_bits, stream):
        if rows < 0 or cols <= 0 or g <= 0 or cols % g != 0:  # the C-ABI's own argument check (moq_stream.hip)
            self._err = b"moq_awq_scale_qdq: cols must be a positive multiple of g"
            return -1
        self.o.orc_awq_scale_qdq(_vp(w), _vp(s), _vp(y), I64(rows), I64(cols), int(g), int(dt), int(num_bits))
        return OK

    def moq_awq_weight_scale(self, w, rows, cols, g, dt, out, workspace, stream):
        if rows < 0 or cols <= 0 or g <= 0 or cols % g != 0:  # moq_reduce.hip: needs cols % g == 0
            self._err = b"moq_awq_weight_scale: needs cols % g == 0"
            return -1
        self.o.orc_awq_weight_scale(_vp(w), I64(rows), I64(cols), int(g), int(dt), _vp(out))
        return OK

    def moq_scale_cols_multi(self, w, s, y, rows, cols, n_scales, dt, stream):
        elem = 4 if dt == 0 else 2
        for a in range(int(n_scales)):
            self.o.orc_scale_cols(_vp(w), ctypes.c_void_p(_addr(s) + a * cols * 4),
                                  ctypes.c_void_p(_addr(y) + a * rows * cols * elem), I64(rows), I64(cols), int(dt))
        return OK

    def moq_awq_err_gemm_workspace(self, tokens, cout):
        return 0

    def moq_awq_err_gemm_multi(self, x, w, out_actual, bias, tokens, cout, cin, dt, n_cand, x_stride, w_stride, partial,
                               loss_acc, stream):
        elem = 4 if dt == 0 else 2
        self.o.orc_awq_err_gemm.restype = ctypes.c_double
        acc = _f32_view(loss_acc, n_cand)
        for a in range(int(n_cand)):
            v = self.o.orc_awq_err_gemm(ctypes.c_void_p(_addr(x) + a * x_stride * elem),
                                        ctypes.c_void_p(_addr(w) + a * w_stride * elem), _vp(out_actual), _vp(bias), None,
                                        I64(tokens), I64(cout), I64(cin), int(dt))
            acc[a] += np.float32(v)
        return OK

    def moq_awq_clip_loss(self, x, n_tok, x_row_stride, w, cout, cin, g, dt, amax, amax_dt, shrinks, n_shrink, num_bits,
                          loss, stream):
        """The oracle takes contiguous token rows and writes [K, cout, nblk]; the C-ABI strides over tokens and
        accumulates into [K, nblk, cout]."""
        xa = self._as_2d(x, 1, (n_tok - 1) * x_row_stride + cin, dt)[0]
        rows = np.ascontiguousarray(np.stack([xa[t * x_row_stride:t * x_row_stride + cin] for t in range(int(n_tok))]))
        nblk = (cin + g - 1) // g
        tmp = np.zeros((int(n_shrink), int(cout), int(nblk)), dtype=np.float32)
        self.o.orc_awq_clip_loss(oracle._p(rows), I64(n_tok), _vp(w), I64(cout), I64(cin), int(g), int(dt), _vp(amax),
                                 int(amax_dt), _vp(shrinks), int(n_shrink), int(num_bits), oracle._p(tmp))
        out = _f32_view(loss, n_shrink * nblk * cout).reshape(int(n_shrink), int(nblk), int(cout))
        out += tmp.transpose(0, 2, 1)
        return OK

    def moq_sgpt_block_sweep(self, w, rows, ld, i1, bs, hinv, delta, prune_n, prune_m, stream):
        self.o.orc_sgpt_block_sweep(_vp(w), I64(rows), I64(ld), I64(i1), int(bs), _vp(hinv), _vp(delta), int(prune_n),
                                    int(prune_m))
        return OK

    def moq_gptq_block_sweep(self, w, rows, ld, i1, bs, hinv, delta, amax, amax_row_stride, g, fmt, num_bits, is_unsigned,
                             narrow, stream):
        self.o.orc_gptq_block_sweep(_vp(w), I64(rows), I64(ld), I64(i1), int(bs), _vp(hinv), _vp(delta), _vp(amax),
                                    I64(amax_row_stride), I64(g), int(fmt), int(num_bits), int(is_unsigned), int(narrow))
        return OK

    # -- Gram-matrix AWQ search / SparseGPT Hessian: numpy restatements (fp64 accumulation) of the MFMA entries
    def moq_sgpt_trailing_update(self, w, rows, ld, i1, bs, delta, hinv, stream):
        self.o.orc_sgpt_trailing_update(_vp(w), I64(rows), I64(ld), I64(i1), int(bs), _vp(delta), _vp(hinv))
        return OK

    @staticmethod
    def _bf16_round(f32):
        u = np.ascontiguousarray(f32, dtype=np.float32).view(np.uint32)
        r = ((u >> 16) & 1) + np.uint32(0x7FFF)
        return ((u + r) >> 16).astype(np.uint16)

    def moq_transpose16_ld(self, x, y, rows, cols, y_ld, stream):
        xa = np.ctypeslib.as_array(ctypes.cast(_addr(x), ctypes.POINTER(ctypes.c_uint16)), shape=(int(rows), int(cols)))
        ya = np.ctypeslib.as_array(ctypes.cast(_addr(y), ctypes.POINTER(ctypes.c_uint16)),
                                   shape=((int(cols) - 1) * int(y_ld) + int(rows),))
        for c in range(int(cols)):
            ya[c * y_ld:c * y_ld + rows] = xa[:, c]
        return OK

    def moq_transpose16(self, x, y, rows, cols, stream):
        return self.moq_transpose16_ld(x, y, rows, cols, rows, stream)

    def moq_hessian_accum(self, xt, cin, tokens, dt, hessian, decay, scale, upper_only, stream):
        x = self._to_f32(self._as_2d(xt, cin, tokens, dt), dt).astype(np.float64)
        h = _f32_view(hessian, cin * cin).reshape(int(cin), int(cin))
        h[:] = (h.astype(np.float64) * float(decay) + float(scale) * (x @ x.T)).astype(np.float32)
        return OK

    def moq_symmetrize(self, h, n, stream):
        a = _f32_view(h, n * n).reshape(int(n), int(n))
        iu = np.triu_indices(int(n), 1)
        a[(iu[1], iu[0])] = a[iu]
        return OK

    def moq_awq_err_weight(self, w, s, r, e_out, a_out, rows, cols, g, dt, num_bits, planes, stream):
        y = np.empty((int(rows), int(cols)), dtype=np.float32 if dt == 0 else np.uint16)
        self.o.orc_awq_scale_qdq(_vp(w), _vp(s), oracle._p(y), I64(rows), I64(cols), int(g), int(dt), int(num_bits))
        wa = self._as_2d(w, rows, cols, dt)
        wf = wa.astype(np.float32) if dt == 0 else self._to_f32(wa, dt)
        yf = y if dt == 0 else self._to_f32(y, dt)
        e = (yf * _f32_view(r, cols)[None, :] - wf).astype(np.float32)
        _f32_view(e_out, rows * cols).reshape(int(rows), int(cols))[:] = e
        hi = self._bf16_round(e)
        lo = self._bf16_round(e - self._to_f32(hi, 2))
        ao = np.ctypeslib.as_array(ctypes.cast(_addr(a_out), ctypes.POINTER(ctypes.c_uint16)),
                                   shape=(int(rows), int(planes) * int(cols)))
        for i, plane in enumerate({1: [hi], 2: [hi, lo], 3: [hi, hi, lo]}[int(planes)]):
            ao[:, i * cols:(i + 1) * cols] = plane
        return OK

    def moq_awq_quadform(self, a, b, ref, rows, cols, k, dt, partial, loss_acc, inv_count, stream):
        af = self._to_f32(self._as_2d(a, rows, k, dt), dt).astype(np.float64)
        bf = self._to_f32(self._as_2d(b, cols, k, dt), dt).astype(np.float64)
        rf = _f32_view(ref, rows * cols).reshape(int(rows), int(cols)).astype(np.float64)
        _f32_view(loss_acc, 1)[0] += np.float32(float(inv_count) * float(((af @ bf.T) * rf).sum()))
        return OK

    def moq_mask_2to4(self, w, rows, cols, dt, mask, stream):
        self.o.orc_mask_2to4(_vp(w), I64(rows), I64(cols), int(dt), _vp(mask))
        return OK

    # -- packers
    def moq_int4_pack(self, x, scales, out, n, g, dt, rounding, stream):
        self.o.orc_int4_pack(_vp(x), _vp(scales), _vp(out), I64(n), int(g), int(dt), int(rounding))
        return OK

    def moq_int4_unpack(self, q, scales, out, n_bytes, g, dt, stream):
        self.o.orc_int4_unpack(_vp(q), _vp(scales), _vp(out), I64(n_bytes), int(g), int(dt))
        return OK

    def moq_int4_pack_export(self, w, wsf, out, rows, cols, g, dt, stream):
        self.o.orc_int4_pack_export(_vp(w), _vp(wsf), _vp(out), I64(rows), I64(cols), int(g), int(dt))
        return OK

    def moq_fp8_pack(self, x, scales, scale_dt, out, n, dt, mode, axis_size, inner, stream):
        self.o.orc_fp8_pack(_vp(x), _vp(scales), int(scale_dt), _vp(out), I64(n), int(dt), int(mode), I64(axis_size),
                            I64(inner))
        return OK

    def moq_fp8_unpack(self, q, scales, out, n, dt, mode, axis_size, inner, stream):
        self.o.orc_fp8_unpack(_vp(q), _vp(scales), _vp(out), I64(n), int(dt), int(mode), I64(axis_size), I64(inner))
        return OK

    @staticmethod
    def _as_2d(p, rows, cols, dt):
        ctype = ctypes.c_float if dt == 0 else ctypes.c_uint16
        return np.ctypeslib.as_array(ctypes.cast(_addr(p), ctypes.POINTER(ctype)), shape=(int(rows), int(cols)))

    @staticmethod
    def _to_f32(a16, dt):
        if dt == 1:  # MOQ_F16
            return a16.view(np.float16).astype(np.float32)
        return (a16.astype(np.uint32) << 16).view(np.float32)  # MOQ_BF16

    def moq_fp8_pack_tile(self, x, scales, scale_dt, out, rows, cols, br, bc, dt, stream):
        """One oracle call (scalar scale) per tile; fp32 scales on a 16-bit tensor promote the quotient to fp32."""
        xa = self._as_2d(x, rows, cols, dt)
        oa = np.ctypeslib.as_array(ctypes.cast(_addr(out), ctypes.POINTER(ctypes.c_uint8)), shape=(int(rows), int(cols)))
        promote = scale_dt == 0 and dt != 0
        n_t = (rows // br) * (cols // bc)
        sa = (_f32_view(scales, n_t) if scale_dt == 0 else
              np.ctypeslib.as_array(ctypes.cast(_addr(scales), ctypes.POINTER(ctypes.c_uint16)), shape=(n_t,)))
        for i in range(rows // br):
            for j in range(cols // bc):
                tile = np.ascontiguousarray(xa[i * br:(i + 1) * br, j * bc:(j + 1) * bc])
                tdt = dt
                if promote:
                    tile, tdt = np.ascontiguousarray(self._to_f32(tile, dt)), 0
                sc = np.ascontiguousarray(sa[i * (cols // bc) + j:i * (cols // bc) + j + 1])
                res = np.empty(tile.size, dtype=np.uint8)
                self.o.orc_fp8_pack(oracle._p(tile), oracle._p(sc), int(0 if promote else scale_dt), oracle._p(res),
                                    I64(tile.size), int(tdt), 0, I64(1), I64(1))
                oa[i * br:(i + 1) * br, j * bc:(j + 1) * bc] = res.reshape(br, bc)
        return OK

    def moq_fp8_unpack_tile(self, q, scales, out, rows, cols, br, bc, dt, stream):
        qa = np.ctypeslib.as_array(ctypes.cast(_addr(q), ctypes.POINTER(ctypes.c_uint8)), shape=(int(rows), int(cols)))
        oa = self._as_2d(out, rows, cols, dt)
        n_t = (rows // br) * (cols // bc)
        sa = self._as_2d(scales, 1, n_t, dt)[0]
        for i in range(rows // br):
            for j in range(cols // bc):
                tile = np.ascontiguousarray(qa[i * br:(i + 1) * br, j * bc:(j + 1) * bc])
                sc = np.ascontiguousarray(sa[i * (cols // bc) + j:i * (cols // bc) + j + 1])
                res = np.empty((br, bc), dtype=oa.dtype)
                self.o.orc_fp8_unpack(oracle._p(tile), oracle._p(sc), oracle._p(res), I64(tile.size), int(dt), 0, I64(1), I64(1))
                oa[i * br:(i + 1) * br, j * bc:(j + 1) * bc] = res
        return OK

    def moq_mxfp4_pack(self, x, packed, e8m0, n_blocks, block, dt, stream):
        self.o.orc_mxfp4_pack(_vp(x), _vp(packed), _vp(e8m0), I64(n_blocks), int(block), int(dt))
        return OK

    # -- segment tables (multi_tensor.SegmentTable): host arrays of moq_seg / moq_fold_seg rows, walked here row by row
    @staticmethod
    def _rows(p, n, width):
        return np.ctypeslib.as_array(ctypes.cast(_addr(p), ctypes.POINTER(ctypes.c_int64)), shape=(int(n), width))

    def moq_mt_fold_mx_fused(self, segs, blk, side, n_seg, n_chunks, block, dt, fmt, stream):
        elem = 4 if dt == 0 else 2
        for (x, y, _, n), (scale, cols, _) in zip(self._rows(segs, n_seg, 4), self._rows(side, n_seg, 3)):
            rows = int(n) // int(cols)
            src = int(x)
            if scale:
                tmp = np.empty(int(n) * elem, dtype=np.uint8)
                self.o.orc_scale_cols(_vp(int(x)), _vp(int(scale)), oracle._p(tmp), I64(rows), I64(int(cols)), int(dt))
                src = tmp.ctypes.data
            self.o.orc_mx_fused_amax_convert2(_vp(src), _vp(int(y)), I64(rows), I64(int(cols)), int(block), int(dt), int(fmt),
                                              oracle.MX_TYPES["E8M0"], _vp(None))
        return OK

    def moq_mt_fold_mxfp4_pack(self, segs, blk, side, n_seg, n_chunks, block, dt, stream):
        elem = 4 if dt == 0 else 2
        for (x, y, _, n), (scale, cols, e8) in zip(self._rows(segs, n_seg, 4), self._rows(side, n_seg, 3)):
            src = int(x)
            if scale:
                tmp = np.empty(int(n) * elem, dtype=np.uint8)
                self.o.orc_scale_cols(_vp(int(x)), _vp(int(scale)), oracle._p(tmp), I64(int(n) // int(cols)), I64(int(cols)), int(dt))
                src = tmp.ctypes.data
            self.o.orc_mxfp4_pack(_vp(src), _vp(int(y)), _vp(int(e8)), I64(int(n) // int(block)), int(block), int(dt))
        return OK

    def moq_mxfp4_unpack(self, packed, e8m0, out, n_blocks, block, dt, stream):
        self.o.orc_mxfp4_unpack(_vp(packed), _vp(e8m0), _vp(out), I64(n_blocks), int(block), int(dt))
        return OK


def install(monkeypatch, moa):
    """Route the package's C-ABI calls to the oracle for the duration of a test and let CPU tensors through."""
    from model_optimizer_amd import _lib, ops

    fake = HostMemLib()
    monkeypatch.setattr(_lib, "lib", lambda: fake)
    monkeypatch.setattr(ops, "_require_gpu", lambda t, what: None)

    @contextmanager
    def _on(t):
        yield None

    monkeypatch.setattr(ops, "_on", _on)
    monkeypatch.setattr(ops, "_is_gpu", lambda t: True)
    from model_optimizer_amd import multi_tensor

    monkeypatch.setattr(multi_tensor, "_on", _on)  # (bound by name at import)
    monkeypatch.setattr(multi_tensor, "_require_gpu", lambda t, what: None)
    return fake
