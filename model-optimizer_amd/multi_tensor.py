"""Whole-layer / whole-model weight passes in ONE launch (the multi-tensor forms of the C-ABI).

The reference walks the model and calls one kernel per weight (weight_only_quantize,
quantization/model_calib.py:187-199).  At HBM speed an 8-100 MB tensor takes 2-20 us, the same order as a
launch gap, so here the tensors of a layer / model are described by a device-resident segment table and
calibrated / quantize-dequantized by a single grid.
"""

from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import MoquantError, check
from .ops import _dt, _on, _p, _require_gpu


class SegmentTable:
    """Device-side table of (x, y, amax, n) segments + chunk prefix sums.

    inputs      : list of contiguous GPU tensors of one dtype
    outputs     : list of same-shaped tensors (default: new tensors; pass inputs for in place)
    group_size  : None -> one fp32 amax per tensor; g -> n/g fp32 amax per tensor (per-group formats)
    """

    def __init__(self, inputs, outputs=None, group_size: int | None = None):
        if not inputs:
            raise MoquantError("SegmentTable needs at least one tensor")
        for t in inputs:
            _require_gpu(t, "SegmentTable")
            if not t.is_contiguous():
                raise MoquantError("SegmentTable tensors must be contiguous")
            if t.dtype != inputs[0].dtype or t.device != inputs[0].device:
                raise MoquantError("SegmentTable tensors must share dtype and device")
        self.inputs = list(inputs)
        self.outputs = [torch.empty_like(t) for t in inputs] if outputs is None else list(outputs)
        self.group_size = group_size
        self.device = inputs[0].device
        self.dtype_code = _dt(inputs[0])
        n_seg = len(inputs)
        sizes = [t.numel() for t in inputs]
        if group_size is None:
            self.amax_flat = torch.zeros(n_seg, dtype=torch.float32, device=self.device)
            offs = list(range(n_seg))
            self.amax = [self.amax_flat[i:i + 1] for i in range(n_seg)]
        else:
            if any(s % group_size for s in sizes):
                raise MoquantError("every tensor's numel must be a multiple of group_size")
            counts = [s // group_size for s in sizes]
            self.amax_flat = torch.zeros(sum(counts), dtype=torch.float32, device=self.device)
            offs, acc = [], 0
            for c in counts:
                offs.append(acc)
                acc += c
            self.amax = [self.amax_flat[o:o + c] for o, c in zip(offs, counts)]
        # host plan -> device table
        n_arr = (ctypes.c_int64 * n_seg)(*sizes)
        blk = (ctypes.c_int64 * (n_seg + 1))()
        total = _lib.lib().moq_mt_plan(n_arr, n_seg, blk)
        if total < 0:
            check(int(total))
        self.n_chunks = int(total)
        self.n_seg = n_seg
        rows = []
        base = self.amax_flat.data_ptr()
        for i, (x, y) in enumerate(zip(self.inputs, self.outputs)):
            rows.append([x.data_ptr(), y.data_ptr(), base + 4 * offs[i], sizes[i]])
        # pointers are < 2^63 so int64 storage is lossless; layout == struct moq_seg (4 x 8 bytes)
        self._segs = torch.tensor(rows, dtype=torch.int64).to(self.device)
        self._blk = torch.tensor(list(blk), dtype=torch.int64).to(self.device)
        self.bytes_in = sum(s * inputs[0].element_size() for s in sizes)

    # -- a1 over the table
    def calibrate_amax(self, atomic: bool = False):
        """Per-tensor abs-max of every tensor of the table.  Default: two-stage (per-chunk values in a scratch buffer,
        then one fold per tensor; the faster sweep order); atomic=True: the single-launch atomicMax form."""
        if self.group_size is not None:
            raise MoquantError("calibrate_amax is the per-tensor pass (table built with group_size)")
        with _on(self._segs) as stream:
            if atomic:
                check(_lib.lib().moq_mt_amax(_p(self._segs), _p(self._blk), self.n_seg, self.n_chunks,
                                             self.dtype_code, stream))
            else:
                if getattr(self, "_scratch", None) is None:
                    self._scratch = torch.empty(max(self.n_chunks, 1), dtype=torch.float32, device=self.device)
                check(_lib.lib().moq_mt_amax_ws(_p(self._segs), _p(self._blk), self.n_seg, self.n_chunks,
                                                self.dtype_code, _p(self._scratch), stream))
        return self.amax_flat

    # -- a7 over the table (uses the per-tensor amax slots)
    def fake_quant_e4m3(self):
        with _on(self._segs) as stream:
            check(_lib.lib().moq_mt_fake_quant_e4m3(_p(self._segs), _p(self._blk), self.n_seg, self.n_chunks,
                                                    self.dtype_code, stream))
        return self.outputs

    # -- a6 over the table (per-tensor amax)
    def fake_quant_int(self, num_bits=8, unsigned=False, narrow_range=True):
        with _on(self._segs) as stream:
            check(_lib.lib().moq_mt_fake_quant_int(_p(self._segs), _p(self._blk), self.n_seg, self.n_chunks,
                                                   self.dtype_code, int(num_bits), int(unsigned),
                                                   int(narrow_range), stream))
        return self.outputs

    # -- fused per-group amax + QDQ over the table
    def amax_qdq_int_group(self, num_bits=4, unsigned=False, narrow_range=False):
        if self.group_size is None:
            raise MoquantError("table was built without group_size")
        with _on(self._segs) as stream:
            check(_lib.lib().moq_mt_amax_qdq_int_group(_p(self._segs), _p(self._blk), self.n_seg,
                                                       self.n_chunks, int(self.group_size), self.dtype_code,
                                                       int(num_bits), int(unsigned), int(narrow_range),
                                                       stream))
        return self.outputs

    # -- a8 over the table (dynamic MX blocks along the flattened tensors; every numel % block == 0)
    def mx_fused_amax_convert(self, block_size: int = 32, fmt: str = "E2M1"):
        if any(t.numel() % block_size or t.shape[-1] % block_size for t in self.inputs):
            raise MoquantError("every tensor's last dim must be a multiple of the MX block size")
        if any(t.data_ptr() % 16 for t in self.inputs) or any(t.data_ptr() % 16 for t in self.outputs):
            raise MoquantError("multi-tensor MX needs 16-byte aligned tensors")
        with _on(self._segs) as stream:
            check(_lib.lib().moq_mt_mx_fused_amax_convert(_p(self._segs), _p(self._blk), self.n_seg, self.n_chunks,
                                                          int(block_size), self.dtype_code, _lib.MX_TYPES[fmt], stream))
        return self.outputs

    # -- SmoothQuant's fold composed with a8 / the MXFP4 packer: ONE read of every weight (moq_mt_fold_mx_fused / _pack)
    def fold_side(self, scales, block_size, e8m0=None):
        """Device table of moq_fold_seg rows (scale pointer, cols, e8m0 pointer) beside the segment table.  scales[i]: the
        fp32 column vector of tensor i (its last dim) or None (no fold for that tensor)."""
        if len(scales) != self.n_seg:
            raise MoquantError("one scale vector (or None) per tensor")
        rows, keep = [], []
        for i, (t, s) in enumerate(zip(self.inputs, scales)):
            cols = t.shape[-1]
            if cols % block_size or t.numel() % block_size:
                raise MoquantError("every tensor's last dim must be a multiple of the MX block size")
            if t.data_ptr() % 16:
                raise MoquantError("multi-tensor MX needs 16-byte aligned tensors")
            ptr = 0
            if s is not None:
                s = s.detach().reshape(-1)
                if s.dtype != torch.float32 or s.device != self.device or not s.is_contiguous():
                    s = s.to(device=self.device, dtype=torch.float32).contiguous()
                if s.numel() != cols:
                    raise MoquantError("scale length must equal the tensor's last dim")
                if s.data_ptr() % 16:
                    s = s.clone()
                keep.append(s)
                ptr = s.data_ptr()
            rows.append([ptr, cols, 0 if e8m0 is None else e8m0[i].data_ptr()])
        side = torch.tensor(rows, dtype=torch.int64).to(self.device)  # layout == struct moq_fold_seg (3 x 8 bytes)
        return side, keep  # (a caller that repeats the launch keeps this pair and passes it as `side=`)

    def fold_mx_fused(self, scales=None, block_size: int = 32, fmt: str = "E2M1", side=None):
        """outputs[i] = MXQDQ(dt(inputs[i] * scales[i][None, :])) for every tensor in ONE launch -- _apply_weight_pre_quant_scale
        (model_calib.py:1208-1216) followed by the MX weight quantizer's forward, one read and one write per element;
        bit-equal to ops.scale_cols + mx_fused_amax_convert."""
        if any(t.data_ptr() % 16 for t in self.outputs):
            raise MoquantError("multi-tensor MX needs 16-byte aligned tensors")
        table, keep = side if side is not None else self.fold_side(scales, block_size)
        with _on(self._segs) as stream:
            check(_lib.lib().moq_mt_fold_mx_fused(_p(self._segs), _p(self._blk), _p(table), self.n_seg, self.n_chunks,
                                                  int(block_size), self.dtype_code, _lib.MX_TYPES[fmt], stream))
        del keep
        return self.outputs

    def fold_mxfp4_pack(self, scales, block_size: int = 32):
        """MXFP4QTensor.quantize(dt(inputs[i] * scales[i][None, :])) for every tensor in ONE launch: `outputs` must be the
        packed uint8 tensors [..., K / 2]; returns (outputs, [e8m0 uint8 [n / block, 1] per tensor])."""
        for x, y in zip(self.inputs, self.outputs):
            if y.dtype != torch.uint8 or y.numel() * 2 != x.numel() or not y.is_contiguous() or y.data_ptr() % 4:
                raise MoquantError("fold_mxfp4_pack: outputs must be contiguous uint8 tensors of half the inputs' elements")
        e8 = [torch.empty(x.numel() // block_size, 1, dtype=torch.uint8, device=self.device) for x in self.inputs]
        side, keep = self.fold_side(scales, block_size, e8m0=e8)
        with _on(self._segs) as stream:
            check(_lib.lib().moq_mt_fold_mxfp4_pack(_p(self._segs), _p(self._blk), _p(side), self.n_seg, self.n_chunks,
                                                    int(block_size), self.dtype_code, stream))
        del keep
        return self.outputs, e8

    # -- a14 over the table: outputs must be uint8 / bool tensors of the inputs' shapes
    def mask_2to4(self):
        for x, m in zip(self.inputs, self.outputs):
            if m.element_size() != 1 or m.numel() != x.numel() or x.shape[-1] % 4:
                raise MoquantError("mask_2to4: outputs must be 1-byte masks of the inputs' shapes (last dim % 4 == 0)")
        with _on(self._segs) as stream:
            check(_lib.lib().moq_mt_mask_2to4(_p(self._segs), _p(self._blk), self.n_seg, self.n_chunks,
                                              self.dtype_code, stream))
        return self.outputs

    # -- a14 + the multiply that follows it (+ a1 of the result): mask, masked weight in place, optionally its abs-max
    def mask_2to4_apply(self, calibrate: bool = False):
        """2:4 magnitude masks into `outputs` AND `inputs` rewritten in place as dtype(w * mask), one pass over the
        tensors (sparsify's mask + `weight.mul_(mask)`).  calibrate=True: `amax_flat[i]` additionally becomes the abs-max of
        the masked tensor i -- the statistic a per-tensor max calibration of the sparsified model starts with."""
        if self.group_size is not None:
            raise MoquantError("mask_2to4_apply is a per-tensor pass (table built with group_size)")
        for x, m in zip(self.inputs, self.outputs):
            if m.element_size() != 1 or m.numel() != x.numel() or x.shape[-1] % 4 or m.data_ptr() == x.data_ptr():
                raise MoquantError("mask_2to4_apply: outputs must be 1-byte masks of the inputs' shapes (last dim % 4 == 0)")
        scratch = None
        if calibrate:
            if getattr(self, "_scratch", None) is None:
                self._scratch = torch.empty(max(self.n_chunks, 1), dtype=torch.float32, device=self.device)
            scratch = self._scratch
        with _on(self._segs) as stream:
            check(_lib.lib().moq_mt_mask_2to4_apply(_p(self._segs), _p(self._blk), self.n_seg, self.n_chunks,
                                                    self.dtype_code, _p(scratch), stream))
        return self.outputs
