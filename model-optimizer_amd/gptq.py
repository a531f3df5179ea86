"""GPTQ weight update -- the `gptq` calibration algorithm of the path (quantization/model_calib.py:2192-2271 and
quantization/utils/calib_utils.py:50-276: update_hessian, compute_hessian_inverse, GPTQHelper, gptq_blockwise_update).

Per quantized linear: the Hessian of its inputs is accumulated during one forward loop (weight quantizers off), the
damped inverse is factored (upper Cholesky of H^-1), and the columns are quantized one block at a time, each column's
rounding error pushed onto the columns to its right through that factor.

What is MI355X-native here:
  * Hessian: 16-bit activations are transposed and contracted on the matrix cores (ops.hessian_accum -- the SYRK of
    the SparseGPT path, upper tiles only, mirrored once); linears fed by the SAME tensor (q / k / v, gate / up) share
    one Hessian and, when their dead columns agree, one inverse factor.
  * Column sweep: the reference fake-quantizes the WHOLE weight matrix once per column and keeps one column of it.
    With a calibrated amax the quantizer is elementwise, so the sweep over a block is one kernel
    (ops.gptq_block_sweep: a wave per row, the block's columns in registers); the update of the columns right of the
    block is ops.sgpt_trailing_update (fp32 matrix cores, defined summation order).  MX formats (dynamic blocks with
    E8M0 scales: MXFP4, MXFP8 -- the 4-bit format MI355X multiplies natively) are NOT elementwise: the scale of a block
    follows the weights as they are updated, which is what the reference's full-matrix call is for; the same kernel
    recomputes the pivot block's abs-max from the block's lanes at every column.  Two-level block scales (an element
    format as the scale format, relative to the calibrated tensor-wide amax: the NVFP4-style formats) ride the same
    kernel; only a quantizer without such a calibrated amax takes the reference's own loop through the quantizer.
  * Data-parallel replicas (distributed.declare_data_parallel): Hessians are combined sample-weighted on one owner
    rank each, the owner updates the linears that read them and broadcasts the weights -- every replica ends with the
    same model (the reference leaves ranks to diverge).
"""

from __future__ import annotations

import time
import warnings

import torch
from torch import nn

from . import ops
from .nn import is_quantized_linear
from .sparsity import HessianState
from .tensor_quantizer import TensorQuantizer

GPTQ_STATS: dict = {}


class TokenHessian(HessianState):
    """update_hessian (calib_utils.py:50-77): the running mean counts TOKENS (rows of the flattened input), where
    SparseGPT's hook counts batches: H <- H * n / (n + b) + (2 / (n + b)) X^T X.  (The reference scales the inputs by
    sqrt(2 / n) before the product; the scale is applied to the fp32 accumulators here -- the same sum up to rounding,
    and a GEMM's summation order is the library's in the reference anyway.)
    The [Cin, Cin] matrix is allocated by the first update: linears that share another one's Hessian never own one
    (Llama-3-70B: 2.1 GB per layer instead of 4.1)."""

    def __init__(self, cols: int, device):
        self._cols, self._device = cols, device
        self._h = None
        self._upper = False
        self.samples = 0

    @property
    def shape(self):
        return (self._cols, self._cols)

    def _ensure(self):
        if self._h is None:
            try:
                self._h = torch.zeros(self._cols, self._cols, dtype=torch.float32, device=self._device)
            except torch.cuda.OutOfMemoryError as e:
                raise torch.cuda.OutOfMemoryError(
                    f"gptq: no room for a {self._cols} x {self._cols} fp32 Hessian -- the Hessians of ALL linears are alive "
                    "in a whole-model pass; use algorithm {'method': 'gptq', 'layerwise': {'enable': True}} (one decoder "
                    "layer's Hessians at a time)") from e

    @torch.no_grad()
    def update(self, inp: torch.Tensor):
        x2 = inp.reshape(-1, inp.shape[-1])
        b = x2.shape[0]
        if b == 0:  # in MoEs some experts receive no tokens
            return
        self._ensure()
        decay = self.samples / (self.samples + b)
        self.samples += b
        scale = 2.0 / self.samples
        if x2.dtype in (torch.bfloat16, torch.float16) and x2.shape[1] % 4 == 0 and x2.is_cuda:
            ops.hessian_accum(self._h, x2, decay, scale, upper_only=True)
            self._upper = True
        else:
            xf = x2.float()
            self.hessian.mul_(decay).addmm_(xf.t(), xf, alpha=scale)


def update_hessian(input: torch.Tensor, hessian: torch.Tensor, n_samples: int):
    """Functional form of calib_utils.update_hessian: returns (hessian, n_samples); `hessian` is updated in place."""
    st = TokenHessian(hessian.shape[0], hessian.device)
    st._h, st.samples = hessian, int(n_samples)
    st.update(input)
    return st.hessian, st.samples


def dead_columns(weight: torch.Tensor) -> torch.Tensor:
    """Columns of the weight that are zero in every row (calib_utils.py:97): bool [Cin]."""
    return weight.eq(0).all(dim=0)


def compute_hessian_inverse(hessian: torch.Tensor, weight: torch.Tensor | None, perc_damp: float,
                            zero_cols: torch.Tensor | None = None) -> torch.Tensor:
    """calib_utils.py:80-113: dead-neuron columns of `weight` are cut out of the Hessian (row and column zeroed, unit
    diagonal), the diagonal is damped by perc_damp * mean(diag), and the upper Cholesky factor of the inverse is
    returned; a Hessian that is not positive definite gives the identity (with the reference's warning)."""
    h = hessian.clone()
    zero = zero_cols if zero_cols is not None else dead_columns(weight)
    if bool(zero.any()):
        h[zero, :] = 0
        h[:, zero] = 0
        idx = torch.nonzero(zero).flatten()
        h[idx, idx] = 1
    damp = perc_damp * torch.mean(torch.diag(h))
    diag = torch.arange(h.shape[0], device=h.device)
    h[diag, diag] += damp
    try:
        h = torch.cholesky_inverse(torch.linalg.cholesky(h))
        return torch.linalg.cholesky(h, upper=True).contiguous()
    except (RuntimeError, torch.linalg.LinAlgError):
        warnings.warn("Warning: Hessian is not positive definite, using identity matrix")
        return torch.eye(h.shape[0], device=h.device, dtype=h.dtype)


def _static_layout(q, weight: torch.Tensor):
    """(fmt, num_bits, unsigned, narrow, amax fp32 flat, amax_row_stride, g) when `q` is a TensorQuantizer whose
    quantize-dequantize of a 2-D fp32 weight is ELEMENTWISE with a calibrated amax -- what the sweep kernel takes --,
    else None (the quantizer's own forward is used column by column, like the reference)."""
    if not isinstance(q, TensorQuantizer) or weight.dim() != 2:
        return None
    if q._disabled or not q.fake_quant or q._dynamic or q.pre_quant_scale is not None:
        return None
    if getattr(q, "_bias", None) is not None:
        return None  # an affine offset: QDQ(w - bias) + bias is the quantizer's own forward, column by column
    if q.is_mx_format:
        # dynamic blocks with E8M0 scales (MXFP4 / MXFP8 / ...): the block scale follows the CURRENT weights, which the
        # sweep kernel recomputes per column from the block's lanes (fmt 3); blocks along the last dim only
        nb = q._num_bits if not isinstance(q._num_bits, list) else tuple(q._num_bits)
        fmt = {(2, 1): "E2M1", (4, 3): "E4M3", (5, 2): "E5M2", (3, 2): "E3M2", (2, 3): "E2M3", 8: "INT8"}.get(nb)
        bsz = q._block_sizes
        g = bsz.get(-1, None) or bsz.get(1, None)
        if (fmt is None or not g or set(bsz) - {-1, 1, "type", "scale_bits"} or g > 64 or g & (g - 1)
                or weight.shape[1] % g):
            return None
        return 3, fmt, False, False, None, 0, int(g)
    if q._block_dynamic:
        # block scales in an element format relative to the tensor-wide amax (NVFP4-style two-level scales): row-local once
        # that amax is calibrated (fmt 4); a tensor-wide amax taken from the current weights would couple the rows
        amax = getattr(q, "_amax", None)
        nb = q._num_bits if not isinstance(q._num_bits, list) else tuple(q._num_bits)
        names = {(2, 1): "E2M1", (4, 3): "E4M3", (5, 2): "E5M2", (3, 2): "E3M2", (2, 3): "E2M3", 8: "INT8"}
        sb = q._block_sizes.get("scale_bits", None)
        fmt, sfmt = names.get(nb), names.get(tuple(sb) if isinstance(sb, (list, tuple)) else sb)
        bsz = q._block_sizes
        g = bsz.get(-1, None) or bsz.get(1, None)
        if (amax is None or amax.numel() != 1 or fmt is None or sfmt is None or not g or g > 64 or g & (g - 1)
                or set(bsz) - {-1, 1, "type", "scale_bits"} or weight.shape[1] % g):
            return None
        return 4, fmt, sfmt, False, amax.detach().float().reshape(1), 0, int(g)
    amax = getattr(q, "_amax", None)
    if amax is None:
        return None
    nb = q._num_bits if not isinstance(q._num_bits, list) else tuple(q._num_bits)
    if nb == (4, 3):
        fmt, bits = 2, 8
    elif isinstance(nb, int) and 2 <= nb <= 16:
        fmt, bits = 1, nb
    else:
        return None
    rows, cols = weight.shape
    am = amax.detach().float().reshape(-1)
    bsz = q._block_sizes
    if bsz is None:
        axis = q._axis
        if axis is None and am.numel() == 1:
            stride, g = 0, cols
        elif axis in (0, (0,), [0], -2, (-2,)) and am.numel() == rows:
            stride, g = 1, cols
        else:
            return None
    else:
        if not q.is_static_block_quant or set(bsz) - {-1, 1, "type"} or fmt != 1:
            return None
        g = bsz.get(-1, None) or bsz.get(1, None)
        if not g or cols % g or am.numel() != rows * (cols // g):
            return None
        stride = cols // g
    return fmt, bits, bool(q._unsigned), bool(q._narrow_range), am, int(stride), int(g)


@torch.no_grad()
def gptq_blockwise_update(weight: torch.Tensor, h_inv: torch.Tensor, block_size: int, quantize_fn) -> dict:
    """calib_utils.py:241-276 on the fp32 working copy `weight` [Cout, Cin], in place.  Returns {"kernel": bool}."""
    num_cols = weight.shape[1]
    layout = _static_layout(quantize_fn, weight)
    if (layout is not None and layout[0] in (3, 4) and block_size % layout[6]) or (layout is not None and block_size > 128):
        layout = None  # (an MX block must not straddle two column blocks of the update)
    if layout is not None and weight.dtype == torch.float32 and weight.is_contiguous():
        fmt, bits, unsigned, narrow, am, stride, g = layout
        h_inv = h_inv.float().contiguous()
        for i1 in range(0, num_cols, block_size):
            bs = min(block_size, num_cols - i1)
            errs = ops.gptq_block_sweep(weight, i1, bs, h_inv, am, stride, g, fmt, bits, unsigned, narrow)
            if i1 + bs < num_cols:
                ops.sgpt_trailing_update(weight, i1, errs, h_inv)
        return {"kernel": True}
    # No kernel for this quantizer: every column is quantized by running the quantizer over the WHOLE working matrix (its
    # scales may depend on any of it) and keeping that one column -- the semantics of calib_utils.py:241-276, column by column
    n_rows = weight.shape[0]
    for lo in range(0, num_cols, block_size):
        hi = min(lo + block_size, num_cols)
        factor = h_inv[lo:hi, lo:hi]
        work = weight.clone()  # running values of the block's columns (the others stay as they are)
        block_err = weight.new_zeros(n_rows, hi - lo)
        for k in range(hi - lo):
            col = lo + k
            before = work[:, col].clone()
            quantized_col = quantize_fn(work)[:, col]
            weight[:, col] = quantized_col
            e = (before - quantized_col) / factor[k, k]
            work[:, col:hi] -= torch.outer(e, factor[k, k:])  # product rounded, then subtracted
            block_err[:, k] = e
        if hi < num_cols:
            weight[:, hi:] -= block_err @ h_inv[lo:hi, hi:]
    return {"kernel": False}


def relative_mse(weight_new: torch.Tensor, weight_orig: torch.Tensor, hessian: torch.Tensor) -> float:
    """GPTQHelper._print_mse_error (calib_utils.py:232-238): Hessian-weighted relative error of the update."""
    delta = weight_new - weight_orig
    num = delta.mm(hessian).mul(delta).mean()
    den = weight_orig.mm(hessian).mul(weight_orig).mean() + 1e-6
    return float(num / den)


class GPTQHelper:
    """Per-module state of calib_utils.GPTQHelper: owns (or shares) the Hessian, hooks the module during the collection
    pass, runs the blockwise update."""

    def __init__(self, module: nn.Module, name: str):
        self.module, self.name = module, name
        self.state: TokenHessian | None = TokenHessian(module.weight.shape[-1], module.weight.device)
        self.owner: "GPTQHelper" = self  # the helper whose Hessian this linear reads
        self._handle = None

    # -- collection
    def setup(self, shared: dict):
        helper = self

        def pre_hook(mod, args, kwargs):
            x = args[0] if args else kwargs.get("input")  # (the reference's patched forward takes `input=` too)
            if x is None:
                raise RuntimeError(f"gptq: {helper.name} was called without an input tensor")
            x = x.to_local() if hasattr(x, "to_local") else x
            iq = getattr(mod, "input_quantizer", None)
            if iq is not None and iq.is_enabled:
                h_in = iq(x)
                key = None  # a quantized copy: nothing to share by identity
            else:
                h_in, key = x, (None if x.is_inference() else x)  # (inference tensors carry no version counter: no sharing)
            # the same tensor OBJECT in the same state (an in-place write in between makes it another input), and only a
            # helper that has no samples of its own may start following: its accumulated Hessian would be dropped
            first = shared.get("input") is key and key is not None and shared.get("version") == key._version
            if first and shared["owner"].state.shape == helper.state.shape and helper.owner in (helper, shared["owner"]):
                if helper.owner is helper and getattr(helper.state, "samples", 0):
                    raise RuntimeError(f"gptq: {helper.name} accumulated its own Hessian in earlier batches and now reads the "
                                       "tensor another linear has just read; the sharing structure must not change between batches")
                helper.owner = shared["owner"]
                return
            if helper.owner is not helper:
                raise RuntimeError("gptq: a linear that shared its input with another one in an earlier batch got a "
                                   "different tensor now")
            helper.state.update(h_in)
            shared["input"], shared["owner"], shared["version"] = key, helper, (key._version if key is not None else None)

        self._handle = self.module.register_forward_pre_hook(pre_hook, with_kwargs=True)

    def cleanup(self):
        if self._handle is not None:
            self._handle.remove()
            self._handle = None
        if self.owner is not self:
            self.state = None  # never written

    def free(self):
        self.state = None


def _weight_quantizers_off(model):
    qs = [m.weight_quantizer for m in model.modules() if is_quantized_linear(m) and isinstance(m.weight_quantizer, TensorQuantizer)]
    saved = [q._disabled for q in qs]
    for q in qs:
        q._disabled = True
    return qs, saved


@torch.no_grad()
def gptq(model: nn.Module, forward_loop, perc_damp: float = 0.01, block_size: int = 128, fused: bool = False,
         shard_weights: bool | None = None, report_mse: bool = True):
    """model_calib.gptq (model_calib.py:2192-2271).  `model` is the full model, or one decoder layer when called by
    layerwise.layerwise_calibrate.  Steps: max_calibrate (amax from the current activations / weights); Hessians of
    every quantized linear's input from ONE forward loop with the weight quantizers off (the input quantizer, when
    enabled, is applied to what the Hessian sees); blockwise weight update.  `fused` (the reference's Triton kernel for
    static NVFP4) has no meaning here: the elementwise formats are always one kernel per block.  `report_mse`: the
    Hessian-weighted relative error the reference prints per linear (two fp32 GEMMs against the Hessian each -- as much
    arithmetic as the update itself for wide linears) goes to GPTQ_STATS["relative_mse"]."""
    from . import distributed as mdist
    from . import model_calib
    from .sparsity import _combine_hessians

    t0 = time.perf_counter()
    model_calib.max_calibrate(model, forward_loop)
    layers = [(n, m) for n, m in model.named_modules()
              if is_quantized_linear(m) and isinstance(m.weight_quantizer, TensorQuantizer) and m.weight_quantizer.is_enabled]
    GPTQ_STATS.clear()
    GPTQ_STATS.update({"linears": len(layers), "kernel_linears": 0, "shared_hessians": 0, "relative_mse": {}})
    if not layers:
        return model
    helpers = {m: GPTQHelper(m, n) for n, m in layers}
    shared = {"input": None, "owner": None}
    for h in helpers.values():
        h.setup(shared)
    qs, saved = _weight_quantizers_off(model)
    try:
        forward_loop(model)
    finally:
        for q, d in zip(qs, saved):
            q._disabled = d
        for h in helpers.values():
            h.cleanup()
        shared["input"] = shared["owner"] = None
    GPTQ_STATS["shared_hessians"] = sum(1 for h in helpers.values() if h.owner is not h)
    shard = mdist.resolve_shard(shard_weights)
    placed = None
    mods = [m for _, m in layers]
    for h in helpers.values():
        if h.owner is h:
            h.state._ensure()  # (a linear the forward loop never reached -- or a rank without batches -- has a zero Hessian)
    if shard:
        owner_of = {m: helpers[m].owner.module for m in mods if helpers[m].owner is not helpers[m]}
        placed = _combine_hessians(mods, {m: helpers[m].owner.state for m in mods}, owner_of)
    inverse_cache: dict = {}
    users: dict = {}
    for m in mods:
        users[helpers[m].owner] = users.get(helpers[m].owner, 0) + 1
    t_upd = time.perf_counter()
    for name, m in layers:
        h = helpers[m]
        own = h.owner
        if placed is not None and placed[own.module] != placed["me"]:
            users[own] -= 1
            continue
        hessian = own.state.hessian
        w_orig = m.weight.data
        weight = w_orig.to(torch.float32, copy=True)
        zero = dead_columns(weight)
        key = (own, bytes(zero.cpu().numpy().tobytes()) if bool(zero.any()) else b"")
        if key not in inverse_cache:
            inverse_cache[key] = compute_hessian_inverse(hessian.to(weight.device), None, perc_damp, zero_cols=zero)
        info = gptq_blockwise_update(weight, inverse_cache[key], block_size, m.weight_quantizer)
        GPTQ_STATS["kernel_linears"] += int(info["kernel"])
        if report_mse:
            GPTQ_STATS["relative_mse"][name] = relative_mse(weight, w_orig.float(), hessian.to(weight.device))
        w_orig.copy_(weight.reshape(w_orig.shape))  # in place (the reference rebinds `.data`): views and tables of the weight stay valid
        users[own] -= 1
        if users[own] == 0:
            own.free()
            for k in [k for k in inverse_cache if k[0] is own]:
                del inverse_cache[k]
    if placed is not None:
        mdist.broadcast_from_owners([m.weight.data for m in mods], group=mdist.replica_group(),
                                    owners=[placed[helpers[m].owner.module] for m in mods])
    for h in helpers.values():
        h.free()
    now = time.perf_counter()
    GPTQ_STATS["update_s"] = round(now - t_upd, 4)
    GPTQ_STATS["total_s"] = round(now - t0, 4)
    return model
