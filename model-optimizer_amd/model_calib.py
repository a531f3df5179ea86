"""Calibration algorithms of the path -- mirrors of quantization/model_calib.py on our kernels:
max_calibrate (:310-498), smoothquant (:1273-1359), awq_lite (:1394-1721) and their helpers.

Host-side orchestration is the reference's; the passes over tensor data are HIP kernels:
  weights            -> ONE multi-tensor launch per pass for all linears that share a format
  activations        -> fused column statistics (sum |x| + abs-max in one read)
  AWQ search weights -> fused (W * s) -> group amax -> INT-k QDQ
  cross-rank sync    -> one bucketed collective per reduce op (distributed.py)
"""

from __future__ import annotations

import math
import time
import warnings

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from . import distributed as mdist
from . import numerics
from . import ops
from .multi_tensor import SegmentTable
from .hf_experts import is_quant_fused_experts
from .nn import QuantLinear, is_quantized_linear
from ._lib import MoquantUnsupported
from .tensor_quantizer import SequentialQuantizer, TensorQuantizer


def _quantizers(model):
    return [m for m in model.modules() if isinstance(m, TensorQuantizer)]


def _dist_on() -> bool:
    return mdist.active()


def enable_stats_collection(model: nn.Module, distributed_sync: bool = False):
    """model_calib.py:1128-1141.  Every enabled quantizer stops quantizing while statistics are collected -- the
    dynamic ones too (their `enable_calib` is a no-op), so that the static quantizers downstream of them calibrate on
    clean activations; a quantizer without a calibrator is switched off altogether (:1140-1141).

    distributed_sync (data-parallel calibration): histogram calibrators take the bin width of rank 0's first batch
    (one broadcast at their first collect), so that the ranks' counts can be summed exactly at the end."""
    for q in _quantizers(model):
        if not q.is_enabled:
            continue
        if q._use_constant_amax or q._constant_amax is not None:
            q.disable_quant()  # nothing to calibrate; no quantization error into the other quantizers' statistics (:1132-1136)
            continue
        if q._calibrator is not None:
            q.disable_quant()
            q.enable_calib()
            if hasattr(q._calibrator, "share_range_across_ranks"):
                q._calibrator.share_range_across_ranks(distributed_sync and _dist_on())
        else:
            q.disable()


def finish_stats_collection(model: nn.Module, method: str | None = None, distributed_sync: bool = False, **kwargs):
    """model_calib.py:1144-1167: load_calib_amax on every calibrated quantizer, back to quant mode.

    distributed_sync: the raw histograms of all histogram calibrators are SUM-reduced in one bucket before their
    amax is computed (the reference leaves every rank with its own histogram, calib/histogram.py:158-163); running
    maxima need nothing here -- their amax is MAX-reduced afterwards (sync_amax_bucketed)."""
    from . import calib as _calib

    if _calib.DeferredAmax.current is not None:
        _calib.DeferredAmax.current.flush()  # nobody reads a calibrator before the requests noted for it are answered
    qs = [q for q in _quantizers(model) if q.is_enabled]
    live = [q for q in qs if not (q._use_constant_amax or q._constant_amax is not None)]  # the others were never calibrating
    if distributed_sync and _dist_on():
        mdist.sync_calibrators_bucketed([q._calibrator for q in live if q._calibrator is not None and not q._dynamic
                                         and hasattr(q._calibrator, "share_range_across_ranks")])
    # histogram calibrators: every threshold search is QUEUED first (device kernels, calib.HistogramCalibrator.begin_amax)
    # and read afterwards, so a model's searches overlap and the host does not wait once per quantizer
    tickets = {}
    if method:
        for q in live:
            if q._calibrator is not None and not q._dynamic and hasattr(q._calibrator, "begin_amax"):
                tickets[id(q)] = q._calibrator.begin_amax(method, **kwargs)
    from .calib import MaxCalibrator

    plain_max = [q._calibrator for q in live if type(q._calibrator) is MaxCalibrator and not q._dynamic] if not method else []
    max_verified = bool(plain_max) and MaxCalibrator.verify_finite(plain_max)  # one host read for the NaN / inf asserts
    for q in qs:
        if q._use_constant_amax or q._constant_amax is not None:
            q.enable_quant()  # (:1150-1153)
            continue
        if q._calibrator is not None and not q._dynamic:
            if id(q) in tickets:
                amax = q._calibrator.finish_amax(tickets.pop(id(q)))
            elif max_verified and type(q._calibrator) is MaxCalibrator:
                amax = q._calibrator.compute_amax(verified=True)
            else:
                amax = q._calibrator.compute_amax(**({"method": method, **kwargs} if method else {}))
            if amax is not None:  # quantizers that saw no data keep whatever amax they had (:1155-1161)
                q.replace_amax(amax)  # (buffer shape of the reference for N-D block grids, like load_calib_amax)
        if q.bias_calibrator is not None and q.bias_type == "static":
            q.load_calib_bias()  # the affine offset of the KV-cache presets (:1163-1164)
        q.enable_quant()  # dynamic quantizers come back on here (:1166)
        q.disable_calib()


def _foreign_weight_pairs(m) -> list:
    """(weight, quantizer) pairs of a quantized module of ANOTHER package that enumerates its own weights -- the reference's
    `QuantModule.iter_weights_for_calibration` (nn/modules/quant_module.py:123-129; the per-expert slices of its fused MoE
    containers, plugins/huggingface.py:1085-1101).  Under the algorithm seam such a module keeps its class for the call while
    its quantizers are this package's (modelopt_algorithms.Adoption); the reference calibrates those weights from the tensors
    themselves, whatever the routing of the calibration batches did (model_calib.py:348-352), and so does this."""
    it = getattr(m, "iter_weights_for_calibration", None)
    if not callable(it) or is_quantized_linear(m) or is_quant_fused_experts(m):
        return []
    return [(w, q) for w, q in it() if isinstance(q, (TensorQuantizer, SequentialQuantizer))]


def weight_only_quantize(model: nn.Module, shard: bool = False):
    """model_calib.py:187-199: pass every weight through its quantizer (collects weight statistics).

    Fast path: all enabled per-tensor 'max' weight quantizers of one dtype are calibrated by ONE
    multi-tensor abs-max launch instead of one reduction per layer.

    shard (data-parallel replicas): the (weight, quantizer) list is dealt round-robin over the ranks
    (distributed.shard_list) and every rank calibrates only its share -- independent units, no data-path collective;
    the values reach the other ranks with the amax MAX-reduction that follows (a rank without a value joins with the
    identity, distributed.sync_amax_bucketed).  Only valid when that reduction runs (max_calibrate does both)."""
    pairs = [(m.weight, m.weight_quantizer) for m in model.modules()
             if is_quantized_linear(m) and m.weight_quantizer.is_enabled]
    for m in model.modules():  # fused MoE expert containers: one (slice, quantizer) pair per expert and projection
        if is_quant_fused_experts(m):
            pairs += [(w, q) for w, q in m.iter_weights_for_calibration() if q.is_enabled]
        else:
            pairs += [(w, q) for w, q in _foreign_weight_pairs(m) if q.is_enabled]
    if shard and _dist_on():
        pairs = mdist.shard_list(pairs)
    batched = []
    for w, wq in pairs:
        per_tensor_max = (isinstance(wq, TensorQuantizer) and wq._if_calib and wq.axis is None and wq.block_sizes is None
                          and type(wq._calibrator).__name__ == "MaxCalibrator" and w.is_cuda
                          and w.is_contiguous() and wq.pre_quant_scale is None and wq.bias is None)
        if per_tensor_max:
            batched.append((w, wq))
        else:
            wq(w)
            if isinstance(wq, TensorQuantizer) and wq._if_calib:
                wq.mark_weight_stats_done(w)
    by_key = {}
    for w, wq in batched:
        by_key.setdefault((w.dtype, w.device), []).append((w, wq))
    for group in by_key.values():
        tab = SegmentTable([w.detach() for w, _ in group], outputs=[w.detach() for w, _ in group])
        amax = tab.calibrate_amax()
        held = amax.clone()  # ONE copy; every fresh calibrator keeps a one-element view of it (was a clone launch per weight)
        for i, (w, wq) in enumerate(group):
            cal = wq._calibrator
            if cal._buf is None:
                cal._buf = held[i:i + 1]
                cal._shape, cal._dtype = (), w.dtype
            else:
                torch.maximum(cal._buf, amax[i:i + 1], out=cal._buf)
            wq.mark_weight_stats_done(w)


_MAX_CALIBRATE_DEPTH = [0]


def _deferred_stats_plan(model: nn.Module, forward_loop, defer_stats):
    """(device, decoder layers) when the per-tensor running maxima of this calibration may be answered layer by layer
    (calib.DeferredAmax), else None.  Automatic (defer_stats=None) for Hugging Face decoder stacks on a GPU -- the flush
    points are their decoder layers' ends, and between a linear's forward and the end of its layer such a model leaves the
    linear's input alone (a write is caught at the flush); defer_stats=True asks for it on any model that has a decoder
    stack layerwise.get_decoder_layers finds; False never defers."""
    from . import calib as _calib

    if defer_stats is False or forward_loop is None or _calib.DeferredAmax.current is not None or not isinstance(model, nn.Module):
        return None
    dev = next((p.device for p in model.parameters()), None)
    if dev is None or dev.type != "cuda":
        return None
    from . import layerwise
    from .hf_attention import _is_supported_hf_model

    layers = layerwise.get_decoder_layers(model)
    if layers is None or (defer_stats is None and not _is_supported_hf_model(model)):
        return None
    return dev, list(layers)


@torch.no_grad()
def max_calibrate(model: nn.Module, forward_loop=None, distributed_sync: bool = True, shard_weights: bool | None = None,
                  sync_expert_weight_amax: bool = False, defer_stats: bool | None = None):
    """model_calib.py:310-498 (DP part): collect abs-max statistics for weights and activations, load them,
    then MAX-reduce every amax across the data-parallel group in ONE bucket; each rank's forward_loop sees its own
    share of the calibration batches.

    shard_weights: deal the weight statistics over the ranks (weight_only_quantize(shard=True)); the values reach the
    other ranks with the MAX bucket.  Only correct when every rank holds the SAME weights (data-parallel replicas):
    under tensor parallelism / FSDP the ranks hold different shards and each must calibrate all of its own -- so the
    default (None) shards only after `distributed.declare_data_parallel()`, and never without `distributed_sync`.
    Tensor-parallel callers pass distributed_sync=False and synchronise by `distributed.sync_amax_tensor_parallel`.

    sync_expert_weight_amax: blocks of separate expert modules share one weight amax per projection as well (the input
    amax is always shared, hf_moe.layer_sync_moe_local_experts_amax; model_calib.py:365-368).

    defer_stats: ONE statistics launch per decoder layer instead of one per quantizer call (calib.DeferredAmax;
    _deferred_stats_plan says when).  Same amax, bit for bit: a maximum does not care in which launch it was taken.

    MAX_CALIBRATE_STATS (phase clock, device drained at the phase boundaries) is filled by the OUTERMOST call only: nested
    calls -- per linear inside awq(), per layer in layerwise / gptq, per starved expert in hf_moe -- neither wipe it nor
    add host syncs to the product path."""
    sync = distributed_sync and _dist_on()
    outer = _MAX_CALIBRATE_DEPTH[0] == 0
    stats = MAX_CALIBRATE_STATS if outer else {}
    stats.clear()
    dev0 = next((p.device for p in model.parameters()), None)

    def lap(name, t=[None]):
        if not outer:
            return
        if dev0 is not None and dev0.type == "cuda":
            torch.cuda.synchronize(dev0)
        now = time.perf_counter()
        if t[0] is not None:
            stats[name] = round(now - t[0], 4)
        t[0] = now

    _MAX_CALIBRATE_DEPTH[0] += 1
    try:
        lap(None)
        enable_stats_collection(model, distributed_sync=sync)
        lap("enable_s")
        weight_only_quantize(model, shard=sync and mdist.resolve_shard(shard_weights))
        lap("weights_s")
        if forward_loop is not None:
            plan = _deferred_stats_plan(model, forward_loop, defer_stats)
            if plan is None:
                forward_loop(model)
            else:
                from . import calib as _calib

                # (defer_stats=None: nobody vouched for the model -- the first pass through the layers only watches, see
                # DeferredAmax.probation; defer_stats=True: deferred from the first request, a write is an error)
                batch = _calib.DeferredAmax(plan[0], probation=defer_stats is None, flush_points=len(plan[1]))
                hooks = [layer.register_forward_hook(lambda *_a, _b=batch, _k=id(layer): _b.flush(key=_k)) for layer in plan[1]]
                _calib.DeferredAmax.current = batch
                try:
                    forward_loop(model)
                    batch.flush()  # (requests from outside the decoder stack: the head, a vision tower, ...)
                finally:
                    _calib.DeferredAmax.current = None
                    batch.entries.clear()
                    for h in hooks:
                        h.remove()
                stats["deferred_stats"] = dict(batch.stats)
        lap("forward_loop_s")
        finish_stats_collection(model)
        from . import hf_moe

        hf_moe.layer_sync_moe_local_experts_amax(
            model, sync_weight_amax=sync_expert_weight_amax,
            calibrate_missing=lambda q, w: max_calibrate(q, lambda m: m(w), distributed_sync=False))
        lap("finish_s")
    finally:
        _MAX_CALIBRATE_DEPTH[0] -= 1
    if sync:
        dev = next((p.device for p in model.parameters()), None)
        # the data-parallel group when one was declared (the reference reduces over its DP group, :390-407), else the
        # world; the dealt weight statistics reach the other replicas through this same reduction
        group = mdist.replica_group() if mdist.replicas_declared() else None
        mdist.sync_amax_bucketed(_quantizers(model), group=group, device=dev)
    promote_static_block_weight_quantizers(model)


@torch.no_grad()
def histogram_calibrate(model: nn.Module, forward_loop, method: str = "percentile", distributed_sync: bool = True,
                        **kwargs):
    """The classic histogram flow -- enable_stats_collection, forward, finish_stats_collection(method=...) as users of
    the reference write it by hand (model_calib.py:1128-1167 with calib/histogram.py) -- for models whose quantizers
    were configured with `calibrator: "histogram"`; max calibrators in the same model are loaded as usual.

    Data parallel: every rank feeds its share of the batches; histogram calibrators bin with rank 0's first-batch
    width and their int64 counts are SUM-reduced in one bucket before the threshold search, so every rank computes the
    amax of the WHOLE calibration set -- equal to a single-rank run over all batches when rank 0 holds the first
    batch.  Contract: every rank reaches every histogram-calibrated quantizer (dense models)."""
    sync = distributed_sync and _dist_on()
    enable_stats_collection(model, distributed_sync=sync)
    weight_only_quantize(model, shard=False)
    forward_loop(model)
    hist, rest = nn.ModuleList(), nn.ModuleList()
    for q in _quantizers(model):
        (hist if hasattr(q._calibrator, "share_range_across_ranks") else rest).append(q)
    finish_stats_collection(hist, method, distributed_sync=sync, **kwargs)
    finish_stats_collection(rest)
    if sync:
        dev = next((p.device for p in model.parameters()), None)
        mdist.sync_amax_bucketed(list(rest), device=dev)
    promote_static_block_weight_quantizers(model)


def promote_static_block_weight_quantizers(model: nn.Module) -> int:
    """utils/core_utils.py:1077-1150, INT branch: once their amax is final, enabled static-block INT weight
    quantizers of quantized linears keep `_amax` in fp32 (the reference swaps their class to
    StaticBlockScaleQuantizer).  Later amax writes (MSE refinement, AWQ re-calibration) are then not rounded to
    the weight dtype."""
    n = 0
    for m in model.modules():
        if not is_quantized_linear(m):
            continue
        wq = m.weight_quantizer
        if isinstance(wq, SequentialQuantizer):
            continue
        if (wq.is_enabled and wq.is_static_block_quant and wq.amax is not None and isinstance(wq._num_bits, int)
                and not getattr(wq, "_is_static_block_scale_quantizer", False)):
            wq.promote_static_block()
            n += 1
    return n


# ------------------------------------------------------------------------------------------------ mse
def _mse_quant_func(x, amax, quantizer):
    """model_calib.py:639-662: QDQ `x` with a trial amax through the quantizer itself."""
    had = hasattr(quantizer, "_amax")
    original = quantizer._amax.clone() if had else None
    quantizer._amax = amax
    state = (quantizer._if_quant, quantizer._if_calib)
    quantizer._if_quant, quantizer._if_calib = True, False
    try:
        if hasattr(quantizer, "_original_shape"):
            x = quantizer._reset_to_original_shape(x)
        xq = quantizer(x)
        if hasattr(quantizer, "_block_reshape_size"):
            xq = quantizer._process_for_blockquant(xq)
    finally:
        quantizer._if_quant, quantizer._if_calib = state
        if had:
            quantizer._amax = original
        else:
            delattr(quantizer, "_amax")
    return xq


@torch.no_grad()
def mse_calibrate(model: nn.Module, forward_loop=None, distributed_sync: bool = True, step_size: float = 0.1,
                  start_multiplier: float = 0.25, stop_multiplier: float = 4.0, fp8_scale_sweep: bool = False,
                  shared_states=None):
    """model_calib.py:732-826 (multiplier search, fp8_scale_sweep=False): max calibration first, then every
    eligible weight quantizer's amax is refined by the MSE sweep -- here one fused kernel per weight."""
    if fp8_scale_sweep or shared_states:
        # the FP8-scale sweep of NVFP4 static block scales and shared quantizer state (model_calib.py:740-741, :770-826)
        # belong to formats outside this path; the reference's defaults (False / None) are accepted
        raise MoquantUnsupported("mse_calibrate: fp8_scale_sweep / shared_states are outside this path")
    max_calibrate(model, forward_loop, distributed_sync)
    _mse_calibrate_weights(model, step_size, start_multiplier, stop_multiplier)


def _mse_calibrate_weights(model: nn.Module, step_size: float, start_multiplier: float, stop_multiplier: float,
                           error_func_for=None):
    """_mse_calibrate_weights (model_calib.py:780-826): the weight search of mse_calibrate / local_hessian_calibrate.
    `error_func_for(weight_quantizer)` -> error function or None (plain squared error: the fused one-kernel sweep)."""
    from functools import partial

    from .calib import MseCalibrator

    # every QuantModule's (weight, quantizer) pairs (quant_module.py:123-129): the linears, and a fused MoE expert container's
    # slice per expert and projection (huggingface.py:1084-1100)
    pairs = []
    for m in model.modules():
        if is_quantized_linear(m):
            pairs.append((m.weight, m.weight_quantizer))
        elif is_quant_fused_experts(m):
            pairs += list(m.iter_weights_for_calibration())
        else:
            pairs += _foreign_weight_pairs(m)
    for weight, wq in pairs:
        if (not isinstance(wq, TensorQuantizer) or not wq.is_enabled or wq._dynamic or wq._block_dynamic
                or wq._calibrator is None or getattr(wq, "_amax", None) is None):
            continue  # _make_weight_mse_calibrator's eligibility test (model_calib.py:681-689)
        nb = wq._num_bits
        fused = (nb, wq._unsigned, wq._narrow_range) if isinstance(nb, int) or tuple(nb) == (4, 3) else None
        error_func = error_func_for(wq) if error_func_for is not None else None
        cal = MseCalibrator(amax=wq._amax.clone().detach(), axis=wq._calibrator._axis, step_size=step_size,
                            start_multiplier=start_multiplier, stop_multiplier=stop_multiplier,
                            quant_func=partial(_mse_quant_func, quantizer=wq), error_func=error_func,
                            fused_format=fused if error_func is None else None)
        wq._calibrator = cal
        wq.disable_quant()
        wq.enable_calib()
        wq(weight)
        wq.load_calib_amax(strict=False)
        wq.enable_quant()
        wq.disable_calib()
        cal.reset()


# ------------------------------------------------------------------------------------------------ local Hessian
class _LocalHessianAccumulator:
    """sum X^T X of one linear's input, kept per block of `block_size` input features (model_calib.py:829-899): the
    metric (W - Wq)^T H (W - Wq) of a block of the weight approximates that block's share of the output error.  One
    accumulator serves every linear that reads the same tensor; Cin that the block does not divide has none (plain MSE)."""

    def __init__(self, cout: int, cin: int, block_size: int):
        self.cout, self.cin, self.block_size = cout, cin, block_size
        self.n_blocks = cin // block_size
        self.is_enabled = cin % block_size == 0
        self.hessian_per_block = None  # fp32 [n_blocks, bs, bs], allocated by the first batch
        self.num_samples = 0
        self._mean = None

    @torch.no_grad()
    def accumulate(self, activation: torch.Tensor):
        if not self.is_enabled:
            return
        tokens = activation.reshape(-1, self.cin).float()          # fp32 like the reference (no 16-bit products)
        blocks = tokens.t().reshape(self.n_blocks, self.block_size, -1)  # [n_blocks, bs, tokens]
        gram = torch.bmm(blocks, blocks.transpose(1, 2))
        if self.hessian_per_block is None:
            self.hessian_per_block = gram
        else:
            self.hessian_per_block += gram
        self.num_samples += tokens.shape[0]

    def normalized_hessian(self):
        """H / samples, cached (the raw sum may be released once the error functions exist)."""
        if self._mean is None and self.hessian_per_block is not None and self.num_samples > 0:
            self._mean = self.hessian_per_block / self.num_samples
        return self._mean

    def build_error_func(self, cout: int, keep_buffer: bool = False):
        """error(x, xq): every element gets its Hessian block's weighted error dw^T H dw, so that a sum over the quantizer's
        reduce axes ranks the candidates by it; None without samples."""
        h = self.normalized_hessian()
        if h is None:
            return None
        if not keep_buffer:
            self.hessian_per_block = None
        bs = self.block_size

        def weighted_error(x: torch.Tensor, xq: torch.Tensor) -> torch.Tensor:
            diff = (x - xq).reshape(cout, -1, bs)
            per_block = torch.einsum("cnb,nbd,cnd->cn", diff, h, diff)
            return per_block.reshape(-1, 1).expand(-1, bs).reshape(x.shape)

        return weighted_error


@torch.no_grad()
def local_hessian_calibrate(model: nn.Module, forward_loop=None, distributed_sync: bool = True, step_size: float = 0.1,
                            start_multiplier: float = 0.25, stop_multiplier: float = 4.0, fp8_scale_sweep: bool = True,
                            block_size: int = 16, debug: bool = False, shared_states=None):
    """model_calib.local_hessian_calibrate (model_calib.py:1005-1127): the amax search of mse_calibrate with the
    Hessian-weighted error (W - Wq)^T H (W - Wq), H = per-block sum X^T X of the linear's input taken in a forward with
    the weight quantizers off.  Phases as in the reference: max calibration; Hessians; weight search.
    `fp8_scale_sweep` (the reference's default True) selects its FP8-scale sweep, which exists for static NVFP4
    quantizers only -- every other quantizer is then left at its max-calibrated amax, as in the reference
    (_make_weight_mse_calibrator, :695-718); pass False for the multiplier search over the quantizers of this path.
    The Hessians are not combined across ranks (the reference warns the same)."""
    if forward_loop is None:
        warnings.warn("forward_loop must be provided for local_hessian; skipping local_hessian")
        return
    if shared_states:
        raise MoquantUnsupported("local_hessian_calibrate: shared_states are outside this path")
    max_calibrate(model, forward_loop, distributed_sync)
    linears = [(n, m) for n, m in model.named_modules() if is_quantized_linear(m) and isinstance(m.weight_quantizer, TensorQuantizer)
               and m.weight_quantizer.is_enabled and m.weight.dim() == 2]
    accumulators: dict = {}
    followers: set = set()
    shared = {"input": None, "acc": None}
    for n, m in linears:
        cin = m.weight.shape[1]
        if cin % block_size:
            warnings.warn(f"local_hessian: {n} input features ({cin}) not divisible by block_size ({block_size}); "
                          "falling back to plain MSE for these weights.")
        qb = (m.weight_quantizer.block_sizes or {}).get(-1)
        if qb is not None and qb != block_size:
            warnings.warn(f"local_hessian: block_size ({block_size}) != quantizer scale block ({qb}) for {n}; Hessian "
                          "weighting will not align with the scale blocks.")

    def hook(mod, args):
        if not args:
            return
        x = args[0]
        x = x.to_local() if hasattr(x, "to_local") else x
        wq = mod.weight_quantizer
        acc = accumulators.get(id(wq))
        if acc is None:
            # q / k / v and gate / up read one tensor: the per-block Hessian depends on the input and the block size only
            if shared["input"] is x and shared["acc"].cin == mod.weight.shape[1]:
                accumulators[id(wq)] = shared["acc"]
                followers.add(id(wq))
                return
            acc = accumulators[id(wq)] = _LocalHessianAccumulator(mod.weight.shape[0], mod.weight.shape[1], block_size)
        elif id(wq) in followers:
            if not (shared["input"] is x and shared["acc"] is acc):
                raise RuntimeError("local_hessian: a linear that shared its input with another one in an earlier batch got "
                                   "a different tensor now")
            return  # (this batch is already in the shared accumulator)
        acc.accumulate(x)
        shared["input"], shared["acc"] = x, acc

    handles = [m.register_forward_pre_hook(hook) for _, m in linears]
    qs = [m.weight_quantizer for _, m in linears]
    saved = [q._disabled for q in qs]
    for q in qs:
        q._disabled = True
    try:
        forward_loop(model)
    finally:
        for q, d in zip(qs, saved):
            q._disabled = d
        for h in handles:
            h.remove()
        shared["input"] = shared["acc"] = None
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        warnings.warn("local_hessian: Hessian is not synced across ranks; refined weight amaxes may diverge under "
                      "tensor/data parallelism. Treat local_hessian as single-rank for now.")
    if fp8_scale_sweep:
        warnings.warn("local_hessian: fp8_scale_sweep=True searches static NVFP4 quantizers only (none on this path): "
                      "every weight keeps its max-calibrated amax; pass fp8_scale_sweep=False for the multiplier search.")
        accumulators.clear()
        return
    cout_of = {id(m.weight_quantizer): m.weight.shape[0] for _, m in linears}
    funcs = {qid: acc.build_error_func(cout_of[qid], keep_buffer=debug) for qid, acc in accumulators.items()}
    _mse_calibrate_weights(model, step_size, start_multiplier, stop_multiplier, error_func_for=lambda q: funcs.get(id(q)))
    funcs.clear()
    if debug:
        model._local_hessian_accumulators = accumulators
    else:
        accumulators.clear()


# ------------------------------------------------------------------------------------------------ smoothquant
@torch.no_grad()
def apply_pre_quant_scale_and_smooth(linear: QuantLinear, pre_quant_scale: torch.Tensor, fold_later: list | None = None):
    """model_calib.py:1226-1270: input quantizer gets s (in W.dtype), W <- (W * (1/s)_fp32).to(dtype),
    weight amax recalibrated, input amax <- max(amax_for_smoothing * s).

    fold_later: a list that takes (linear, 1/s) INSTEAD of the weight being folded and re-calibrated here -- the caller folds
    all of them in one launch that also bakes the weight quantizer in (smoothquant(fold_weights=True))."""
    assert linear.input_quantizer.pre_quant_scale is None, "pre_quant_scale should be None first!"
    assert torch.all(pre_quant_scale > 0), "pre_quant_scale should be positive"
    pre_quant_scale = pre_quant_scale.to(torch.float32)
    linear.input_quantizer._enable_pre_quant_scale = True
    linear.input_quantizer.pre_quant_scale = pre_quant_scale.to(linear.weight.dtype)
    inv_scale = 1.0 / pre_quant_scale
    if fold_later is not None:
        fold_later.append((linear, inv_scale))
    else:
        ops.scale_cols(linear.weight.data, inv_scale, out=linear.weight.data)  # fp32 multiply, one rounding
        linear.weight_quantizer.reset_amax()
        max_calibrate(linear, lambda lin: lin.weight_quantizer(lin.weight), distributed_sync=False)
    if linear.input_quantizer.amax is not None:
        dev, dt = linear.weight.device, linear.weight.dtype
        a = linear.input_quantizer._amax_for_smoothing.to(device=dev, dtype=dt)
        linear.input_quantizer.amax = (a * pre_quant_scale.to(dev)).amax().to(dt)


_MX_ELEMENT_FORMATS = {(2, 1): "E2M1", (4, 3): "E4M3", (5, 2): "E5M2", (3, 2): "E3M2", (2, 3): "E2M3", 8: "INT8"}


def _fold_rides_in_mx(m) -> tuple | None:
    """(block, element format) when this linear's fold W <- dt(W / s) can ride inside its MX weight quantizer's fake
    quantization (ops: moq_mt_fold_mx_fused): a dynamic-block E8M0 quantizer along the last axis of a dense 2-D GPU weight --
    nothing to re-calibrate after the fold (block scales come from every input), so fold + QDQ is one read and one write."""
    wq, w = m.weight_quantizer, m.weight
    if not isinstance(wq, TensorQuantizer) or not wq.is_enabled or not wq.is_mx_format or wq.pre_quant_scale is not None:
        return None
    nb = tuple(wq._num_bits) if isinstance(wq._num_bits, (list, tuple)) else wq._num_bits
    fmt = _MX_ELEMENT_FORMATS.get(nb)
    if fmt is None or not set(wq._block_sizes) <= {-1, w.dim() - 1, "type", "scale_bits"}:
        return None
    g = wq._block_sizes.get(-1, None) or wq._block_sizes.get(w.dim() - 1, None)
    vec = 4 if w.dtype == torch.float32 else 8
    if (not g or w.dim() != 2 or not ops._is_gpu(w) or not w.is_contiguous() or w.data_ptr() % 16 or w.shape[-1] % g
            or w.dtype not in (torch.float32, torch.float16, torch.bfloat16) or g % vec or (g // vec) not in (1, 2, 4, 8)):
        return None
    return int(g), fmt


def _broadcast_smoothed(linears, group):
    """Data-parallel SmoothQuant with the fold dealt over the ranks: linear i was smoothed on rank i % world only; the
    folded weight, the pre-quant scale and the re-calibrated amaxes reach the other ranks from there (whole weights
    in place, the small vectors in one bucket per owner)."""
    world, me = dist.get_world_size(group), dist.get_rank(group)
    payload, owners = [], []
    for i, m in enumerate(linears):
        r = i % world
        iq, wq = m.input_quantizer, m.weight_quantizer
        if r != me:  # buffers of the right shape to receive into
            iq._enable_pre_quant_scale = True
            iq.pre_quant_scale = torch.empty(m.weight.shape[1], dtype=m.weight.dtype, device=m.weight.device)
        for t in (m.weight.data, iq._pre_quant_scale, getattr(wq, "_amax", None), getattr(iq, "_amax", None)):
            if t is not None:
                payload.append(t)
                owners.append(r)
    mdist.broadcast_from_owners(payload, group=group, owners=owners)


@torch.no_grad()
def smoothquant(model: nn.Module, forward_loop, alpha: float = 1.0, formats: str = "int8",
                shard_weights: bool | None = None, fold_weights: bool = False):
    """model_calib.py:1273-1359.

    shard_weights (data-parallel replicas; None follows distributed.declare_data_parallel): the weight side of the
    smoothing -- |W| column abs-max, the fold W <- W / s, the re-calibration of the folded weight: three passes over
    every weight -- is dealt over the ranks and the results are broadcast from their owners.  The per-channel
    activation amax is MAX-reduced before (max_calibrate), so every rank would compute the same scales.

    formats = "int8" (default): the reference's behaviour -- only linears whose input AND weight quantizers are INT8
              are smoothed, every other one is skipped with a warning (:1344-1346).
    formats = "all": the per-channel scale math (:1309-1335) and the fold (:1226-1270) do not depend on the number
              format, so any linear with an enabled input quantizer is smoothed -- BASELINE configs[4]: SmoothQuant
              scaling composed with MXFP4 (g = 32, E8M0 block scales).  Per-tensor formats get the reference's
              post-smoothing input amax max(act_amax * s); dynamic-block (MX) quantizers, which recompute their block
              scales from every input, keep no amax at all.  The scales and folded weights are bit-identical to the
              reference's INT8 run on the same model (tests/test_host_flows_cpu.py: sq_mxfp4 fixture).

    fold_weights = True: the call ends like `smoothquant(...); model_quant.fold_weight(model)` -- every weight quantizer baked
              into its weight and switched off (mtq.fold_weight, model_quant.py:728-736) -- and the result is that sequence's
              bit for bit; but a linear whose weight quantizer is a dynamic MX block format (configs[4]: MXFP4 g32) is neither
              folded nor re-calibrated on the way: its smoothing fold rides INSIDE the MX quantize-dequantize, all such
              linears in ONE launch at one read + one write per element (multi_tensor.fold_mx_fused) instead of the fold's read
              + write, the re-calibration's read and the QDQ's read + write."""
    assert forward_loop is not None, "forward_loop must be provided for smoothquant"
    if formats not in ("int8", "all"):
        raise ValueError(f"smoothquant: formats must be 'int8' or 'all', got {formats!r}")
    for m in model.modules():
        if is_quantized_linear(m) and m.input_quantizer.is_enabled and m.input_quantizer.axis is None:
            m.input_quantizer.axis = -1
    max_calibrate(model, forward_loop, shard_weights=shard_weights)
    shard = mdist.resolve_shard(shard_weights)
    world, me = (dist.get_world_size(mdist.replica_group()), dist.get_rank(mdist.replica_group())) if shard else (1, 0)
    smoothed = 0
    dealt = []
    fold_later = []  # (linear, 1 / s) whose fold rides inside the MX QDQ (fold_weights=True)
    for name, m in model.named_modules():
        if not is_quantized_linear(m):
            continue
        iq, wq = m.input_quantizer, m.weight_quantizer
        if not hasattr(iq, "_amax"):
            warnings.warn(f"{name} is not calibrated, skip smoothing")
            continue
        if formats == "int8" and (iq.num_bits != 8 or wq.num_bits != 8):
            warnings.warn(f"Only int8 smoothing is supported, skip {name}")
            continue
        if iq.axis != -1:
            warnings.warn(f"Only per-channel smoothing is supported, skip {name}")
            continue
        act_amax = iq._amax.float().reshape(-1)  # the buffer: `amax` reads None on MX inputs (formats="all")
        dealt.append(m)
        if shard and (len(dealt) - 1) % world != me:
            # another rank smooths this linear; the state every rank sets without touching the weight
            iq.reset_amax()
            iq.axis = None
            if not iq._block_dynamic:
                iq._amax_for_smoothing = act_amax.cpu()
                iq.amax = act_amax.amax().to(dtype=m.weight.dtype, device=m.weight.device)
            smoothed += 1
            continue
        # |W|.amax(dim=0) stays in the WEIGHT dtype and so does its power (model_calib.py:1311, :1319): for a 16-bit
        # model weight_scale^(1 - alpha) is rounded to 16 bits before the fp32 division
        weight_scale = ops.reduce_amax(m.weight, axis=(0,)).reshape(-1)
        scale_a = weight_scale.pow(1 - alpha) / act_amax.pow(alpha)
        iq.reset_amax()
        iq.axis = None
        if not iq._block_dynamic:
            iq._amax_for_smoothing = act_amax.cpu()
            iq.amax = act_amax.amax().to(dtype=m.weight.dtype, device=m.weight.device)
        epsilon = 1.0 / (1 << 31)
        if scale_a.min() <= epsilon:
            scale_a[act_amax <= epsilon] = 1
        scale_a = scale_a.clamp(min=1e-4, max=1e4)
        rides = fold_weights and not shard and _fold_rides_in_mx(m) is not None
        apply_pre_quant_scale_and_smooth(m, scale_a, fold_later=fold_later if rides else None)
        smoothed += 1
    if shard and dealt:
        _broadcast_smoothed(dealt, mdist.replica_group())
    if fold_weights:
        groups = {}
        for m, inv_scale in fold_later:
            groups.setdefault((_fold_rides_in_mx(m), m.weight.dtype, m.weight.device), []).append((m, inv_scale))
        for ((block, fmt), _, _), pairs in groups.items():
            ws = [m.weight.data for m, _ in pairs]
            SegmentTable(ws, outputs=ws).fold_mx_fused([s for _, s in pairs], block, fmt)
        for m, _ in fold_later:
            m.weight_quantizer.disable()  # baked in; fold_weight below leaves it alone and drops what it holds
        from .model_quant import fold_weight

        fold_weight(model, shard_weights=shard_weights)
    return smoothed


# ------------------------------------------------------------------------------------------------ AWQ lite
def get_scale(x_max, w_max, alpha):
    """model_calib.py:1474-1487 (no tensor parallel group here).

    The [Cin] vector math (pow, divide, clamp, normalise) runs where numerics.mode() puts it.  "host" (default): IEEE fp32
    on the host whatever device the statistics live on -- the GPU math library's pow and torch's reciprocal-multiply for
    `tensor / scalar` differ from the host's in the last bit, which would make the folded weights, and with them every byte
    of the exported checkpoint, depend on the device the search ran on; the result equals the reference's CPU run.
    "device": the same expression on the statistics' device -- the reference's run on THAT device bit for bit, no round
    trip.  The result goes back to x_max's device."""
    dev = x_max.device
    x, w = numerics.vec(x_max), numerics.vec(w_max)
    w = w.to(x.device)
    scales = (x.pow(alpha) / (w.pow(1 - alpha) + torch.finfo(torch.float32).tiny)).clamp(min=1e-4, max=1e4).view(-1)
    return (scales / (scales.max() * scales.min()).sqrt()).view(-1).to(dev)


import contextlib


@contextlib.contextmanager
def host_math_threads(n: int = 1):
    """The [Cin]-sized host math of the AWQ search (get_scale and friends: ~30 small CPU tensor ops per candidate) on
    ONE thread.  torch parallelises some of these ops (sqrt, reductions) over its OpenMP pool even for a few thousand
    elements; on hosts where the visible CPUs exceed what the process may use (the GPU boxes show 256 CPUs under a
    16-CPU quota) every such fork/join stalls for milliseconds -- measured 1.35 s vs 0.25 s for the 2464 scale vectors
    of Llama-3-8B in the build container, 1.3-1.6 s of the flow on the GPU box.  Elementwise results do not depend on
    the thread count."""
    before = torch.get_num_threads()
    try:
        torch.set_num_threads(n)
        yield
    finally:
        torch.set_num_threads(before)


class _WeightCacheBudget:
    """HBM budget for keeping the 11 search weights QDQ(W * s_alpha) of every linear resident between
    calibration batches (the reference re-quantizes them on every forward, model_calib.py:1552-1554).  288 GB of
    HBM3E hold all of Llama-3-8B's candidates (11 x 14 GB); larger models cache what fits and recompute the rest."""

    host_bytes = 0  # budget on a non-GPU device (the CPU test tier's stand-in backend sets it)

    def __init__(self, device):
        self.left = int(self.host_bytes)
        if device.type == "cuda":
            # what the flow can still get: the driver's free memory PLUS what torch's caching allocator holds without using
            # (a process that has run anything large before sits on such blocks -- and a harness may park the flow's memory
            # there on purpose, tools/awq_bench.py: counting the driver's share alone sent that run into a second pass)
            free, _ = torch.cuda.mem_get_info(device)
            cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
            self.left = int((free + max(cached, 0)) * 0.6)

    def reserve(self, nbytes: int) -> bool:
        if nbytes <= self.left:
            self.left -= nbytes
            return True
        return False

    def release(self, nbytes: int):
        self.left += nbytes


def _awq_searched(m) -> bool:
    """awq_lite's own rule (model_calib.py:1563-1567): EVERY quantized linear whose weight quantizer is enabled -- whatever
    its format: a per-layer override that puts FP8 or INT8 on part of an INT4-AWQ model is smoothed and searched too."""
    return is_quantized_linear(m) and m.weight_quantizer.is_enabled


def _awq_fused_format(wq, weight) -> bool:
    """INT-k static blocks along the last axis only: the format the fused search kernels (scale + per-group QDQ, weight scale,
    Gram scores) are written for.  Everything else takes the GENERIC route: the same search with the scaled weight passed
    through the quantizer itself (dynamic amax of its own layout) and the error-GEMM engine."""
    if not (isinstance(wq, TensorQuantizer) and wq.is_static_block_quant and isinstance(wq._num_bits, int)):
        return False
    axes = {(k % weight.dim()) for k in wq._block_sizes if isinstance(k, int)}
    return axes == {weight.dim() - 1}


def _awq_block_size(wq, weight):
    """_get_awq_quantizer_block_size (model_calib.py:1943-1952): None without block sizes (whole rows)."""
    if wq.block_sizes is None:
        return None
    if -1 in wq.block_sizes:
        return wq.block_sizes[-1]
    if 1 in wq.block_sizes:
        return wq.block_sizes[1]
    raise ValueError("AWQ requires block quantization along -1 axis")


def _awq_weight_scale_generic(weight: torch.Tensor, block) -> torch.Tensor:
    """get_weight_scale (model_calib.py:1453-1469) for the layouts moq_awq_weight_scale does not take (whole rows, blocks that
    are no power-of-two number of packets): every |w| over the abs-max of its block (+ tiny) in the weight dtype, the mean over
    the output rows (torch: fp32 accumulation, one rounding to the dtype), widened to fp32."""
    rows, cin = weight.shape
    w = weight.detach()
    if block and cin % block:
        w = F.pad(w, (0, block - cin % block), "constant", 0)
    groups = w.contiguous().view(-1, block) if block else w.contiguous()
    top = ops.reduce_amax(groups, axis=1, keepdims=True).to(w.dtype)
    share = (groups.abs() / (top + torch.finfo(w.dtype).tiny)).view(rows, -1)[:, :cin]
    return share.mean(0).to(torch.float32)


class AWQLiteHelper:
    """Per-linear state of awq_lite (model_calib.py:1416-1451)."""

    def __init__(self, module: QuantLinear, alpha_step: float):
        wq = module.weight_quantizer
        self.fused = _awq_fused_format(wq, module.weight)
        if not self.fused:
            self._init_generic(module, alpha_step)
            return
        self.block_size = wq.block_sizes.get(-1, None) or wq.block_sizes.get(module.weight.dim() - 1)
        # Cin that is not a block multiple: the reference zero-pads the last block (get_weight_scale, :1453-1469; the
        # static block quantizer, tensor_quantizer.py:975-1043).  The kernels take whole blocks, so such a linear works on
        # a zero-padded copy of its weight and cuts the padding off again: zeros change neither a block's abs-max nor
        # any quantized value.
        self.cin = module.weight.shape[1]
        self.dtype = module.weight.dtype  # 1 / s is rounded to it where the forward uses it (prepare_scales)
        self.pad = (-self.cin) % self.block_size
        if numerics.on_host():
            self.weight_scale = ops.awq_weight_scale(self._padded(module.weight), self.block_size)[:self.cin].contiguous()
        else:
            # numerics "device" = the reference's run on this device: like the activation mean (patched_forward), this statistic
            # is a 16-bit-rounded MEAN -- over Cout of |w| / (group amax + tiny) -- whose last bit depends on the summation
            # order at full width; torch's own expression (get_weight_scale, :1453-1469), a handful of passes over the weight,
            # once per linear
            self.weight_scale = _awq_weight_scale_generic(module.weight, self.block_size).contiguous()
        self._init_state(module, alpha_step)

    def _init_generic(self, module, alpha_step):
        self.block_size = _awq_block_size(module.weight_quantizer, module.weight)
        self.cin = module.weight.shape[1]
        self.dtype = module.weight.dtype
        self.pad = 0
        self.weight_scale = _awq_weight_scale_generic(module.weight, self.block_size)
        self._init_state(module, alpha_step)

    def _init_state(self, module, alpha_step):
        self.act_sum = torch.zeros(module.weight.shape[1], dtype=torch.float32, device=module.weight.device)
        self.act_scale = None
        self.num_cache_steps = 0
        self.num_search_steps = 0
        self.num_tokens = 0
        self.alphas = [k.item() for k in torch.arange(0, 1.0 + alpha_step, alpha_step)]  # same float32 keys as :1431
        # one device buffer for all per-alpha losses; loss[alpha] is a 1-element view the kernels accumulate into
        self.loss_buf = torch.zeros(len(self.alphas), dtype=torch.float32, device=module.weight.device)
        self.loss = {a: self.loss_buf[i:i + 1] for i, a in enumerate(self.alphas)}
        self.best_alpha = None
        self.best_scale = None
        self.is_enabled = True  # False: NaN in a scale on some rank, or no tokens anywhere (:1605-1629)
        self.is_input_quantized = False  # set by awq_lite's setup (:1432)
        self.setup_disabled = False  # an input quantizer with a channel axis other than the last one (:1441-1443)
        # search-pass caches (alpha -> tensors); scales depend on alpha only once act_scale is final
        self._inv_scale = None
        self._scale_dt = None
        self._w_hat = None
        self._cache_w = False
        self._s_host = self._s_dev = self._r_dev = None  # prepare_scales
        # Gram-matrix search: G = sum_b X_b^T X_b / T_b (fp32 [Cin, Cin]), accumulated in the cache pass
        self.gram = None
        self.use_gram = False
        self.gram_owner = None  # the helper whose Gram matrix this one aliases (same input tensor)
        self.gram_stage = None  # ops.GramStage: several calibration batches per Gram launch
        self.gram_stage_denied = False
        self.num_gram_steps = 0  # batches that reached the Gram accumulation (own or aliased)
        self.gram_symmetrized = False
        self.gram_bytes = 0  # what this helper holds of the HBM budget for its Gram matrix
        # tie-aware re-evaluation (search="auto"): candidates whose Gram loss lies within the margin of the best one
        # are re-scored by the exact-rounding error-GEMM engine; `gram_loss` keeps the Gram scores for inspection
        self.scored_here = False  # this rank evaluated the Gram scores (data parallel: one rank per Gram matrix)
        self.gram_loss = None
        self.contenders = None  # indices into `alphas`, ascending, of every re-scored candidate; None: the Gram scores decide
        self.exact_buf = None  # fp32 [len(contenders)]: their exact scores
        self.num_exact_steps = 0
        self.pending = None  # candidates the NEXT exact pass scores (indices, ascending); exact_pass_buf accumulates them
        self.exact_pass_buf = None
        self.exact_scores = {}  # candidate index -> exact score (host floats, summed over the ranks)
        self.margin_used = None  # relative Gram margin the contender set was cut at (widened by the tie check)
        self.tie_need = None  # TIE_SPREAD_FACTOR * S + gap_w of the last check, relative (None: not re-scored)
        self.tie_rounds = 0  # how often the margin of this linear was widened
        self.stored = []  # store_activations: (input [T, Cin], out_actual [T, Cout]) of every cache-pass batch
        self.act_owner = None  # the helper whose act_sum this one aliases (same input tensor in the cache pass)

    def _padded(self, w: torch.Tensor, value: float = 0.0) -> torch.Tensor:
        return F.pad(w, (0, self.pad), "constant", value) if self.pad else w

    def prepare_scales(self, act_host: torch.Tensor, weight_host: torch.Tensor, dtype: torch.dtype):
        """All candidate scale vectors of this linear at once, on the HOST (get_scale), uploaded in ONE copy: s_alpha
        (fp32) and the input-side 1 / s_alpha as the forward uses it (rounded to the model dtype).  Done for every linear
        before the first scoring kernel is queued: a device -> host round trip per candidate inside the scoring loop
        would drain the stream 2464 times for Llama-3-8B (measured: +6 s on 18 s)."""
        self.upload_scales(self.host_scales(act_host, weight_host, dtype))

    def host_scales(self, act_host: torch.Tensor, weight_host: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
        """The host half of prepare_scales: [2 A, Cin] fp32 = all s_alpha, then all (1 / s_alpha) rounded to `dtype`.
        Pure CPU tensor math on this linear's own vectors."""
        s = torch.stack([get_scale(act_host, weight_host, a) for a in self.alphas])  # [A, Cin] fp32
        return torch.cat([s, (1.0 / s).to(dtype).float()])

    def upload_scales(self, both: torch.Tensor):
        n = len(self.alphas)
        self._s_host = both[:n]
        up = both.to(self.act_scale.device)
        self._s_dev, self._r_dev = up[:n], up[n:]

    def scale(self, alpha):
        """get_scale(act_scale, weight_scale, alpha) on the statistics' device (computed on the host, see prepare_scales)."""
        if self._s_dev is None:
            self.prepare_scales(numerics.vec(self.act_scale), numerics.vec(self.weight_scale), self.dtype)
        return self._s_dev[self.alphas.index(alpha)]

    def search_operands(self, module, subset=None):
        """(inv_s [A, Cin] fp32 = (1/s_alpha) rounded to the weight dtype, w_hat [A, Cout, Cin] = QDQ(W * s_alpha))
        for all candidates (or the `subset` of candidate indices -- fixed for the life of the caches).  Scales depend
        on alpha only (act_scale is final in the search pass) and are computed once; w_hat stays resident when the
        budget allows, otherwise it is rebuilt per batch."""
        dt = module.weight.dtype
        if self._inv_scale is None:
            alphas = self.alphas if subset is None else [self.alphas[i] for i in subset]
            scales = [self.scale(a) for a in alphas]
            idx = torch.tensor([self.alphas.index(a) for a in alphas], device=self._r_dev.device)
            self._inv_scale = self._r_dev.index_select(0, idx).contiguous()
            self._scale_dt = [s.to(dt) for s in scales]
        w_hat = self._w_hat
        if w_hat is None:
            w_hat = torch.empty(len(self._scale_dt), *module.weight.shape, dtype=dt, device=module.weight.device)
            if not self.fused:
                # the scaled weight through the quantizer itself, which holds no amax during the search and takes that of
                # its input in its own layout (per tensor, per channel, 2-D blocks, ...): the weight quantizer with
                # pre_quant_scale = s of the patched forward (:1548-1554)
                for i, s in enumerate(self._scale_dt):
                    w_hat[i].copy_(module.weight_quantizer(ops.scale_cols(module.weight.detach(), s.float())))
            elif self.pad:  # whole blocks for the kernel (zero weight columns, unit scales), the padding cut off again
                w_pad = self._padded(module.weight)
                for i, s in enumerate(self._scale_dt):
                    y = ops.awq_scale_qdq(w_pad, self._padded(s, 1.0), self.block_size, module.weight_quantizer.num_bits)
                    w_hat[i].copy_(y[:, :self.cin])
            else:
                for i, s in enumerate(self._scale_dt):
                    ops.awq_scale_qdq(module.weight, s, self.block_size, module.weight_quantizer.num_bits, out=w_hat[i])
            if self._cache_w:
                self._w_hat = w_hat
        return self._inv_scale, w_hat

    def release(self):
        self._inv_scale = self._scale_dt = self._w_hat = None


def _gram_losses(h: AWQLiteHelper, module: QuantLinear):
    """All per-alpha losses of one linear from its Gram matrix, without touching the activations again.

    The reference's loss (model_calib.py:1489-1495, :1552-1556) is sum_b mean_{t,n} (out_alpha - out_actual)^2 with
    out_alpha = (x * 1/s) @ QDQ(W * s)^T and out_actual = x @ W^T, i.e. out_alpha - out_actual = x @ E^T with the
    fp32 error weight E = QDQ(W * s) * (1/s) - W.  Hence  sum_b ||X_b E^T||_F^2 / (T_b N) = trace(E G E^T) / N  with
    G = sum_b X_b^T X_b / T_b: the 11 GEMMs over every calibration token become 11 Cout x Cin x Cin contractions that
    do not depend on the number of tokens (Llama-3-8B, 512 x 512 tokens: 4.4e16 -> 0.6e16 FLOP including the Gram
    accumulation).  What it leaves out is the reference's rounding of x/s, out_alpha and out_actual to the model
    dtype -- a noise floor ~1e-3 below the INT4 error energy (fp32 models agree to 4e-7, bf16 within 1e-2 on the
    reference fixtures, same best alpha; tests/test_gpu_host.py, tests/test_gpu_awq_search.py)."""
    w = module.weight
    dt = w.dtype
    n_out = w.shape[0]
    bits = module.weight_quantizer.num_bits
    # 16-bit models: the Cout x Cin x Cin contraction runs on the matrix cores in split precision (three bf16
    # products summed in the fp32 accumulators, ~1e-5 relative) with the <., E> product fused; fp32 models run the same
    # contraction on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32 products; agreement with the reference to
    # ~1e-6 is asserted for them).  The library GEMM is left for widths the kernels do not take and for CPU tensors.
    mfma = dt in (torch.bfloat16, torch.float16) and w.shape[1] % 8 == 0 and w.shape[1] % h.block_size == 0
    wf = None if mfma else w.float()
    f32_mfma = (not mfma and dt == torch.float32 and w.is_cuda and w.shape[1] % 8 == 0 and w.shape[0] % 4 == 0
                and h.gram.is_contiguous())
    planes = gram_score_planes(dt)
    gram_op = ops.gram_operand(h.gram, planes) if mfma else None
    for i, alpha in enumerate(h.alphas):
        s = h.scale(alpha)
        r = h._r_dev[i]  # (1 / s) rounded to the model dtype: input_quantizer.pre_quant_scale as the forward uses it (:1551)
        if mfma:
            # E and its split-precision MFMA operand from ONE read of W, then <E G, E> on the matrix cores
            err, a_op = ops.awq_err_weight(w, s.to(dt), r, h.block_size, bits, planes)
            ops.awq_quadform(err, gram_op, h.loss_buf[i:i + 1], 1.0 / n_out, a_operand=a_op)
        else:
            w_hat = ops.awq_scale_qdq(w, s.to(dt), h.block_size, bits)  # QDQ((W * s).to(dtype)), one kernel
            err = w_hat.float().mul_(r).sub_(wf)
            if f32_mfma:  # <E G, E> on the fp32 matrix cores (G is symmetric: the NT kernel's G^T is G)
                ops.awq_quadform(err, h.gram, h.loss_buf[i:i + 1], 1.0 / n_out)
            else:
                h.loss_buf[i] += (torch.matmul(err, h.gram) * err).sum() / n_out


# Relative margin inside which two candidates' Gram scores do not decide the search (search="auto").  The Gram loss
# differs from the reference-structured loss by d(alpha) = loss_gemm - loss_gram: the energy of the roundings of x/s,
# `out` and `out_actual` to the model dtype.  Two candidates can only swap places if d differs between them by more than
# their score gap.  Measured on the full-size synthetic Llama-3-8B run, every candidate scored by BOTH engines (224
# linears x 11 candidates, 64 x 4096 tokens; profiles/r02_awq_tie_margin.md): d is a smooth function of the loss -- over
# all 12 320 candidate pairs |d_i - d_j| is at most 6 % of the score gap for gaps above 1e-3 (1.3 % above 5e-3), so
# distant candidates never swap -- plus a zero-mean part (cross term error x rounding) that decides only pairs closer
# than ~1e-5 (the largest gap any pair was overturned at: 6.4e-6).  The margins are 100x that for bf16 and, with the
# 8x finer rounding, 30x for f16; fp32 models agree with the reference to 4e-7 (tests/test_gpu_host.py).  The zero-mean
# part shrinks with the number of outputs in the loss: GRAM_TIE_NOISE / sqrt(tokens * Cout) is added (~8 sigma; 1.6e-5
# at full size, 2e-3 for 200-token test shapes).
GRAM_TIE_MARGIN = {torch.bfloat16: 1e-3, torch.float16: 2e-4, torch.float32: 2e-5}
GRAM_TIE_NOISE = 0.5
# what the last awq_lite call did: wall-clock per stage, passes over the calibration data, re-scoring counts (tools,
# bench.py `extra`); overwritten by every call
AWQ_LITE_STATS: dict = {}
# wall-clock of the last max_calibrate call by phase (the device is drained at the phase boundaries)
MAX_CALIBRATE_STATS: dict = {}
# bf16 planes of the Gram scoring contraction <E G, E> (ops.gram_operand): 3 = split precision in both factors.  bf16
# models screen with ONE plane (E and G rounded to bf16, a third of the contraction and of the operand memory): on the
# full-size run (profiles/r02_awq_tie_margin.md, "scoring planes") the one-plane scores differ from the three-plane ones
# by at most 2.9e-4 relative, by at most 2.7e-4 between any two candidates of a linear and 6e-5 between candidates within
# 5e-3 of the best; GRAM_PLANES_SLACK widens the re-scoring margin by the larger figure.  Same 224 / 224 alphas, same
# 106 re-scored candidates.  f16 models (margin 2e-4) keep the three planes.
GRAM_SCORE_PLANES = {torch.bfloat16: 1, torch.float16: 3}
GRAM_PLANES_SLACK = 3e-4
# The margin checks itself (search="auto", tie_check=True).  For every re-scored candidate i both scores are known, hence
# d_i = (exact_i - gram_i) / best_gram; S = max d - min d over the re-scored set measures how far the two engines disagree
# about THIS linear's candidates.  A candidate k left out (Gram gap > margin) could still beat the exact winner w only if
# d_w - d_k exceeds its Gram distance to w, i.e. margin - gap_w at least.  The unscored d_k is not known; it is taken to
# lie within TIE_SPREAD_FACTOR x S of the scored ones (d varies smoothly with alpha; the factor covers the extrapolation
# past the scored set).  So a linear is settled when  margin >= TIE_SPREAD_FACTOR * S + gap_w ; otherwise its margin is
# widened to twice that requirement, the newly admitted candidates are re-scored in a further pass, and the check repeats
# (at most TIE_CHECK_MAX_ROUNDS times, then every candidate of the linear is scored).  Full-size measurements
# (profiles/r03_awq_tie_check.md): the synthetic outlier stack shows S <= 4.6e-5 against a 1.3e-3 margin (requirement /
# margin 0.07 whatever the factor); a random-init HF Llama-3-8B, whose 11 candidates lie within 0.1-0.9 % of each other,
# shows S up to 8e-4 among its contenders and a largest overturned Gram gap of 1.1e-4: factor 1.5 settles every linear
# at once with requirement / margin up to 0.98 -- at the edge -- factor 2 widens 5 of 224 linears, same alphas.  Round 3
# shipped 1.5 because a widening cost one more PASS over the calibration data; since round 4 the exact pass of a decoder
# stack is a replay of stored activations (layer_local), a widening re-scores a handful of candidates of that one linear,
# and the default is the safer 2.0.
TIE_SPREAD_FACTOR = 2.0
# ... and with room to spare (round 6).  Rounds 4 and 5 settled the random-init HF model with requirement / margin 0.984 on
# its tightest linear: inside the rule, by 2 %, on the only adversarial input measured.  A linear now counts as settled only
# when the requirement is at most TIE_HEADROOM of its margin; one that passes by less is widened like one that fails.  The
# reported `max_need_over_margin` of a run is therefore <= 0.7 by construction, and what the extra widenings cost is a few
# more candidates re-scored from stored activations (measured on the stored full-size tables, tests/test_awq_tie_check_cpu.py).
TIE_HEADROOM = 0.7
TIE_CHECK_MAX_ROUNDS = 3


def tie_margin_check(gram, exact_scores: dict, margin: float, rounds: int):
    """One round of the margin's self-check for one linear (see TIE_SPREAD_FACTOR).  gram: the Gram scores of all
    candidates; exact_scores: {candidate index: exact score} of the re-scored ones; margin: the relative Gram margin
    they were admitted with; rounds: how often it was widened already.  Returns (need, new margin, new candidates):
    need = TIE_SPREAD_FACTOR * S + gap_w (None when it cannot be formed); new candidates = [] when the linear is
    settled (need <= TIE_HEADROOM * margin, or every candidate is scored)."""
    scored = sorted(exact_scores)
    best = min(gram)
    if len(scored) >= len(gram) or not math.isfinite(best) or best <= 0.0:
        return None, margin, []
    e = [exact_scores[i] for i in scored]
    if not all(math.isfinite(v) for v in e):  # NaN / inf somewhere: score everything
        return None, float("inf"), [i for i in range(len(gram)) if i not in exact_scores]
    d = [(exact_scores[i] - gram[i]) / best for i in scored]
    spread = max(d) - min(d)
    w = scored[min(range(len(e)), key=e.__getitem__)]
    need = TIE_SPREAD_FACTOR * spread + (gram[w] - best) / best
    if need <= TIE_HEADROOM * margin:
        return need, margin, []
    margin = float("inf") if rounds + 1 >= TIE_CHECK_MAX_ROUNDS else max(2.0 * need, 2.0 * margin)
    new = [i for i, v in enumerate(gram) if i not in exact_scores and v <= best * (1.0 + margin)]
    while not new and math.isfinite(margin):
        # nothing lies inside the widened margin: the requirement is then judged against THAT margin (settled with room), or
        # the margin grows until it admits a candidate / covers the requirement
        if need <= TIE_HEADROOM * margin:
            return need, margin, []
        margin *= 2.0
        new = [i for i, v in enumerate(gram) if i not in exact_scores and v <= best * (1.0 + margin)]
    return need, margin, new


def gram_score_planes(dtype) -> int:
    """bf16 planes of the screening contraction for this model dtype (GRAM_SCORE_PLANES; a study that wants another
    precision edits that table from its own script -- the product path reads no environment knob here)."""
    return GRAM_SCORE_PLANES.get(dtype, 3)


def _layer_local_plan(model: nn.Module, explicit: bool = False):
    """The decoder layers awq_lite may walk one at a time, or None.  Conditions: a decoder stack exists (layerwise.
    get_decoder_layers) and -- unless the caller asked for it explicitly -- the model is a Hugging Face PreTrainedModel whose
    stack is made of *DecoderLayer modules (the contract "layer N+1's first argument is layer N's output" is theirs; any
    nn.ModuleList of equal children would otherwise qualify); every searched linear lives inside the stack; no OTHER
    quantizer is enabled (KV-cache / input quantizers outside the search add their noise in the reference's search pass,
    which a single pass cannot reproduce)."""
    from . import layerwise
    from .hf_attention import _is_supported_hf_model

    layers = layerwise.get_decoder_layers(model)
    if layers is None:
        return None
    if not explicit and not (_is_supported_hf_model(model) and type(layers[0]).__name__.endswith("DecoderLayer")):
        return None
    mods = [m for m in model.modules() if _awq_searched(m)]
    inside = {id(m) for lyr in layers for m in lyr.modules()}
    if not mods or any(id(m) not in inside for m in mods):
        return None
    if not all(_awq_fused_format(m.weight_quantizer, m.weight) for m in mods):
        return None  # a linear in another format (generic route of the search): the whole-model flow
    searched = {id(q) for m in mods for q in (m.weight_quantizer, m.input_quantizer)}
    if any(q.is_enabled and id(q) not in searched for q in _quantizers(model)):
        return None
    return layers


_NO_INSTANCE_FORWARD = object()


def _restore_forward(m, previous):
    """Undo `m.forward = patched`: put back the INSTANCE attribute the module had before (an accelerate hook, a user's patch) or
    remove the instance attribute, so that the class's own forward is looked up again -- assigning the bound method read before the
    patch would pin the module to the class it had during the search (under the algorithm seam that class is handed back to the
    reference afterwards; the reference's `unpatch_forward_method` does the same: utils/network.py)."""
    if previous is _NO_INSTANCE_FORWARD:
        m.__dict__.pop("forward", None)
    else:
        m.forward = previous


@torch.no_grad()
def _awq_lite_layer_local(model: nn.Module, forward_loop, layers, **kw):
    """awq_lite one decoder layer at a time (see awq_lite, `layer_local`).  The model's own forward runs up to the first
    decoder layer once per batch (embeddings, rotary tables: `layerwise._capture_inputs`); from there every layer is
    called on the previous layer's outputs with the captured arguments.  KV-cache objects among those arguments are
    dropped (a layer replayed outside its model must not append to a cache another call owns)."""
    from . import layerwise

    dev = next(model.parameters()).device
    total = {"search": kw.get("search", "auto"), "passes": 0, "replayed_passes": 0, "layer_local": True, "stages_s": {},
             "linears": 0, "rescored_linears": 0, "rescored_candidates": 0, "layers": len(layers),
             "tie_check": {"enabled": bool(kw.get("tie_check", True)), "spread_factor": TIE_SPREAD_FACTOR,
                           "checked_linears": 0, "widened_linears": 0, "max_need_over_margin": 0.0, "max_need": 0.0}}

    def drained_clock():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        return time.perf_counter()

    t0 = drained_clock()
    inputs = layerwise._capture_inputs(model, layers[0], forward_loop)

    without_cache = layerwise.without_cache

    reached = torch.tensor([float(len(inputs) > 0)], device=dev)
    if _dist_on():  # data parallel: the same flow on every rank (a rank whose shard is empty walks the layers with no batches)
        dist.all_reduce(reached, op=dist.ReduceOp.MAX)
    if reached.item() == 0:
        return None  # forward_loop does not drive the decoder stack (e.g. it calls the linears directly)
    inputs = [(args, without_cache(kwargs)) for args, kwargs in inputs]
    total["stages_s"]["embed"] = round(drained_clock() - t0, 4)
    helpers = {}
    most_passes = 0
    for idx, layer in enumerate(layers):
        nxt, calls, is_last = [], {"n": 0}, idx + 1 == len(layers)

        def layer_loop(m, _in=inputs, _nxt=nxt, _calls=calls, _last=is_last):
            first = _calls["n"] == 0  # the cache pass: the layer still has its original weights and the searched linears
            _calls["n"] += 1          # return out_actual -- its outputs are the un-quantized inputs of the next layer
            for args, kwargs in _in:
                out = m(*args, **kwargs)
                if first and not _last:
                    _nxt.append(((layerwise._first_tensor(out).detach(), *args[1:]), kwargs))

        helpers.update(awq_lite(layer, layer_loop, layer_local=False, store_activations=True, **kw))
        inner = AWQ_LITE_STATS
        for k, v in (inner.get("stages_s") or {}).items():
            total["stages_s"][k] = round(total["stages_s"].get(k, 0.0) + v, 4)
        most_passes = max(most_passes, inner.get("passes", 0))
        total["replayed_passes"] += inner.get("replayed_passes", 0)
        for k in ("linears", "rescored_linears", "rescored_candidates"):
            total[k] += inner.get(k, 0)
        tc, ic = total["tie_check"], inner.get("tie_check") or {}
        for k in ("checked_linears", "widened_linears"):
            tc[k] += ic.get(k, 0)
        for k in ("max_need_over_margin", "max_need"):
            tc[k] = max(tc[k], ic.get(k, 0.0))
        inputs = nxt
    total["passes"] = most_passes  # forwards through every layer: 1 when every layer's exact pass was a replay
    AWQ_LITE_STATS.clear()
    AWQ_LITE_STATS.update(total)
    return helpers


@torch.no_grad()
def awq_lite(model: nn.Module, forward_loop, alpha_step: float = 0.1, search: str = "auto",
             tie_margin: float | None = None, tie_check: bool = True, layer_local: bool | None = None,
             store_activations: bool | str = "auto"):
    """AWQ-lite (model_calib.py:1394-1721) over every quantized linear whose weight quantizer is enabled (:1563-1567).
    INT-k static blocks along the last axis (INT4_AWQ_CFG, the first stage of W4A8_AWQ_BETA_CFG) take the fused kernels and
    the engines below; a linear in any other format (a per-layer override: per-tensor FP8, per-channel INT8, 2-D blocks, ...)
    takes the GENERIC route -- weight scale over its own blocks or whole rows, every candidate's scaled weight through the
    quantizer itself with the amax of its own layout, scored by the error-GEMM engine (AWQLiteHelper.fused).

    search = "gemm": the reference's structure -- two passes of forward_loop (cache: act scales; search: for every
             alpha the patched forward's GEMM, here one batched MFMA error-GEMM launch per linear and batch);
    search = "gram": ONE pass of forward_loop that also accumulates every linear's Gram matrix on the matrix cores
             (ops.hessian_accum); all alpha losses then come from trace(E G E^T) (see _gram_losses);
    search = "auto" (default): the Gram scores screen the candidates of every linear whose Gram matrix fits the HBM
             budget ("gemm" for the rest); candidates whose score lies within `tie_margin` (relative; default
             GRAM_TIE_MARGIN of the weight dtype) of the linear's best one are re-scored in a further pass by the
             error-GEMM engine -- the reference's arithmetic with all its roundings -- and the best alpha is the
             first minimum of THOSE scores, so that the selection equals search="gemm" (model_calib.py:1489-1495,
             :1548-1556, :1637).  Linears with a clear winner cost nothing extra; when no linear has a near-tie the
             extra pass is skipped.  tie_check (default on): the margin verifies itself against the measured
             disagreement of the two engines on the re-scored candidates and widens per linear when it was too
             small (TIE_SPREAD_FACTOR); AWQ_LITE_STATS["tie_check"] reports the largest requirement / margin ratio.

    store_activations: True -- the cache pass keeps every searched linear's input and `out_actual` of every batch (references,
             counted against the HBM budget); the exact pass then REPLAYS them linear by linear instead of running
             forward_loop again -- no second forward, no second library GEMM for out_actual.  Only a call whose
             activations fit can do that (one decoder layer: 36 GB for Llama-3-8B at 64 x 4096 tokens); when a
             reservation fails the stores are dropped and the pass is a real one.
             "inputs" -- only the INPUTS are kept, each distinct tensor object once (q / k / v and gate / up share theirs; a
             model fed the same tensors layer after layer shares them across layers); the replay recomputes `out_actual`
             with the linear's own un-folded weight, and only for the linears that have candidates to re-score.  A stored
             tensor that was written in place afterwards (its version counter moved) sends the pass back to a real forward.
             "auto" (default) = "inputs" for search="auto" when no other quantizer is enabled (their noise belongs to a real
             search pass), else off; an automatic store stops at half of what the HBM budget has left (the forward's own peak is
             not in that budget) and the pass is then a real one.  False = always a second forward.
    layer_local (None = for search="auto" on Hugging Face decoder stacks): the model is walked ONE DECODER LAYER AT A TIME
             (layerwise.py's contract: layer N+1's input is layer N's output): the layer's batches run through it once --
             statistics, Gram matrices, stored activations and the inputs of the next layer all come from that one
             run -- its linears are scored, their near-ties re-scored from the stores, the layer is folded, the next one
             follows.  One pass over the calibration data in total (`passes == 1`), same statistics as the whole-model
             flow (every layer sees the un-quantized output of its predecessor, as in the reference's cache pass)."""
    if forward_loop is None:  # (:1412-1414)
        warnings.warn("forward_loop must be provided for awq_lite; skipping awq_lite")
        return None
    if search not in ("auto", "gram", "gemm"):
        raise ValueError(f"awq_lite: unknown search mode {search!r}")
    if layer_local is None or layer_local:
        plan = _layer_local_plan(model, explicit=bool(layer_local)) if search == "auto" or layer_local else None
        if plan is not None:
            done = _awq_lite_layer_local(model, forward_loop, plan, alpha_step=alpha_step, search=search,
                                         tie_margin=tie_margin, tie_check=tie_check)
            if done is not None:
                return done
            plan = None  # the forward loop never called the first decoder layer (it feeds the linears itself): whole-model flow
        elif layer_local:
            raise ValueError("awq_lite(layer_local=True): the model has no decoder stack that holds every searched linear, "
                             "or other quantizers are enabled (their noise belongs to the search pass)")
    stats = AWQ_LITE_STATS
    stats.clear()
    stats.update({"search": search, "passes": 0, "stages_s": {}})
    clock = {"t": None, "dev": None}

    def stage(name):
        """Wall-clock of the flow's stages (tools / bench `extra`): the device is drained only here, at points where the
        host waits for statistics anyway."""
        if clock["dev"] is not None and clock["dev"].type == "cuda":
            torch.cuda.synchronize(clock["dev"])
        now = time.perf_counter()
        if name is not None and clock["t"] is not None:
            stats["stages_s"][name] = round(stats["stages_s"].get(name, 0.0) + now - clock["t"], 4)
        clock["t"] = now

    mods = [(n, m) for n, m in model.named_modules() if _awq_searched(m)]
    # the stages cover the WHOLE call (setup = helpers, weight scales, Gram buffers; teardown = the bookkeeping after the
    # fold), so that  sum(stages_s) == wall-clock of awq_lite  by construction
    clock["dev"] = mods[0][1].weight.device if mods else None
    stage(None)
    helpers = {m: AWQLiteHelper(m, alpha_step) for _, m in mods}
    stage("weight_scales")  # (helper construction: one moq_awq_weight_scale launch and four small buffers per linear)
    for _, m in mods:
        # quantized inputs (W4A8 AWQ; setup, :1436-1444): the input quantizer is bypassed for the whole search -- the
        # losses are taken on UNquantized activations -- and max-calibrated per input channel in the cache pass; the
        # per-tensor amax it ends up with is that of the smoothed activation, max_c(amax_c * pre_quant_scale_c)
        h, iq = helpers[m], m.input_quantizer
        h.is_input_quantized = iq.is_enabled
        if h.is_input_quantized:
            iq.disable()
            if iq.axis not in (None, -1):
                h.is_enabled = False
                h.setup_disabled = True
            else:
                iq.axis = -1
    if store_activations not in (True, False, "auto", "inputs"):
        raise ValueError(f"awq_lite: store_activations must be True, False, 'inputs' or 'auto', got {store_activations!r}")
    store_mode = {True: "full", False: False, "inputs": "inputs"}.get(store_activations, "inputs" if search == "auto" else False)
    state = {"mode": "cache", "do_gemm": True, "do_exact": False, "store": store_mode, "stored_bytes": 0,
             "store_last": None, "store_seen": {}, "store_cap": None}
    if mods:
        budget = _WeightCacheBudget(mods[0][1].weight.device)
        for _, m in mods:
            h = helpers[m]
            cin = m.weight.shape[1]
            # ragged rows (Cin not a multiple of the block: zero-padded, AWQLiteHelper) stay on the error-GEMM engine
            gram_ok = h.fused and cin % 4 == 0 and cin % h.block_size == 0
            fits = search != "gemm" and gram_ok and budget.reserve(4 * cin * cin)
            if fits or (search == "gram" and gram_ok):
                h.gram = torch.zeros(cin, cin, dtype=torch.float32, device=m.weight.device)
                h.gram_bytes = 4 * cin * cin if fits else 0
                h.use_gram = True
            else:
                h._cache_w = budget.reserve(len(h.alphas) * m.weight.numel() * m.weight.element_size())

    stage("gram_buffers")  # (device-memory query of the budget + one zeroed [Cin, Cin] fp32 matrix per Gram-scored linear)
    searched = {id(q) for _, m in mods for q in (m.weight_quantizer, m.input_quantizer)}
    others = [q for q in _quantizers(model) if id(q) not in searched and q.is_enabled]
    others_holder = nn.ModuleList(others)
    # With other quantizers active (FP8 KV cache, ...) the reference's search pass sees THEIR quantization noise in
    # the activations; the Gram matrices are then accumulated in a second pass instead of the cache pass.
    state["gram_pass"] = "search" if others else "cache"
    if others:
        state["store"] = False  # the search pass must SEE the other quantizers' noise: nothing of the cache pass is reusable
    if state["store"] and store_activations == "auto" and mods:
        # nobody asked for the store: in the whole-model flow the kept inputs of EVERY searched linear and batch share the HBM
        # with the forward's own peak activations, which the budget (a fraction of what was free at setup) does not see --
        # the automatic store stops at half of what the budget has left (30 % of the memory that was available, Gram matrices
        # taken off first) and the search pass is then a real forward, where an explicit store_activations=True / "inputs" may
        # take all of it
        state["store_cap"] = budget.left // 2

    def accumulate_gram(h, input, x2):
        h.num_gram_steps += 1
        # Linears fed by the SAME tensor (q / k / v of an attention block, gate / up of an MLP) have the same
        # Gram matrix: the first one accumulates it, the others alias it.  Identity of the tensor object is the
        # test (a reference to the last input is kept, so its address cannot be reused in between).
        owner = state.get("gram_owner") if state.get("gram_input") is input else None
        if owner is not None and owner.gram is not None and owner.gram.shape == h.gram.shape \
                and h.gram_owner in (None, owner):
            if h.gram_owner is None:
                h.gram_owner = owner
                h.gram = owner.gram  # the own buffer is released
                budget.release(h.gram_bytes)
                h.gram_bytes = 0
            return
        if h.gram_owner is not None:
            raise RuntimeError("awq_lite (Gram search): a linear that shared its input with another one in an "
                               "earlier batch got a different tensor now; use search='gemm' for this model")
        if x2.dtype in (torch.bfloat16, torch.float16):
            # G += X^T X / T_b on the matrix cores; several batches per launch when the staging buffer fits
            if h.gram_stage is None and not h.gram_stage_denied:
                if budget.reserve(ops.GramStage.nbytes(x2.shape[1], x2.shape[0])):
                    h.gram_stage = ops.GramStage(h.gram, x2.shape[0], x2.dtype)
                else:
                    h.gram_stage_denied = True
            if h.gram_stage is not None:
                h.gram_stage.add(x2)
            else:
                ops.hessian_accum(h.gram, x2, 1.0, 1.0 / x2.shape[0], upper_only=True)
        else:
            xf = x2.float()
            h.gram.addmm_(xf.t(), xf, alpha=1.0 / x2.shape[0])
        state["gram_input"], state["gram_owner"] = input, h

    def error_gemms(self, h, x2, out2, subset, loss_buf):
        """loss_buf[j] += mean((linear(x / s_a, QDQ(W s_a)) - out_actual)^2) for the candidates `subset` (None: all)
        with the reference's roundings (x/s, the output and out_actual in the model dtype)."""
        inv_s, w_hat = h.search_operands(self, subset)
        if ops.mfma_gemm_supported(x2, self.weight):
            # all candidates in two launches: xs[a] = x * (1/s_a) (one read of x), then the batched MFMA
            # contraction with the (out - out_actual)^2 mean fused -- `out` never reaches HBM
            xs = ops.scale_cols_multi(x2, inv_s)
            ops.awq_err_gemm_multi(xs, w_hat, out2, self.bias, loss_buf)
        else:  # widths the kernels do not take: library GEMM, the reference's own arithmetic
            for j in range(inv_s.shape[0]):
                out = F.linear(ops.scale_cols(x2, inv_s[j]), w_hat[j], self.bias)
                loss_buf[j] += (out - out2).float().pow(2).mean()

    def patched_forward(self, input):
        h = helpers[self]
        out_actual = F.linear(input, self.weight, self.bias)
        if input.numel() == 0 or h.setup_disabled:
            return out_actual
        x2 = input.reshape(-1, input.shape[-1])
        if state["mode"] == "cache":
            # act_scale += mean_tokens |x| in the activation dtype (get_act_scale, :1471-1472), one kernel pass.  Linears
            # fed by the SAME tensor object (q / k / v, gate / up) have the same statistic: the first one of a batch
            # accumulates it, the others alias its buffer (the rule of the Gram matrices below, same check: identity of the
            # tensor object, which is kept alive in between) -- three reads of a 33 MB activation less per decoder layer
            # and batch, bit-identical values
            owner = state.get("act_owner") if state.get("act_input") is input else None
            if owner is not None and owner.act_sum.shape == h.act_sum.shape and h.act_owner in (None, owner):
                if h.act_owner is None:
                    h.act_owner = owner
                    h.act_sum = owner.act_sum
            else:
                if h.act_owner is not None:
                    raise RuntimeError("awq_lite: a linear that shared its input with another one in an earlier batch got a "
                                       "different tensor now")
                if numerics.on_host():
                    ops.col_abs_mean_accum(x2, h.act_sum)
                else:
                    # numerics "device" = the reference's run ON THIS DEVICE, and this statistic is the one place where that
                    # needs torch's own kernel: both sides sum |x| in fp32 and round the mean to the activation dtype, but
                    # in different orders, and a channel whose mean sits on a 16-bit rounding boundary lands on either side
                    # (Llama-3-8B width on the MI355X: 23 of 56 scale / amax vectors one step apart, profiles/r06_dropin.md;
                    # the one-pass kernel is the default mode's, where the result must not depend on the device).  Three
                    # passes over a 33 MB activation instead of one, next to the layer's GEMMs.
                    h.act_sum.add_(x2.abs().mean(0).to(torch.float32))
                state["act_input"], state["act_owner"] = input, h
            h.num_cache_steps += 1
            h.num_tokens += x2.shape[0]
            if h.is_input_quantized:  # :1534-1538: running per-channel amax of the raw input
                iq = self.input_quantizer
                iq.enable()
                max_calibrate(iq, lambda q: q(input), distributed_sync=False)
                iq.disable()
        if state["mode"] == state["gram_pass"] and h.gram is not None and h.is_enabled:
            accumulate_gram(h, input, x2)
        if state["mode"] == "cache" and state["store"] and h.is_enabled:
            store_batch(h, input, x2, out_actual)
        if state["mode"] == "cache" or not h.is_enabled:
            return out_actual
        out2 = out_actual.reshape(-1, out_actual.shape[-1])
        if h.use_gram:
            # scores came from the Gram matrix; only near-ties are re-scored with the reference's arithmetic
            if state["do_exact"] and h.pending:
                error_gemms(self, h, x2, out2, h.pending, h.exact_pass_buf)
                h.num_exact_steps += 1
        elif state["do_gemm"]:
            error_gemms(self, h, x2, out2, None, h.loss_buf)
            h.num_search_steps += 1
        return out_actual

    def store_batch(h, input, x2, out_actual):
        """store_activations: keep this batch's input and out_actual of this linear for the replayed exact pass.  The
        tensors exist anyway -- the store only keeps them alive -- and are charged to the HBM budget (an input shared by
        q / k / v or gate / up once)."""
        if state["store"] == "inputs" and input.is_inference():
            # (forward_loop under torch.inference_mode(): no version counter says whether the tensor is still what the cache
            # pass read when the replay gets to it -- nothing is stored, the search pass is a real forward)
            drop_stores("inference tensors carry no version counter")
            return
        if state["store_cap"] is not None and state["stored_bytes"] + x2.numel() * x2.element_size() > state["store_cap"]:
            drop_stores("the automatic store reached its cap (half of what the HBM budget had left)")
            return
        if state["store"] == "inputs":
            # the input only, charged once per distinct tensor object (kept alive in `store_seen`, so an id cannot be
            # reused); out_actual is recomputed by the replay from the linear's own weight
            nbytes = 0 if state["store_seen"].get(id(input)) is input else x2.numel() * x2.element_size()
            if x2.data_ptr() != input.data_ptr():  # a strided input: the flattened copy is this linear's own
                nbytes = x2.numel() * x2.element_size()
            if not budget.reserve(nbytes):
                drop_stores()
                return
            state["store_seen"][id(input)] = input
            state["stored_bytes"] += nbytes
            h.stored.append((x2, None, input, input._version))
            return
        nbytes = out_actual.numel() * out_actual.element_size()
        if state["store_last"] is not input:
            nbytes += x2.numel() * x2.element_size()
        if not budget.reserve(nbytes):
            drop_stores()  # does not fit: the exact pass will be a real one
            return
        state["store_last"] = input
        state["stored_bytes"] += nbytes
        h.stored.append((x2, out_actual.reshape(-1, out_actual.shape[-1]), None, None))

    def drop_stores(why="no room in the HBM budget"):
        stats.setdefault("store_dropped", why)  # (the first reason: why this call's search pass is a real forward)
        for hh in helpers.values():
            hh.stored = []
        budget.release(state["stored_bytes"])
        state["stored_bytes"], state["store"], state["store_last"] = 0, False, None
        state["store_seen"] = {}

    def stores_intact() -> bool:
        """"inputs" mode: every kept input still holds what the cache pass read (no in-place write since)."""
        return all(src is None or src._version == ver for hh in helpers.values() for _, _, src, ver in hh.stored)

    def replay_search_pass():
        """The search pass from the stored activations, linear by linear: every candidate this pass has to score of a
        linear (its near-ties, or all of them for a linear without a Gram matrix) over all stored batches, then the
        next linear -- one linear's search weights are alive at a time."""
        for _, m in mods:
            h = helpers[m]
            if not h.is_enabled or h.act_scale is None or not h.stored:
                continue
            if h.use_gram:
                if not (state["do_exact"] and h.pending):
                    continue
                subset, buf = h.pending, h.exact_pass_buf
            elif state["do_gemm"]:
                subset, buf = None, h.loss_buf
            else:
                continue
            keep, h._cache_w = h._cache_w, True  # resident for this linear's batches, released right after
            for x2, out2, _, _ in h.stored:
                if out2 is None:  # "inputs": out_actual from the un-folded weight, as the patched forward computes it
                    out2 = F.linear(x2, m.weight, m.bias)
                error_gemms(m, h, x2, out2, subset, buf)
                if h.use_gram:
                    h.num_exact_steps += 1
                else:
                    h.num_search_steps += 1
            h._inv_scale = h._scale_dt = h._w_hat = None
            h._cache_w = keep

    def search_pass():
        if state["store"] == "inputs" and not stores_intact():
            drop_stores("a stored input was written in place after the cache pass")
        if state["store"]:
            replay_search_pass()
            stats["replayed_passes"] = stats.get("replayed_passes", 0) + 1
        else:
            forward_loop(model)
            stats["passes"] += 1

    originals = {}
    for _, m in mods:
        originals[m] = m.__dict__.get("forward", _NO_INSTANCE_FORWARD)
        m.forward = patched_forward.__get__(m, type(m))

    def finish_gram_pass():
        state.pop("gram_input", None)
        state.pop("gram_owner", None)
        state.pop("act_input", None)
        state.pop("act_owner", None)
        for h in helpers.values():
            if h.gram_stage is not None:
                h.gram_stage.flush()
                h.gram_stage = None

    def shard_gram_scoring():
        """Data parallel: instead of every rank scoring every linear on its LOCAL Gram matrix (the scores are linear in
        it and get summed), the distinct Gram matrices are dealt over the ranks -- balanced by the scoring work they
        carry -- each is SUM-reduced to its rank over RCCL (a real exchange: sum of Cin^2 fp32 over the model, 33 GB for
        Llama-3-8B, a few hundred ms over xGMI) and only that rank scores the linears that read it; the other ranks
        contribute zeros to the score bucket.  The scoring (11 quadratic forms per linear, a third of the single-GPU
        search) then scales with the number of GPUs like the accumulation does.  Returns {owner helper: rank} or None
        when the ranks do not see the same sharing structure (then every rank scores locally, as before)."""
        if not _dist_on():
            return None
        hs = [helpers[m] for _, m in mods]
        index = {id(h): i for i, h in enumerate(hs)}
        sig = [None if (h.gram is None or h.act_scale is None or not h.is_enabled)
               else (index[id(h.gram_owner or h)], tuple(h.gram.shape), str(h.gram.dtype)) for h in hs]
        gathered = [None] * dist.get_world_size()
        dist.all_gather_object(gathered, sig)
        if any(g != sig for g in gathered):
            return None
        cost, owners = {}, []
        for (_, m), h, s_ in zip(mods, hs, sig):
            if s_ is None:
                continue
            o = hs[s_[0]]
            if id(o) not in cost:
                cost[id(o)] = 0
                owners.append(o)
            cost[id(o)] += m.weight.shape[0] * m.weight.shape[1] * m.weight.shape[1]
        load = [0] * dist.get_world_size()
        placement = {}
        for o in sorted(owners, key=lambda o: -cost[id(o)]):  # largest first onto the least loaded rank
            r = min(range(len(load)), key=load.__getitem__)
            placement[id(o)] = r
            load[r] += cost[id(o)]
        for o in owners:  # same order on every rank
            mdist.reduce_chunked(o.gram, placement[id(o)], dist.ReduceOp.SUM)  # <= 1 GiB per call
        return placement

    def gram_losses():
        """Gram-matrix linears: all alpha losses from the Gram matrix (the loss is linear in it).  Single process: the
        local matrix.  Data parallel: the reduced matrix on the rank it was dealt to (shard_gram_scoring), or -- when
        the ranks disagree on the sharing structure -- every rank's local matrix, summed later in the score bucket."""
        placement = shard_gram_scoring()
        me = dist.get_rank() if placement is not None else 0
        for _, m in mods:
            h = helpers[m]
            if h.gram is not None and h.act_scale is not None:
                own = h.gram_owner or h
                h.num_search_steps = h.num_gram_steps
                if placement is None or placement.get(id(own)) == me:
                    if m.weight.dtype != torch.float32 and not own.gram_symmetrized:
                        ops.symmetrize(h.gram)  # the MFMA accumulation kept the upper tiles only; once per matrix
                        own.gram_symmetrized = True
                    _gram_losses(h, m)
                    h.scored_here = True
                budget.release(h.gram_bytes)  # an owner's matrix really goes when its last alias has been scored
                h.gram_bytes = 0
                h.gram = None  # release Cin^2 floats as soon as the last linear using them is done

    def pick_contenders() -> bool:
        """After the Gram scores are known: which candidates of which linears need the exact engine.  The decision is
        taken on the scores summed over the data-parallel group, so every rank re-scores the same candidates."""
        scored = [(m, helpers[m]) for _, m in mods if helpers[m].use_gram and helpers[m].is_enabled
                  and helpers[m].act_scale is not None]
        if not scored:
            return False
        distributed = dist.is_available() and dist.is_initialized()
        if distributed:
            steps = torch.tensor([float(h.num_search_steps) for _, h in scored] + [float(h.num_tokens) for _, h in scored],
                                 device=scored[0][0].weight.device)
            mdist.all_reduce_bucket([h.loss_buf for _, h in scored] + [steps], dist.ReduceOp.SUM)
            counts = steps.tolist()
            for (_, h), n, t in zip(scored, counts[:len(scored)], counts[len(scored):]):
                h.search_steps_all_ranks = int(n)
                h.tokens_all_ranks = int(t)
                h.loss_synced = True
        if search != "auto":
            return False
        table = torch.stack([h.loss_buf for _, h in scored]).cpu()  # one host sync for all linears
        any_tie = False
        for (m, h), row in zip(scored, table.tolist()):
            h.gram_loss = list(row)
            margin = tie_margin
            if margin is None:
                outputs = max(1, getattr(h, "tokens_all_ranks", h.num_tokens) * m.weight.shape[0])
                margin = GRAM_TIE_MARGIN.get(m.weight.dtype, 5e-3) + GRAM_TIE_NOISE / math.sqrt(outputs)
                if gram_score_planes(m.weight.dtype) < 3 and m.weight.dtype != torch.float32:
                    margin += GRAM_PLANES_SLACK
            best = min(row)  # a NaN score never compares smaller: such a linear keeps the plain first-minimum rule
            if not math.isfinite(best):
                continue
            h.margin_used = margin
            close = [i for i, v in enumerate(row) if v <= best * (1.0 + margin)]
            if len(close) > 1:
                queue_exact(m, h, close)
                any_tie = True
        return any_tie

    def queue_exact(m, h, indices):
        """The next exact pass scores these candidates of this linear."""
        h.pending = list(indices)
        h.exact_pass_buf = torch.zeros(len(indices), dtype=torch.float32, device=m.weight.device)
        h.num_exact_steps = 0
        if h._cache_w and h._w_hat is not None:
            budget.release(h._w_hat.numel() * h._w_hat.element_size())
        h._inv_scale = h._scale_dt = h._w_hat = None
        # (a replayed pass keeps one linear's search weights alive at a time: nothing to reserve per linear)
        h._cache_w = (not state["store"]) and budget.reserve(len(indices) * m.weight.numel() * m.weight.element_size())

    def collect_exact() -> bool:
        """After an exact pass: the scores of the pending candidates (summed over the data-parallel group, so every
        rank takes the same decisions) move to `exact_scores`; with tie_check every re-scored linear's margin is
        verified against the measured disagreement of the two engines and widened if it was too small.  Returns whether
        another exact pass is needed."""
        todo = [(m, helpers[m]) for _, m in mods if helpers[m].pending]
        if not todo:
            return False
        dev = todo[0][0].weight.device
        steps = torch.tensor([float(h.num_exact_steps) for _, h in todo], device=dev)
        if dist.is_available() and dist.is_initialized():
            mdist.all_reduce_bucket([h.exact_pass_buf for _, h in todo] + [steps], dist.ReduceOp.SUM)
        flat = torch.cat([h.exact_pass_buf for _, h in todo] + [steps]).cpu().tolist()  # one host sync
        counts = flat[len(flat) - len(todo):]
        off, again = 0, False
        for (m, h), n_steps in zip(todo, counts):
            vals = flat[off:off + len(h.pending)]
            off += len(h.pending)
            pend, h.pending, h.exact_pass_buf = h.pending, None, None
            if h._cache_w and h._w_hat is not None:
                budget.release(h._w_hat.numel() * h._w_hat.element_size())
            h._inv_scale = h._scale_dt = h._w_hat = None
            h._cache_w = False
            h.exact_steps_all_ranks = int(n_steps)
            if n_steps <= 0:
                continue  # the second pass never reached this linear: the Gram scores stand
            h.exact_scores.update(zip(pend, vals))
            if not tie_check or tie_margin is not None and not math.isfinite(tie_margin):
                continue
            new = check_margin(h)
            if new:
                queue_exact(m, h, new)
                again = True
        return again

    def check_margin(h):
        """The self-check of the re-scoring margin (tie_margin_check): candidates to add, or [] when settled."""
        need, margin, new = tie_margin_check(h.gram_loss, h.exact_scores, h.margin_used, h.tie_rounds)
        # (None: nothing left to weigh -- every candidate of the linear has its exact score, or the scores are not finite)
        h.tie_need = need
        if new:
            h.tie_rounds += 1
        h.margin_used = margin  # (also when it grew without admitting anybody: that is the margin the linear is settled at)
        return new

    try:
        # every OTHER enabled quantizer (KV-cache bmm quantizers, linears outside the search, ...) collects its amax
        # during the cache pass and quantizes during the search pass, as in the reference (:1574-1586); dynamic ones
        # are switched to pass-through for the cache pass
        enable_stats_collection(others_holder)
        stage("setup")
        forward_loop(model)  # cache pass
        stats["passes"] += 1
        state.pop("act_input", None)  # (the last batch's input need not stay alive)
        state.pop("act_owner", None)
        if state["store"] == "inputs":
            stats["stored_input_bytes"] = state["stored_bytes"]
        stage("cache_pass")
        finish_stats_collection(others_holder)
        if others and dist.is_available() and dist.is_initialized():
            mdist.sync_amax_bucketed([q for q in others if not q._dynamic],
                                     device=mods[0][1].weight.device if mods else None)
        if dist.is_available() and dist.is_initialized():
            # the per-channel input amax of W4A8 linears is synchronised like any other amax (:1581-1586, :396-398)
            chan = [m.input_quantizer for _, m in mods if helpers[m].is_input_quantized]
            if chan:
                mdist.sync_amax_bucketed(chan, device=mods[0][1].weight.device)
        if state["gram_pass"] == "cache":
            finish_gram_pass()
        cached = [h for h in helpers.values() if h.num_cache_steps]
        if cached and not numerics.on_host():
            # act_scale = act_sum / steps (:1601) as torch evaluates `tensor / int` on the statistics' device
            for h in cached:
                h.act_scale = h.act_sum.detach().float() / h.num_cache_steps
        elif cached:
            # act_scale = act_sum / steps (:1601) as IEEE division on the host (see get_scale) -- for ALL linears in one
            # device -> host copy and one upload instead of a stream drain per linear
            flat = torch.cat([h.act_sum.detach().float().reshape(-1) for h in cached]).cpu()
            with host_math_threads():
                steps = torch.cat([torch.full((h.act_sum.numel(),), float(h.num_cache_steps)) for h in cached])
                flat = (flat / steps).to(cached[0].act_sum.device)
            off = 0
            for h in cached:
                h.act_scale = flat[off:off + h.act_sum.numel()]
                off += h.act_sum.numel()
        if mods:
            # DP: act_scale average + the any-NaN vote for ALL linears in ONE bucket (reference: one all_reduce and
            # one object gather per linear, :1588-1619); ranks whose shard never reached a linear join with zeros
            hs = [helpers[m] for _, m in mods]
            synced, enabled = mdist.sync_awq_act_scales([h.act_scale for h in hs], [h.weight_scale for h in hs],
                                                        [m.weight.shape[1] for _, m in mods], mods[0][1].weight.device)
            for h, a, ok in zip(hs, synced, enabled):
                h.is_enabled = ok
                h.act_scale = a if ok else None
                if not ok:
                    h.gram = None
                    h.use_gram = False
            # every candidate scale vector of every linear: ONE device -> host copy of all statistics, the [Cin]-sized
            # math on the host (get_scale), one upload per linear -- all before the first scoring kernel is queued
            live = [(m, helpers[m]) for _, m in mods if helpers[m].act_scale is not None]
            if live and not numerics.on_host():
                for m, h in live:  # the same tables from device tensors (numerics "device": no copy, no host math)
                    h.prepare_scales(h.act_scale, h.weight_scale, m.weight.dtype)
            elif live:
                flat = torch.cat([t.detach().float().reshape(-1) for _, h in live for t in (h.act_scale, h.weight_scale)]).cpu()
                jobs, off = [], 0
                for m, h in live:
                    c = h.act_scale.numel()
                    jobs.append((h, flat[off:off + c], flat[off + c:off + 2 * c], m.weight.dtype))
                    off += 2 * c
                # 11 pow / divide / normalise chains per linear on [Cin] vectors, one thread (see host_math_threads)
                with host_math_threads():
                    tables = [j[0].host_scales(j[1], j[2], j[3]) for j in jobs]
                for (h, _, _, _), both in zip(jobs, tables):
                    h.upload_scales(both)
        stage("scales")
        if state["gram_pass"] == "cache":
            gram_losses()
            state["do_exact"] = pick_contenders()
        stage("gram_scores")
        need_gemm = any(not h.use_gram and h.act_scale is not None for h in helpers.values())
        need_gram = state["gram_pass"] == "search" and any(h.gram is not None for h in helpers.values())
        if need_gemm or need_gram or state["do_exact"]:
            state["mode"] = "search"
            # search pass: error GEMMs (all candidates of the linears without a Gram matrix, the near-ties of the
            # others), and the Gram matrices when they had to wait for quantized inputs -- replayed from the stored
            # activations when the call keeps them (store_activations), a second forward otherwise
            if need_gram:
                drop_stores("the Gram matrices come from the search pass") if state["store"] else None  # (Gram matrices from the search pass need the real forward)
            search_pass()
            stage("search_pass")
            state["do_gemm"] = False
            if need_gram:
                finish_gram_pass()
                gram_losses()
                state["gram_pass"] = None
                state["do_exact"] = pick_contenders()
                stage("gram_scores")
            elif state["do_exact"]:
                state["do_exact"] = collect_exact()
            while state["do_exact"]:
                # the near-ties of linears whose Gram matrix came from the search pass; candidates admitted by the
                # margin's self-check (tie_check)
                search_pass()
                stage("search_pass")
                state["do_exact"] = collect_exact()
        for h in helpers.values():
            if not getattr(h, "loss_synced", False):
                h.search_steps_all_ranks = h.num_search_steps
        if dist.is_available() and dist.is_initialized() and mods:
            # every rank must pick the same alpha -- and take the same "was it searched at all" decision: SUM the
            # per-alpha losses and the search-step counters in one bucket (Gram-scored linears were summed when the
            # near-ties were picked, their exact scores after every exact pass: collect_exact)
            dev = mods[0][1].weight.device
            late = [h for h in helpers.values() if not getattr(h, "loss_synced", False)]
            steps = torch.tensor([float(h.num_search_steps) for h in late], device=dev)
            mdist.all_reduce_bucket([h.loss_buf for h in late] + [steps], dist.ReduceOp.SUM)
            for h, n in zip(late, steps.tolist()):
                h.search_steps_all_ranks = int(n)
    finally:
        for m, f in originals.items():
            _restore_forward(m, f)
        for h in helpers.values():
            h.release()
            h.gram = None
            h.stored = []
        state["store_seen"] = {}

    def restore_input_quantizer(m, h):
        """:1642-1653 / :1707-1714: the per-channel amax is kept (on the host) for the smoothing step and collapses to
        the per-tensor amax the quantizer is exported with; a dynamic input quantizer is just re-enabled."""
        iq = m.input_quantizer
        if not h.is_input_quantized:
            return
        if iq.amax is not None:
            act_amax = iq.amax
            iq._amax_for_smoothing = act_amax.cpu()
            iq.reset_amax()
            iq.axis = None
            iq.amax = act_amax.amax()
        iq.enable()

    for name, m in mods:
        h = helpers[m]
        if h.is_enabled and h.search_steps_all_ranks == 0:
            h.is_enabled = False  # :1665-1672
            warnings.warn("awq_lite: Calling `forward_loop(model)` the second time did not forward data through the "
                          f"{name}. Please provide a valid `forward_loop` function that can be used to forward data "
                          "through the model many times.")
        if not h.is_enabled:
            # :1674-1700: uncalibrated / NaN / never searched -> max-calibrated weights and a neutral pre_quant_scale,
            # so that every linear of the model exports in the same format
            warnings.warn(f"awq_lite: Forcing pre_quant_scale=1 for {name} because the expert was not properly "
                          "exercised during calibration. This may degrade accuracy; consider increasing calibration "
                          "size or using a more diverse dataset.")
            m.awq_lite = h
            m.weight_quantizer.reset_amax()
            max_calibrate(m, lambda lin: lin.weight_quantizer(lin.weight), distributed_sync=False)
            m.input_quantizer._enable_pre_quant_scale = True
            m.input_quantizer.pre_quant_scale = torch.ones(m.weight.shape[1], dtype=m.weight.dtype,
                                                           device=m.weight.device)
            restore_input_quantizer(m, h)
            continue
        if h.exact_scores:
            # near-ties: the exact scores replace the Gram scores of the re-scored candidates and decide among them
            # (ascending alpha, first minimum: the reference's dict order, :1637)
            h.contenders = sorted(h.exact_scores)
            exact = [h.exact_scores[i] for i in h.contenders]
            h.exact_buf = torch.tensor(exact, dtype=torch.float32, device=h.loss_buf.device)
            h.loss_buf[torch.tensor(h.contenders, device=h.loss_buf.device)] = h.exact_buf
            losses = {a: float(v) for a, v in h.loss.items()}
            h.best_alpha = h.alphas[h.contenders[min(range(len(exact)), key=exact.__getitem__)]]
        else:
            losses = {a: float(v) for a, v in h.loss.items()}
            h.best_alpha = min(losses, key=losses.get)  # first minimal alpha (:1637)
        best_idx = h.alphas.index(h.best_alpha)
        h.best_scale = h.scale(h.best_alpha)
        m.awq_lite = h
        # postprocess (:1636-1659) -> apply_pre_quant_scale_and_smooth(module, 1 / best_scale) (:1226-1252): the input
        # gets 1/s in the weight dtype; the weight is multiplied (fp32, one rounding) by 1 / (1/s) -- the fp32 double
        # reciprocal, which is not always s itself -- and recalibrated
        pqs_host = 1.0 / h._s_host[best_idx]  # where every scale vector was formed (get_scale: host IEEE fp32, or the device)
        both = torch.stack([pqs_host, 1.0 / pqs_host]).to(m.weight.device)
        pre_quant_scale = both[0]
        ops.scale_cols(m.weight.data, both[1], out=m.weight.data)
        m.weight_quantizer.reset_amax()
        max_calibrate(m, lambda lin: lin.weight_quantizer(lin.weight), distributed_sync=False)
        m.input_quantizer._enable_pre_quant_scale = True
        m.input_quantizer.pre_quant_scale = pre_quant_scale.to(m.weight.dtype)
        restore_input_quantizer(m, h)
        iq = m.input_quantizer
        if h.is_input_quantized and iq.amax is not None:
            # :1257-1265: amax of the smoothed activation, the product taken in the weight dtype
            smooth = iq._amax_for_smoothing.to(device=m.weight.device, dtype=m.weight.dtype)
            iq.amax = (smooth * pre_quant_scale.to(m.weight.device)).amax().to(m.weight.dtype)
    stage("fold")
    resc = [h for h in helpers.values() if h.contenders is not None]
    checked = [h for h in resc if h.tie_need is not None and h.margin_used and math.isfinite(h.margin_used)]
    stats.update({"linears": len(mods), "rescored_linears": len(resc),
                  "rescored_candidates": sum(len(h.contenders) for h in resc),
                  "tie_check": {"enabled": bool(tie_check), "spread_factor": TIE_SPREAD_FACTOR,
                                "checked_linears": len(checked),
                                "widened_linears": sum(1 for h in resc if h.tie_rounds),
                                "max_need_over_margin": round(max((h.tie_need / h.margin_used for h in checked),
                                                                  default=0.0), 4),
                                "max_need": max((h.tie_need for h in checked), default=0.0)}})
    stage("teardown")
    return helpers


# ------------------------------------------------------------------------------------------------ AWQ clip
class AWQClipHelper:
    """Per-linear state of awq_clip (model_calib.py:1746-1785): the max-calibrated block amax, one block-loss
    table for all clip ratios (device fp32 [K, nblk, Cout], the layout the kernel writes) and the result."""

    def __init__(self, module: QuantLinear, min_clip_ratio: float, shrink_step: float):
        wq = module.weight_quantizer
        self.num_tokens = 0
        self.block_size = wq.block_sizes.get(-1, None) or wq.block_sizes.get(module.weight.dim() - 1)
        wq.reset_amax()  # cache the original amax (:1751-1756)
        enable_stats_collection(wq)
        wq(module.weight)
        finish_stats_collection(wq)
        self.w_amax = wq.amax.clone()
        co, ci = module.weight.shape
        # same float keys as :1759-1761
        self.clip_ratios = [round(float(k), 2) for k in torch.arange(min_clip_ratio, 1.0, shrink_step)] + [1.0]
        self.nblk = math.ceil(ci / self.block_size)
        self.shrinks = torch.tensor(self.clip_ratios, dtype=torch.float32, device=module.weight.device)
        self.loss_buf = torch.zeros(len(self.clip_ratios), self.nblk, co, dtype=torch.float32,
                                    device=module.weight.device)
        self.best_clip_val = None
        self.best_loss = None
        self.is_input_quantized = module.input_quantizer.is_enabled
        wq.disable()

    @property
    def loss(self):
        """{clip ratio: fp32 [Cout, nblk]} -- the reference's per-shrink dict view of the loss table."""
        return {k: self.loss_buf[i].t() for i, k in enumerate(self.clip_ratios)}

    def update_best_params(self):
        """model_calib.py:1788-1798: first strictly smaller loss wins; clip value = w_amax * shrink in
        w_amax's dtype."""
        self.best_loss = torch.ones_like(self.w_amax) * float("inf")
        self.best_clip_val = torch.zeros_like(self.w_amax)
        for shrink, loss in self.loss.items():
            loss = loss.reshape(self.w_amax.shape)
            indices = loss < self.best_loss
            self.best_loss = torch.where(indices, loss.to(self.best_loss.dtype), self.best_loss)
            self.best_clip_val = torch.where(indices, self.w_amax * shrink, self.best_clip_val)


@torch.no_grad()
def awq_clip(model: nn.Module, forward_loop, max_co_batch_size: int = 1024, max_tokens_per_batch: int = 64,
             min_clip_ratio: float = 0.5, shrink_step: float = 0.05, debug: bool = False, **kwargs):
    """AWQ-clip (model_calib.py:1724-1940) for static-block INT weight quantizers: per weight block, the clip
    ratio of the block amax that minimises the block's output error on (sub-sampled) calibration tokens.
    `max_co_batch_size` bounds the reference's broadcast temporaries and has no effect here: the search is one
    kernel pass over the weight per forward call (ops.awq_clip_loss)."""
    assert forward_loop is not None, "forward_loop must be provided for awq_clip"
    mods = [(n, m) for n, m in model.named_modules()
            if is_quantized_linear(m) and m.weight_quantizer.is_enabled and m.weight_quantizer.block_sizes is not None]
    for name, m in mods:
        wq = m.weight_quantizer
        if not wq.is_static_block_quant or not isinstance(wq._num_bits, int) or wq._unsigned or wq._narrow_range:
            raise ValueError(f"awq_clip: {name}: only signed static-block INT weight quantizers are on this path "
                             "(the per-tensor NVFP4 branch, model_calib.py:1804-1813, is outside it)")
    helpers = {m: AWQClipHelper(m, min_clip_ratio, shrink_step) for _, m in mods}

    def patched_forward(self, input):
        h = helpers[self]
        iq = self.input_quantizer
        if h.is_input_quantized:  # :1873-1876: calibrate the input quantizer on this batch, then bypass it
            iq.enable()
            max_calibrate(iq, lambda q: q(input), distributed_sync=False)
            iq.disable()
        x = iq(input)  # applies pre_quant_scale (awq_full: the AWQ-lite scale) even though disabled
        x2 = x.reshape(-1, x.shape[-1])
        if x2.shape[0] > 0:
            step = max(1, x2.shape[0] // max_tokens_per_batch)  # inputs[0::step] (:1820)
            h.num_tokens += -(-x2.shape[0] // step)
            ops.awq_clip_loss(x2, self.weight, h.w_amax, h.shrinks, h.block_size, self.weight_quantizer.num_bits,
                              h.loss_buf, token_step=step)
        return F.linear(x, self.weight, self.bias)  # _forward_no_awq with the weight quantizer disabled (:1893)

    originals = {}
    for _, m in mods:
        originals[m] = m.__dict__.get("forward", _NO_INSTANCE_FORWARD)
        m.forward = patched_forward.__get__(m, type(m))
    try:
        enable_stats_collection(model)  # input / KV quantizers collect during the same pass (:1910-1913)
        forward_loop(model)
        finish_stats_collection(model)
        if dist.is_available() and dist.is_initialized():
            # DP: the reference all-reduces every block-loss tensor inside the search loop (:1861-1863); linear
            # in the loss, so ONE bucketed SUM + divide at the end is the same value
            mdist.all_reduce_bucket([h.loss_buf for h in helpers.values()], dist.ReduceOp.SUM, average=True)
            mdist.sync_amax_bucketed([q for q in _quantizers(model) if q.is_enabled])
    finally:
        for m, f in originals.items():
            _restore_forward(m, f)
    for _, m in mods:
        h = helpers[m]
        if h.num_tokens > 0:  # postprocess (:1921-1929)
            h.update_best_params()
            m.weight_quantizer.amax = h.best_clip_val
            m.weight_quantizer.enable()
            if h.is_input_quantized:
                m.input_quantizer.enable()
        if debug:
            m.awq_clip = h
    return helpers


@torch.no_grad()
def awq(model: nn.Module, forward_loop=None, algorithm: str = "awq_lite", **kwargs):
    """model_calib.py:1362-1391: awq_lite, awq_clip or both (awq_full)."""
    out = {}
    with SequentialQuantizer.convert_to_single_quantizer(model):  # search on the first (INT4) stage only (:1378)
        if algorithm in ("awq_full", "awq_lite"):
            lite_kw = {k: v for k, v in kwargs.items() if k in ("alpha_step", "search", "tie_margin", "tie_check", "layer_local", "store_activations")}
            out["awq_lite"] = awq_lite(model, forward_loop, **lite_kw)
        if algorithm in ("awq_full", "awq_clip"):
            clip_kw = {k: v for k, v in kwargs.items()
                       if k in ("max_co_batch_size", "max_tokens_per_batch", "min_clip_ratio", "shrink_step", "debug")}
            out["awq_clip"] = awq_clip(model, forward_loop, **clip_kw)
    for m in model.modules():  # every stage of a sequential weight quantizer is re-calibrated on the final weight (:1385-1391)
        if is_quantized_linear(m) and isinstance(m.weight_quantizer, SequentialQuantizer):
            max_calibrate(m, lambda linear: linear.weight_quantizer(linear.weight), distributed_sync=False)
    return out


def gptq(model: nn.Module, forward_loop, perc_damp: float = 0.01, block_size: int = 128, fused: bool = False, **kwargs):
    """model_calib.gptq (model_calib.py:2192-2271): the implementation lives in gptq.py."""
    from . import gptq as _gptq

    return _gptq.gptq(model, forward_loop, perc_damp=perc_damp, block_size=block_size, fused=fused, **kwargs)

