"""Cross-rank synchronisation of calibration statistics -- ONE bucketed collective per reduce-op type.

The reference issues one small all-reduce per quantizer (model_calib.py:390-407 -> tensor_quantizer.py:
1373-1385; ~450 for Llama-3-8B) and does not synchronise histograms at all (calib/histogram.py:158-163).
On MI355X these payloads (KB..MB) are latency-bound over xGMI, so all amax values travel in one flat fp32
bucket (MAX), all histograms in one int64 bucket (SUM), all AWQ act-scales / losses in one fp32 bucket
(SUM, then / world).  torch.distributed's "nccl" backend is RCCL on ROCm; the same code runs on gloo/CPU,
which is how the N > 1 path is tested without GPUs (tests/test_distributed_cpu.py).
"""

from __future__ import annotations

import torch
import torch.distributed as dist


def _initialized(group=None) -> bool:
    return dist.is_available() and dist.is_initialized()


def active(group=None) -> bool:
    """Is there a process group whose collectives matter?  More than one rank -- or MOQ_FORCE_DIST=1, which lets a
    single rank walk every collective call site (tests/test_gpu_dist_nccl.py: RCCL accepts a world of one, so the first
    8-GPU run is not the first time these calls meet the nccl backend)."""
    import os

    return _initialized(group) and (dist.get_world_size(group) > 1 or os.environ.get("MOQ_FORCE_DIST") == "1")


# ------------------------------------------------------------------------------------------------ replicas
# Weight-side work (abs-max of the weights, 2:4 masks, SmoothQuant fold, MX QDQ, export packing) is independent per
# tensor, so it can be dealt over the ranks -- but ONLY when every rank holds the same weights (pure data-parallel
# replicas).  Under tensor parallelism or FSDP the ranks hold different shards and each must process all of its own
# tensors.  A process group says nothing about which of the two it is, so sharding is something the caller states:
# `declare_data_parallel(group)` once (or `shard_weights=True` per call); nothing is sharded otherwise.
_REPLICAS = {"declared": False, "group": None}


def declare_data_parallel(group=None, enabled: bool = True):
    """State that the ranks of `group` (default: the world) are replicas holding IDENTICAL weights.  From here on the
    weight-side passes of max_calibrate / smoothquant / sparsify / quantize_weights / export deal their tensors over
    the ranks (`shard_list`) unless a call says otherwise."""
    _REPLICAS["declared"], _REPLICAS["group"] = bool(enabled), group if enabled else None


def replicas_declared() -> bool:
    return bool(_REPLICAS["declared"]) and active(_REPLICAS["group"])


def replica_group():
    return _REPLICAS["group"]


def resolve_shard(shard_weights) -> bool:
    """shard_weights argument of the weight-side flows: True / False are taken literally (True needs a process group of
    more than one rank to mean anything), None follows declare_data_parallel."""
    if shard_weights is None:
        return replicas_declared()
    return bool(shard_weights) and active(_REPLICAS["group"])


def all_reduce_bucket(tensors, op, group=None, average: bool = False):
    """All-reduce a list of tensors as one flat buffer (per dtype), results written back in place."""
    if not tensors or not _initialized(group):
        return
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault((t.dtype, t.device), []).append(t)
    world = dist.get_world_size(group)
    for (_, _), ts in by_dtype.items():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.all_reduce(flat, op=op, group=group)
        if average:
            flat = flat / world
        off = 0
        for t in ts:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n


def _agree_amax_membership(quantizers, group, on_missing: str, device=None):
    """Every rank must put the SAME tensors into the bucket.  A quantizer that saw no data on this rank (a routed
    expert that received no tokens from this rank's calibration shard) has no `_amax` here but may have one
    elsewhere: per-quantizer all-reduces as the reference issues them would then dead-lock (its MoE check,
    model_calib.py:226-245, turns that into an error when expert parallelism is on).  One object all-gather of
    (shape, dtype) per quantizer -- control plane, once per calibration -- settles membership:
      on_missing="adopt": ranks without the amax join with zeros (abs-max identity) and end up with the group's
                          value -- the calibration set as a whole did exercise the quantizer;
      on_missing="raise": the reference's "MoE calibration incomplete" RuntimeError."""
    local = [None if getattr(q, "_amax", None) is None else (tuple(q._amax.shape), str(q._amax.dtype).split(".")[-1])
             for q in quantizers]
    gathered = [None] * dist.get_world_size(group)
    dist.all_gather_object(gathered, local, group=group)
    if device is None:
        device = next((q._amax.device for q in quantizers if getattr(q, "_amax", None) is not None), None)
    for i, q in enumerate(quantizers):
        present = [g[i] for g in gathered if g[i] is not None]
        if not present or len(present) == len(gathered):
            if present and len(set(present)) != 1:
                raise RuntimeError(f"amax of quantizer #{i} differs in shape / dtype across ranks: {set(present)}")
            continue
        if on_missing == "raise":
            raise RuntimeError("MoE calibration incomplete: some experts received no tokens during calibration. "
                               "Increase --calib-size to ensure all experts see calibration data.")
        if local[i] is None:
            shape, dt = present[0]
            dev = device
            if dev is None and isinstance(q, torch.nn.Module):
                dev = next((b.device for b in q.buffers()), None)
            zeros = torch.zeros(shape, dtype=getattr(torch, dt), device=dev if dev is not None else "cpu")
            # (a quantizer module takes it through its `amax` data descriptor -- property or buffer-state -- so that the
            # zeros become a REGISTERED buffer; a bare attribute would not travel with state_dict / .to())
            if isinstance(q, torch.nn.Module) and hasattr(getattr(type(q), "amax", None), "__set__"):
                q.amax = zeros
            else:
                q._amax = zeros


def sync_amax_bucketed(quantizers, group=None, on_missing: str = "adopt", device=None):
    """MAX-reduce every calibrated `_amax` buffer of `quantizers` in one collective
    (semantics of sync_amax_across_distributed_group, tensor_quantizer.py:1373-1385)."""
    if not _initialized(group):
        return
    quantizers = list(quantizers)
    _agree_amax_membership(quantizers, group, on_missing, device)
    bufs = [q._amax for q in quantizers if getattr(q, "_amax", None) is not None]
    # NaN must survive the reduction like it does locally: MAX drops NaN on some backends, so carry a flag
    if not bufs:
        return
    nan_flags = torch.stack([torch.isnan(b).any().to(torch.float32) for b in bufs])
    f32 = [b.float() for b in bufs]
    all_reduce_bucket(f32 + [nan_flags], dist.ReduceOp.MAX, group)
    for b, r, flag in zip(bufs, f32, nan_flags.tolist()):
        b.copy_(torch.full_like(r, float("nan")) if flag else r)


def sync_awq_act_scales(act_scales, weight_scales, widths, device, group=None):
    """Data-parallel step of awq_lite after the cache pass (model_calib.py:1588-1619) for ALL linears in one
    collective.  act_scales[i] is this rank's mean-|x| vector of linear i or None (no tokens here); widths[i] = Cin.
    Returns (synced, enabled): synced[i] = average over the ranks that HAVE a scale (None when no rank has one);
    enabled[i] = False when any rank saw NaN in its act or weight scale (the reference's any-NaN vote) or when no
    rank has data.  The reference averages with ReduceOp.AVG assuming every rank has every linear -- the same
    answer whenever that holds."""
    n = len(act_scales)
    flags = torch.zeros(2 * n, dtype=torch.float32, device=device)  # [has..., nan...]
    vecs = []
    for i, (a, w) in enumerate(zip(act_scales, weight_scales)):
        if a is not None:
            flags[i] = 1.0
            bad = torch.isnan(a).any() | torch.isnan(w).any()
            flags[n + i] = bad.to(torch.float32)
            vecs.append(torch.where(torch.isnan(a), torch.zeros_like(a), a).float().clone())
        else:
            vecs.append(torch.zeros(widths[i], dtype=torch.float32, device=device))
    if _initialized(group):
        all_reduce_bucket(vecs + [flags], dist.ReduceOp.SUM, group)
    has, nan = flags[:n].tolist(), flags[n:].tolist()
    synced = [v / h if h > 0 else None for v, h in zip(vecs, has)]
    enabled = [h > 0 and b == 0 for h, b in zip(has, nan)]
    return synced, enabled


def sync_calibrators_bucketed(calibrators, group=None):
    """Synchronise raw calibrator state before compute_amax: running maxima (MAX) and histograms (SUM).

    Histograms: every rank binned with the SAME width (rank 0's first-batch range, `agree_histogram_range` at the
    first collect) and grew its range independently, so the histograms differ only in length: the ranks agree on the
    largest abs-max any of them extended its range for (MAX, one small collective), grow to it by the reference's own
    rule, and SUM the int64 counts in one bucket.  Contract: a histogram-calibrated quantizer is reached by every rank or by none (dense models)."""
    from .calib import HistogramCalibrator, MaxCalibrator

    if not _initialized(group):
        return
    maxes = [c._buf for c in calibrators if isinstance(c, MaxCalibrator) and c._buf is not None]
    all_reduce_bucket(maxes, dist.ReduceOp.MAX, group)
    hists = [c for c in calibrators if isinstance(c, HistogramCalibrator) and c._calib_hist is not None]
    if hists:
        # common range: the largest abs-max any rank extended its range for; ranks below it grow by the same rule
        # (HistogramCalibrator._grow_to), which leaves every rank with the bins and edges a single rank would have
        dev = hists[0]._calib_hist.device
        tops = torch.stack([(c._grown_to if c._grown_to is not None else torch.zeros(())).float().reshape(()) for c in hists]).to(dev)
        dist.all_reduce(tops, op=dist.ReduceOp.MAX, group=group)
        for c, top in zip(hists, tops.cpu()):
            if top > c._calib_bin_edges[-1]:
                c._grow_to(top)
        all_reduce_bucket([c._calib_hist for c in hists], dist.ReduceOp.SUM, group)


def agree_histogram_range(x_max_local: torch.Tensor, group=None, src: int = 0) -> torch.Tensor:
    """First-batch range agreement: every rank bins with the width of rank `src`'s first batch (a broadcast), so that
    all counts can be SUM-reduced exactly AND the merged histogram is the one a single rank builds when it starts
    with that batch (the growth rule keeps the first width, calib/histogram.py:121-127)."""
    if _initialized(group):
        dist.broadcast(x_max_local, src=src, group=group)
    return x_max_local


def _leaf_quantizers(q):
    """The TensorQuantizers behind a (Sequential / Grouped) quantizer container."""
    if hasattr(q, "_amax") or not isinstance(q, torch.nn.ModuleList | torch.nn.Sequential):
        return [q]
    return [leaf for member in q for leaf in _leaf_quantizers(member)]


def sync_amax_tensor_parallel(model, group, column_parallel, row_parallel):
    """The tensor-parallel amax rules of max_calibrate (model_calib.py:408-485) for a model whose linears are sharded
    by the caller: "the quantization parameters when TP = 8 then changed to TP = 4 then back to TP = 8 should be the same".

      column parallel (weights split along Cout): input AND weight amax are shared over the group when the quantizer's
          axis is None or -1 (per-tensor, or per input channel -- every rank sees the whole input);
      row parallel (weights split along Cin):     input amax shared when axis is None; weight amax when axis is None or 0
          (per-tensor, or per output channel -- every rank holds a slice of every row);
      block-quantized quantizers are left alone (INT4 / W4A8 blocks are local; block_sizes type "dynamic" has no amax);
      scalar KV-cache quantizers (k_bmm / v_bmm) are shared.

    `column_parallel(name, module)` / `row_parallel(name, module)` tell which linears are which (this package wraps plain
    nn.Linear / HF modules and has no parallel_state of its own).  ONE bucketed MAX for everything."""
    from .nn import is_quantized_linear

    if not _initialized(group):
        return []
    picked = []

    def take(q, axes):
        for leaf in _leaf_quantizers(q):
            if getattr(leaf, "block_sizes", None) is not None:
                continue
            if getattr(leaf, "_amax", None) is not None and leaf.axis in axes:
                picked.append(leaf)

    for name, m in model.named_modules():
        if is_quantized_linear(m):
            if column_parallel(name, m):
                take(m.input_quantizer, (None, -1))
                take(m.weight_quantizer, (None, -1))
            elif row_parallel(name, m):
                take(m.input_quantizer, (None,))
                take(m.weight_quantizer, (None, 0))
        for attr in ("k_bmm_quantizer", "v_bmm_quantizer"):
            q = getattr(m, attr, None)
            if q is not None and getattr(q, "_amax", None) is not None and q._amax.numel() == 1:
                picked.append(q)
    sync_amax_bucketed(picked, group=group, on_missing="raise")
    return picked


_REPLICA_GROUP = object()  # default of `group` below: whatever declare_data_parallel named (None = the world)


def shard_list(items, rank: int | None = None, world: int | None = None, group=_REPLICA_GROUP):
    """Round-robin shard of per-layer weight tensors (or calibration batches) over the ranks of the replica group:
    independent units, no data-path collective.  Rank and size are those of `group` (default: the group
    given to declare_data_parallel, else the world) -- the same numbering `owner_rank` / `broadcast_from_owners` use, so
    that under a DP subgroup of a larger world every unit has exactly one owner INSIDE the group."""
    if group is _REPLICA_GROUP:
        group = replica_group()
    if rank is None:
        rank = dist.get_rank(group) if _initialized() else 0
    if world is None:
        world = dist.get_world_size(group) if _initialized() else 1
    if rank < 0:
        raise RuntimeError("shard_list: this rank is not a member of the replica group it was asked to shard over")
    return [it for i, it in enumerate(items) if i % world == rank]


# ------------------------------------------------------------------------------------------------ weight-side shards
# Largest single collective call (bytes).  The payloads here are either KB-sized statistics or whole tensors /
# Gram matrices of hundreds of MB; the big ones are cut into calls of at most this size so that no call depends on
# RCCL staging one multi-GB message (first contact with an 8-GPU node should not be a 33 GB reduce).
MAX_COLLECTIVE_BYTES = 1 << 30


def _global_rank(group, group_rank: int) -> int:
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def _as_bytes(t: torch.Tensor) -> torch.Tensor:
    """Flat uint8 alias of a contiguous tensor (bool travels as bytes: RCCL has no bool type)."""
    t = t.reshape(-1)  # (0-dim tensors cannot be re-viewed with another element size)
    return t.view(torch.uint8) if t.dtype != torch.uint8 else t


def _chunks(nbytes: int, limit: int):
    limit = max(int(limit), 1)
    return [(a, min(a + limit, nbytes)) for a in range(0, nbytes, limit)]


def owner_rank(index: int, world: int | None = None) -> int:
    """Rank that `shard_list` deals unit `index` to."""
    if world is None:
        world = dist.get_world_size(replica_group()) if _initialized() else 1
    return index % world


def broadcast_from_owners(tensors, group=None, small_bytes: int = 32 << 20, max_bytes: int | None = None,
                          owners=None):
    """tensors[i] was computed on rank `owner_rank(i)` only (shard_list order; or on group rank `owners[i]`); afterwards
    every rank holds every result, in place.  All ranks pass same-shaped, same-dtype, contiguous tensors (results or
    empty buffers).

    Large tensors are broadcast in place (no staging copy), in calls of at most `max_bytes`; the small ones of an owner
    travel in one flat bucket.  This is the exchange step of the weight-side flows (masks, folded weights,
    fake-quantized weights) -- whether it PAYS is a bandwidth question: an elementwise pass over a
    replica's own weights runs at HBM speed, a broadcast at xGMI speed."""
    if not _initialized(group):
        return
    limit = MAX_COLLECTIVE_BYTES if max_bytes is None else max_bytes
    world = dist.get_world_size(group)
    me = dist.get_rank(group)
    for r in range(world):
        src = _global_rank(group, r)
        mine = [t for i, t in enumerate(tensors) if (i % world if owners is None else owners[i]) == r]
        small = []
        for t in mine:
            if not t.is_contiguous():
                raise ValueError("broadcast_from_owners: tensors must be contiguous")
            if t.numel() == 0:
                continue
            flat = _as_bytes(t)
            if flat.numel() >= small_bytes:
                for a, b in _chunks(flat.numel(), limit):
                    dist.broadcast(flat[a:b], src=src, group=group)
            else:
                small.append(flat)
        if small:
            bucket = torch.cat(small) if me == r else torch.empty(sum(f.numel() for f in small), dtype=torch.uint8,
                                                                  device=small[0].device)
            for a, b in _chunks(bucket.numel(), limit):
                dist.broadcast(bucket[a:b], src=src, group=group)
            if me != r:
                off = 0
                for f in small:
                    f.copy_(bucket[off:off + f.numel()])
                    off += f.numel()


def reduce_chunked(t: torch.Tensor, dst: int, op=None, group=None, max_bytes: int | None = None):
    """dist.reduce of a (large, contiguous) tensor to group rank `dst`, in calls of at most `max_bytes`."""
    if not _initialized(group):
        return
    op = dist.ReduceOp.SUM if op is None else op
    limit = MAX_COLLECTIVE_BYTES if max_bytes is None else max_bytes
    flat = t.reshape(-1)
    step = max(1, limit // t.element_size())
    for a in range(0, flat.numel(), step):
        dist.reduce(flat[a:a + step], dst=_global_rank(group, dst), op=op, group=group)
