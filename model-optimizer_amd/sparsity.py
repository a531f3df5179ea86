"""2:4 magnitude sparsity -- mirror of modelopt.torch.sparsity.weight_sparsity.magnitude on our mask kernel."""

from __future__ import annotations

import fnmatch
import re
import warnings

import torch

from . import ops

_PATTERN_2_4 = "2:4 sparsity"


def get_nmprune_info(pattern: str):
    m = re.search(r"(\d+):(\d+) sparsity", pattern)
    return (True, *map(int, m.groups())) if m else (False, 0, 0)


def check_weight_size(weight: torch.Tensor, mod_name: str = "") -> bool:
    """MagnitudeSearcher._check_weight_size (magnitude.py:134-144): Cout % 8 == 0 and Cin % 16 == 0."""
    if weight.size(0) % 8 != 0 or weight.size(1) % 16 != 0:
        warnings.warn(f"Skipping sparsifying {mod_name} of size={weight.size()!s} and type={weight.dtype!s} for sparsity")
        return False
    return True


@torch.no_grad()
def create_asp_mask(tensor: torch.Tensor, pattern: str = _PATTERN_2_4) -> torch.Tensor:
    """magnitude.py:91-128: bool mask of tensor's shape; groups of 4 run along dim 1 (conv weights are
    permuted so that they do).  The reference converts to fp32 first; |w| ordering is unchanged by the
    widening, so the kernel reads the storage dtype directly."""
    if pattern != _PATTERN_2_4:
        raise NotImplementedError(f"Unsupported pattern {pattern} for ASP sparsity")
    shape = tensor.shape
    t = tensor.detach()
    if len(shape) == 1:
        t2, back = t.view(1, shape[0]), None
    elif len(shape) == 2:
        t2, back = t, None
    elif len(shape) == 3:
        t2 = t.permute(0, 2, 1).contiguous().view(shape[0] * shape[2], shape[1])
        back = lambda m: m.view(shape[0], shape[2], shape[1]).permute(0, 2, 1).contiguous()  # noqa: E731
    elif len(shape) == 4:
        t2 = t.permute(2, 3, 0, 1).contiguous().view(shape[2] * shape[3] * shape[0], shape[1])
        back = lambda m: m.view(shape[2], shape[3], shape[0], shape[1]).permute(2, 3, 0, 1).contiguous()  # noqa: E731
    else:
        raise NotImplementedError(f"{len(shape)}-d tensors are not supported")
    t2 = t2.contiguous()
    cols = t2.shape[1]
    if cols % 4:  # reshape_1d zero-pads the columns (magnitude.py:43-52)
        padded = t2.new_zeros(t2.shape[0], cols + (4 - cols % 4))
        padded[:, :cols] = t2
        mask = ops.mask_2to4(padded)[:, :cols].contiguous()
    else:
        mask = ops.mask_2to4(t2)
    if back is not None:
        mask = back(mask)
    return mask.view(shape).to(dtype=torch.bool)


# ------------------------------------------------------------------------------------------------ SparseGPT
class HessianState:
    """Running Hessian of one linear's inputs -- mod.hessian / mod.samples of SparseGPTSearcher
    (sparsegpt.py:206-236): fp32 [Cin, Cin] on the weight's device."""

    def __init__(self, cols: int, device):
        self._h = torch.zeros(cols, cols, dtype=torch.float32, device=device)
        self._upper = False  # lower triangle stale (MFMA path updates the upper tiles only)
        self.samples = 0

    @property
    def hessian(self) -> torch.Tensor:
        if self._upper:
            ops.symmetrize(self._h)
            self._upper = False
        return self._h

    @torch.no_grad()
    def update(self, inp: torch.Tensor):
        """_hook_compute_hessian (sparsegpt.py:238-276) for a linear: H <- H * s/(s+b) + (2/(s+b)) X^T X with b the
        batch dimension of the input.  16-bit inputs: transpose + MFMA contraction (ops.hessian_accum); fp32 inputs:
        the library's fp32 GEMM, like the reference."""
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        b = inp.shape[0]
        x2 = inp.reshape(-1, inp.shape[-1])
        decay = self.samples / (self.samples + b)
        self.samples += b
        scale = 2.0 / self.samples
        if x2.dtype in (torch.bfloat16, torch.float16) and x2.shape[1] % 4 == 0:
            ops.hessian_accum(self._h, x2, decay, scale, upper_only=True)
            self._upper = True
        else:
            xf = x2.float()
            self.hessian.mul_(decay).addmm_(xf.t(), xf, alpha=scale)


def invert(hessian: torch.Tensor) -> torch.Tensor:
    """sparsegpt.py:30-42: upper Cholesky factor of H^-1."""
    try:
        h = torch.linalg.cholesky(hessian)
        h = torch.cholesky_inverse(h)
        h = torch.linalg.cholesky(h, upper=True)
    except RuntimeError:
        eps = 1e-6 * torch.eye(hessian.size(0), device=hessian.device)
        h = torch.cholesky_inverse(torch.linalg.cholesky(hessian + eps))
    return h


def prepare_hessian(hessian: torch.Tensor, hessian_damp: float):
    """The weight-independent half of sparsegpt.py:45-69: dead columns (zero diagonal), damping, inverse factor.
    Returns (dead-column mask, hinv).  Linears that read the same tensor have the same Hessian and share this."""
    hessian = hessian.clone()
    zero = torch.diag(hessian) == 0
    hessian[zero, zero] = 1
    damp = hessian_damp * torch.mean(torch.diag(hessian))
    diag = torch.arange(hessian.size(0), device=hessian.device)
    hessian[diag, diag] += damp
    return zero, invert(hessian).contiguous()


def prepare(tensor: torch.Tensor, hessian: torch.Tensor, hessian_damp: float):
    """sparsegpt.py:45-69: dead columns, damping, inverse factor.  Returns (fp32 working weight, hinv)."""
    weight = tensor.detach().clone()
    if weight.dim() == 4:
        weight = weight.flatten(1)
    zero, hinv = prepare_hessian(hessian.to(weight.device), hessian_damp)
    weight[:, zero] = 0
    return weight, hinv


@torch.no_grad()
def create_sgpt_mask(tensor: torch.Tensor, hessian: torch.Tensor, config: dict, hessian_inv: torch.Tensor | None = None,
                     dead_columns: torch.Tensor | None = None):
    """sparsegpt.py:72-133.  The per-column Python loop of the reference is one kernel per column block
    (ops.sgpt_block_sweep); the trailing-block update (an fp32 GEMM in the BLAS library's order there, :124) is
    ops.sgpt_trailing_update: the fp32 fma chain over the block's columns in ascending order on the fp32 matrix cores,
    so the mask is reproducible bit for bit from a given inverse factor.
    `hessian_inv` (+ `dead_columns`) short-cuts prepare() with an already prepared factor: tests, and linears that
    share a Hessian."""
    shape = tensor.size()
    is_nm, n, m = get_nmprune_info(config.get("pattern", _PATTERN_2_4))
    if not is_nm:
        raise NotImplementedError("SparseGPT: only n:m patterns produce a mask (sparsegpt.py:111-115)")
    if hessian_inv is None:
        weight, hessian_inv = prepare(tensor, hessian, config.get("hessian_damp", 0.1))
    else:
        weight = tensor.detach().clone().flatten(1) if tensor.dim() == 4 else tensor.detach().clone()
        if dead_columns is not None:
            weight[:, dead_columns] = 0
    hessian_inv = hessian_inv.float().contiguous()
    rows, cols = weight.size()
    col_bs = config.get("col_block_size", 128)
    row_bs = config.get("row_block_size", -1)
    if row_bs == -1:
        row_bs = rows
    for r1 in range(0, rows, row_bs):
        r2 = min(r1 + row_bs, rows)
        w_rows = weight[r1:r2].float().contiguous()
        for i1 in range(0, cols, col_bs):
            i2 = min(i1 + col_bs, cols)
            delta = ops.sgpt_block_sweep(w_rows, i1, i2 - i1, hessian_inv, n, m)
            if i2 < cols:
                ops.sgpt_trailing_update(w_rows, i1, delta, hessian_inv)
        weight[r1:r2] = w_rows.to(weight.dtype)
    return (weight != 0).view(shape)


def _combine_hessians(mods, states, owner_of) -> dict:
    """Data-parallel SparseGPT: every rank accumulated H_r = (2 / S_r) sum_{its batches} X^T X over S_r samples.  The
    Hessian over ALL batches is sum_r (S_r / S) H_r: each rank scales its matrices by S_r, the distinct matrices are
    dealt over the ranks (largest first onto the least loaded one) and SUM-reduced to their owner in calls of at most
    1 GiB, the owner divides by S.  Returns {Hessian-owning module: group rank, "me": this rank}; every rank must see
    the same input-sharing structure (same model, same forward)."""
    import torch.distributed as dist

    from . import distributed as mdist

    group = mdist.replica_group()
    owners = [m for m in mods if m not in owner_of]
    sig = [(mods.index(m), tuple(states[m]._h.shape)) for m in owners]
    gathered = [None] * dist.get_world_size(group)
    dist.all_gather_object(gathered, sig, group=group)
    if any(g != sig for g in gathered):
        raise RuntimeError("sparsegpt (data parallel): the ranks do not agree on which linears share their input")
    dev = states[owners[0]]._h.device
    samples = torch.tensor([float(states[m].samples) for m in owners], dtype=torch.float64, device=dev)
    total = samples.clone()
    dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
    load = [0] * dist.get_world_size(group)
    placed = {"me": dist.get_rank(group)}
    cost = {m: sum(x.weight.numel() for x in mods if owner_of.get(x, x) is m) * states[m]._h.shape[0] for m in owners}
    for m in sorted(owners, key=lambda m: -cost[m]):
        r = min(range(len(load)), key=load.__getitem__)
        placed[m] = r
        load[r] += cost[m]
    for m, s_r, s in zip(owners, samples.tolist(), total.tolist()):  # same order on every rank
        h = states[m].hessian
        h.mul_(s_r)
        mdist.reduce_chunked(h, placed[m], dist.ReduceOp.SUM, group)
        if placed[m] == placed["me"]:
            h.div_(max(s, 1.0))
            states[m].samples = int(s)
    return placed


def check_weight_size_sgpt(weight: torch.Tensor, pattern: str = _PATTERN_2_4, mod_name: str = "") -> bool:
    """SparseGPTSearcher._check_weight_size (sparsegpt.py:152-165)."""
    _, _, m = get_nmprune_info(pattern)
    if weight.size(0) % m != 0 or weight.size(1) % m != 0:
        warnings.warn(f"Skipping pruning {mod_name} of size={weight.size()!s} and type={weight.dtype!s} for SparseGPT")
        return False
    return True


def _magnitude_masks(targets, pattern: str, shard: bool) -> dict:
    """2:4 magnitude masks of all target linears.  2-D contiguous GPU weights of one dtype are masked by ONE
    multi-tensor launch (SegmentTable.mask_2to4); anything else goes through create_asp_mask.  shard: this rank computes
    the masks of its share of the list (distributed.shard_list) and receives the others from their owners."""
    from . import distributed as mdist
    from .multi_tensor import SegmentTable

    masks = {m: torch.empty(m.weight.shape, dtype=torch.bool, device=m.weight.device) for _, m in targets}
    mine = mdist.shard_list(targets) if shard else targets
    batched: dict = {}
    applied = set()
    for _, m in mine:
        w = m.weight.detach()
        if w.is_cuda and w.dim() == 2 and w.is_contiguous() and w.shape[1] % 4 == 0 and pattern == _PATTERN_2_4:
            batched.setdefault((w.dtype, w.device), []).append(m)
        else:
            masks[m].copy_(create_asp_mask(m.weight, pattern))
    for mods in batched.values():
        # mask AND `weight.mul_(mask)` in one pass over the weights (5 bytes per element; the mask pass followed by a
        # mask-typed multiply per tensor moved 3 + 9)
        SegmentTable([m.weight.detach() for m in mods], outputs=[masks[m] for m in mods]).mask_2to4_apply()
        applied.update(mods)
    if shard:
        mdist.broadcast_from_owners([masks[m] for _, m in targets], group=mdist.replica_group())
    return masks, applied


@torch.no_grad()
def sparsify(model: torch.nn.Module, mode: str = "sparse_magnitude", forward_loop=None, config: dict | None = None,
             shard_weights: bool | None = None):
    """mts.sparsify for the two modes of the path (sparsification.py:32-97): every eligible nn.Linear gets a bool
    `_weight_mask` buffer and its weight is masked in place (the reference's SparseModule multiplies on access).
    "sparsegpt": forward hooks accumulate the input Hessians during forward_loop, then create_sgpt_mask.

    shard_weights (data-parallel replicas; None follows distributed.declare_data_parallel): the mask computation is
    dealt over the ranks and the masks are broadcast from their owners.  SparseGPT: every rank's forward_loop feeds its
    own share of the calibration batches; the distinct Hessians are dealt over the ranks, combined on their owner
    (sample-weighted SUM, the running mean of sparsegpt.py:238-276 over ALL batches), and only the owner inverts the
    Hessian and sweeps the linears that read it."""
    from . import distributed as mdist

    cfg = {"pattern": _PATTERN_2_4, "col_block_size": 128, "row_block_size": -1, "hessian_damp": 0.1, **(config or {})}
    shard = mdist.resolve_shard(shard_weights)
    applied: set = set()
    # weight_sparsity/config.py:27-45: {"nn.Linear": {"*": {}, "*lm_head*": None}} -- the output head stays dense
    linears = [(n, m) for n, m in model.named_modules() if isinstance(m, torch.nn.Linear) and not fnmatch.fnmatch(n, "*lm_head*")]
    if mode == "sparse_magnitude":
        targets = [(n, m) for n, m in linears if check_weight_size(m.weight, n)]
        masks, applied = _magnitude_masks(targets, cfg["pattern"], shard)
    elif mode == "sparsegpt":
        assert forward_loop is not None, "Please provide `data_loader` or `forward_loop`!"
        targets = [(n, m) for n, m in linears if check_weight_size_sgpt(m.weight, cfg["pattern"], n)]
        states = {m: HessianState(m.weight.size(1), m.weight.device) for _, m in targets}
        # Linears fed by the SAME tensor object (q / k / v, gate / up) have the same Hessian: the first one accumulates
        # it, the others point at it -- one X^T X, one Cholesky inverse per distinct input instead of one per linear.
        owner_of: dict = {}
        last = {"input": None, "owner": None}

        def hook(mod, inp, out):
            x = inp[0] if isinstance(inp, tuple) else inp
            owner = last["owner"] if last["input"] is x else None
            if owner is not None and states[owner]._h.shape == states[mod]._h.shape and owner_of.get(mod, owner) is owner:
                owner_of[mod] = owner
                return
            if mod in owner_of:
                raise RuntimeError("sparsegpt: a linear that shared its input with another one in an earlier batch got "
                                   "a different tensor now")
            states[mod].update(x)
            last["input"], last["owner"] = x, mod

        handles = [m.register_forward_hook(hook) for _, m in targets]
        try:
            forward_loop(model)
        finally:
            for h in handles:
                h.remove()
            last["input"] = None
        for mod in owner_of:
            states[mod]._h = None  # never written: release
        prepared: dict = {}
        users: dict = {}
        for _, m in targets:
            own = owner_of.get(m, m)
            users[own] = users.get(own, 0) + 1
        masks = {}
        placed = None
        if shard:
            placed = _combine_hessians([m for _, m in targets], states, owner_of)
        for _, m in targets:
            own = owner_of.get(m, m)
            if placed is not None and placed[own] != placed["me"]:
                masks[m] = torch.empty(m.weight.shape, dtype=torch.bool, device=m.weight.device)
                states[own]._h = None
                continue
            if own not in prepared:
                prepared[own] = prepare_hessian(states[own].hessian, cfg["hessian_damp"])
                states[own]._h = None  # Cin^2 floats: only the inverse factor is needed from here on
            zero, hinv = prepared[own]
            masks[m] = create_sgpt_mask(m.weight, None, cfg, hessian_inv=hinv, dead_columns=zero)
            users[own] -= 1
            if users[own] == 0:
                del prepared[own]
        if placed is not None:
            # (received IN PLACE: the list must hold the mask tensors themselves; they are contiguous by construction)
            mdist.broadcast_from_owners([masks[m] for _, m in targets], group=mdist.replica_group(),
                                        owners=[placed[owner_of.get(m, m)] for _, m in targets])
    else:
        raise ValueError(f"sparsity mode {mode!r} is outside this path")
    for _, m in targets:
        mask = masks[m]
        m.register_buffer("_weight_mask", mask)
        if m not in applied:  # (the fused mask + apply pass has masked these already)
            m.weight.data.mul_(mask.to(m.weight.dtype))
    return model


@torch.no_grad()
def export(model: torch.nn.Module) -> torch.nn.Module:
    """mts.export (sparsification.py:100-123): the sparse model as a regular one -- the masks are folded into the weights and
    no longer kept (nor enforced on later weight updates).  sparsify() above stores the masked weights already, so what is
    left to do is one more application (a weight written since keeps the pattern) and dropping the `_weight_mask` buffers."""
    for m in model.modules():
        mask = m._buffers.get("_weight_mask")
        if mask is not None:
            m.weight.data.mul_(mask.to(m.weight.dtype))
            del m._buffers["_weight_mask"]
    return model
