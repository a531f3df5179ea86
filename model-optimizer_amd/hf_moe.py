"""Sparse MoE blocks whose experts are SEPARATE modules (a ModuleList of MLPs; Mixtral / Qwen-MoE under transformers 4.x,
remote-code models) -- the calibration rules of the reference's `_QuantSparseSequentialMoe`
(quantization/plugins/huggingface.py:623-750, utils/core_utils.py:710-747).  The experts' linears are ordinary quantized
linears; what the block adds is about WHICH tokens calibrate WHICH expert:

* after a max calibration every expert of a block shares one input amax per projection (the element-wise maximum over
  the experts: a token routed elsewhere at run time must not meet a smaller range), optionally one weight amax too
  (`sync_expert_weight_amax`), and an expert that saw no token still gets its weight amax;
* `moe_calib_experts_ratio`: while its experts calibrate, the block first runs with the router's top-k widened to that
  share of the experts (statistics only), then once more as configured for the output it returns; the tokens each
  expert received are counted in `expert_token_count`.

Batched expert containers (transformers >= 5: 3-D `gate_up_proj` / `down_proj`) are hf_experts.py's subject.

The reference re-classes the block (QuantModuleRegistry); here nothing is re-classed: the rules are functions over the
blocks `sparse_moe_blocks` finds, applied by model_calib.max_calibrate and, for the widened routing, by a `forward`
shadowed on the instance while a ratio is set.
"""

from __future__ import annotations

import warnings

import torch
from torch import nn

from .tensor_quantizer import TensorQuantizer


def _num_experts_of(obj):
    for attr in ("num_experts", "n_routed_experts"):  # n_routed_experts: NemotronH-style blocks
        if hasattr(obj, attr):
            return getattr(obj, attr)
    return None


def is_sparse_sequential_moe_block(module: nn.Module) -> bool:
    """huggingface.py:1592-1622: an `experts` child one can iterate over, and the routing attributes (`top_k` and the
    number of experts) on a `gate` child or, for older layouts, on the block itself."""
    experts = getattr(module, "experts", None)
    if experts is None or not hasattr(experts, "__iter__"):
        return False
    gate = getattr(module, "gate", None)
    if gate is not None and hasattr(gate, "top_k") and _num_experts_of(gate) is not None:
        return True
    if hasattr(module, "top_k"):
        if _num_experts_of(module) is None and hasattr(experts, "__len__"):
            module.num_experts = len(experts)
        return _num_experts_of(module) is not None
    return False


def sparse_moe_blocks(model: nn.Module):
    return [(n, m) for n, m in model.named_modules() if is_sparse_sequential_moe_block(m)]


def sync_moe_expert_amax(experts, sync_weight_amax: bool = False, calibrate_missing=None):
    """core_utils.sync_moe_expert_amax (:710-747).  `calibrate_missing(quantizer, weight)` fills the weight amax of an
    expert no token reached (the caller passes a weight-only max calibration)."""
    shared: dict = {}
    for expert in experts:
        for name, q in expert.named_modules():
            if not isinstance(q, TensorQuantizer) or q.amax is None:
                continue
            if "input_quantizer" in name or (sync_weight_amax and "weight_quantizer" in name):
                now = q.amax.detach().clone()
                shared[name] = now if name not in shared else torch.maximum(shared[name], now)
    for expert in experts:
        for name, q in expert.named_modules():
            if isinstance(q, TensorQuantizer) and name in shared:
                q.replace_amax(shared[name].detach().clone())
    for expert in experts:
        for name, q in expert.named_modules():
            if name.endswith("weight_quantizer") and isinstance(q, TensorQuantizer) and q.is_enabled and q.amax is None:
                weight = expert.state_dict().get(name.replace("weight_quantizer", "weight"))
                if weight is not None and calibrate_missing is not None:
                    calibrate_missing(q, weight)


def layer_sync_moe_local_experts_amax(model: nn.Module, sync_weight_amax: bool = False, calibrate_missing=None) -> int:
    """max_calibrate's step after finish_stats_collection (model_calib.py:365-368): every block's experts share their
    input amax.  A block with a calibration ratio keeps per-expert statistics (huggingface.py:738-750).  Returns the
    number of blocks synchronised."""
    n = 0
    for _, block in sparse_moe_blocks(model):
        if getattr(block, "_moe_calib_experts_ratio", None) is not None:
            continue
        sync_moe_expert_amax(block.experts, sync_weight_amax=sync_weight_amax, calibrate_missing=calibrate_missing)
        n += 1
    return n


# ------------------------------------------------------------------------------------------------ widened routing
def _top_k_owner(block):
    gate = getattr(block, "gate", None)
    return gate if gate is not None and hasattr(gate, "top_k") else block


class _CountTokens:
    """Forward hook of the block's router: tokens per expert while the counter is on (huggingface.py:668-682)."""

    def __init__(self, block):
        self.block = block

    def __deepcopy__(self, memo):
        return _CountTokens(memo.get(id(self.block), self.block))

    def __call__(self, gate, args, output):
        block = self.block
        if not getattr(block, "_count_expert_tokens", False) or not hasattr(block, "expert_token_count"):
            return
        with torch.no_grad():
            if isinstance(output, tuple) and len(output) >= 3:
                indices = output[2]  # (logits, scores, indices) routers
            else:
                logits = output if not isinstance(output, tuple) else output[0]
                _, indices = torch.topk(logits.float(), _top_k_owner(block).top_k, dim=-1)
            counts = torch.bincount(indices.reshape(-1), minlength=block.expert_token_count.shape[0])
            block.expert_token_count += counts.to(block.expert_token_count.device)


class _WidenedForward:
    """The block's `forward` while a calibration ratio is set: an object rather than a closure, so that a deep copy of
    the model gets a forward bound to the COPIED block (a copied closure would keep running the original's experts), and
    the block's own forward is looked up on its class at call time."""

    def __init__(self, block):
        self.block = block

    def __deepcopy__(self, memo):
        # (copy._reconstruct enters the new block into the memo before it copies the block's attributes)
        return _WidenedForward(memo.get(id(self.block), self.block))

    def __call__(self, hidden_states, *args, **kwargs):
        block = self.block
        own = type(block).forward.__get__(block)
        ratio = getattr(block, "_moe_calib_experts_ratio", None)
        if ratio is None:
            return own(hidden_states, *args, **kwargs)
        if any(getattr(m, "_if_calib", False) for m in block.experts.modules()):
            block._count_expert_tokens = ratio < 1.0  # all experts calibrated anyway at 1.0
            if block._count_expert_tokens and not hasattr(block, "expert_token_count"):
                n = next((v for v in (_num_experts_of(getattr(block, "gate", None)), _num_experts_of(block),
                                      _num_experts_of(block.experts)) if v), 0)
                if not n:
                    warnings.warn(f"{type(block).__name__}: could not resolve num_experts; expert routing will not be "
                                  "tracked for this layer.")
                else:
                    block.register_buffer("expert_token_count", torch.zeros(n, dtype=torch.long, device=next(block.parameters()).device),
                                          persistent=False)
                    if hasattr(block, "gate"):
                        block.gate.register_forward_hook(_CountTokens(block))
                # (huggingface.py:664: setting the counter up also switches it off for the call that did it -- the
                # block's first calibration batch is not in the table; kept, the table is the reference's)
                block._count_expert_tokens = False
            owner = _top_k_owner(block)
            n_experts = _num_experts_of(owner) or _num_experts_of(block) or _num_experts_of(block.experts) or len(block.experts)
            configured = owner.top_k
            owner.top_k = max(configured, round(n_experts * ratio))
            try:
                own(hidden_states, *args, **kwargs)  # statistics from the widened routing; the output is dropped
            finally:
                owner.top_k = configured
            block._count_expert_tokens = False
        out = own(hidden_states, *args, **kwargs)
        block._count_expert_tokens = False
        return out


def set_moe_calib_experts_ratio(model: nn.Module, ratio) -> int:
    """mode.py:239-247: the share of a block's experts every calibration token is sent to (None: routing as configured;
    the shadowed forward is taken off again).  Returns the number of blocks it applies to."""
    if ratio is not None:
        assert isinstance(ratio, (int, float)) and 0 < ratio <= 1, f"Invalid moe_calib_experts_ratio {ratio!r}"
    blocks = sparse_moe_blocks(model)
    for _, block in blocks:
        shadow = block.__dict__.get("forward")
        if ratio is None:
            if isinstance(shadow, _WidenedForward):
                del block.forward
            block._moe_calib_experts_ratio = None
            continue
        if shadow is not None and not isinstance(shadow, _WidenedForward):
            raise RuntimeError(f"{type(block).__name__}: the block's forward is already replaced on the instance; "
                               "moe_calib_experts_ratio wraps the class's forward")
        block._moe_calib_experts_ratio = ratio
        if shadow is None:
            block.forward = _WidenedForward(block)
    return len(blocks)
