"""2:4 magnitude sparsity -- mirror of modelopt.torch.sparsity.weight_sparsity.magnitude on our mask kernel."""

from __future__ import annotations

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
