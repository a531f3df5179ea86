"""Tie audit of SparseGPT masks (test infrastructure).

create_sgpt_mask's column sweep is sequential fp32 arithmetic (bit-exact against the oracle), but between two column
blocks the remaining weights are updated by an fp32 GEMM (`w_rows[:, i2:] -= delta_blk.matmul(hessian_inv[i1:i2, i2:])`,
sparsity/weight_sparsity/sparsegpt.py:124) whose summation order is the BLAS library's -- different between the
reference's CPU run and any GPU.  A reordered 128-term fp32 sum moves a weight by a few ulp of the terms' magnitude; where
two pruning scores of a 4-group are closer than that, the discrete choice can flip, and from there on the row's later
columns see a different error feedback (the OBS update is not contractive), so EVERYTHING after a row's first flip may
differ legitimately.  An index result that is not bit-equal therefore needs a proof that every row's FIRST disagreement
is such a tie -- anything else would be a real divergence hiding in a tolerance.

`trace()` re-runs the reference algorithm (sparsegpt.py:72-133, restated line by line on torch CPU tensors) and keeps,
for every row and 4-group, the pruning scores and weights at decision time and, for every element, the absolute sum of
all trailing-update terms it has received.  `audit()` takes another mask and checks each row's first disagreeing group:
the elements the two masks treat differently must have reference scores within the worst-case reordering bound of
those sums,  |fl(sum) - sum| <= (n - 1) eps sum|terms|  per 128-term update and side, times SAFETY for the
propagation through the in-block rank-1 updates."""

import torch

EPS = 2.0 ** -24  # unit round-off of fp32
SAFETY = 4.0


@torch.no_grad()
def trace(weight: torch.Tensor, hessian_inv: torch.Tensor, col_bs: int = 128, n: int = 2, m: int = 4):
    """The reference loop on CPU fp32.  Returns (mask [rows, cols] bool, scores [rows, cols] fp32: the score of every
    element when its group was decided, w_at [rows, cols]: its value then, abs_terms [rows, cols]: sum of |update terms|
    received from trailing updates before that)."""
    w_rows = weight.detach().float().cpu().clone()
    hinv = hessian_inv.detach().float().cpu()
    rows, cols = w_rows.shape
    diag = torch.diagonal(hinv)
    scores = torch.zeros(rows, cols)
    w_at = torch.zeros(rows, cols)
    abs_terms = torch.zeros(rows, cols)
    for i1 in range(0, cols, col_bs):
        i2 = min(i1 + col_bs, cols)
        w_blk = w_rows[:, i1:i2].clone()
        q_blk = torch.zeros_like(w_blk)
        delta_blk = torch.zeros_like(w_blk)
        hinv_blk = hinv[i1:i2, i1:i2]
        d_blk = diag[i1:i2]
        mask_blk = torch.zeros_like(w_blk, dtype=torch.bool)
        for j in range(i2 - i1):
            w = w_blk[:, j]
            d = d_blk[j]
            if j % m == 0:
                err = (w_blk[:, j:j + m] ** 2) / (d_blk[j:j + m] ** 2 + 1e-9)
                mask_blk.scatter_(1, j + torch.topk(err, n, dim=1, largest=False)[1], True)
                scores[:, i1 + j:i1 + j + m] = err
                w_at[:, i1 + j:i1 + j + m] = w_blk[:, j:j + m]
            q = w.clone()
            q[mask_blk[:, j]] = 0
            q_blk[:, j] = q
            e = (w - q) / d
            w_blk[:, j:] -= e.unsqueeze(1).matmul(hinv_blk[j, j:].unsqueeze(0))
            delta_blk[:, j] = e
        w_rows[:, i1:i2] = q_blk
        if i2 < cols:
            w_rows[:, i2:] -= delta_blk.matmul(hinv[i1:i2, i2:])
            abs_terms[:, i2:] += delta_blk.abs().matmul(hinv[i1:i2, i2:].abs())
    return w_rows != 0, scores, w_at, abs_terms


def audit(mask_other: torch.Tensor, ref, hessian_inv: torch.Tensor, col_bs: int = 128, m: int = 4):
    """Every row's first 4-group where `mask_other` differs from the traced reference mask must be a tie within the
    reordering bound.  Returns {"rows_differing", "explained", "unexplained": [(row, col, gap, bound), ...]}."""
    mask_ref, scores, w_at, abs_terms = ref
    mask_other = mask_other.cpu().bool()
    diag = torch.diagonal(hessian_inv.detach().float().cpu())
    rows, cols = mask_ref.shape
    diff = (mask_ref != mask_other).view(rows, cols // m, m).any(-1)
    out = {"rows_differing": 0, "explained": 0, "unexplained": [], "worst_gap_over_bound": 0.0}
    for r in diff.any(1).nonzero().flatten().tolist():
        out["rows_differing"] += 1
        g = int(diff[r].nonzero()[0])
        c0 = g * m
        # elements kept by one mask and pruned by the other (pruned = mask False)
        a = [c for c in range(c0, c0 + m) if mask_ref[r, c] and not mask_other[r, c]]  # pruned only by the other
        b = [c for c in range(c0, c0 + m) if mask_other[r, c] and not mask_ref[r, c]]  # pruned only by the reference
        n_terms = col_bs - 1
        dw = SAFETY * 2.0 * n_terms * EPS * abs_terms[r, c0:c0 + m]  # how far a reordering can move each weight
        ds = 2.0 * w_at[r, c0:c0 + m].abs() * dw / (diag[c0:c0 + m] ** 2 + 1e-9) + dw ** 2 / (diag[c0:c0 + m] ** 2 + 1e-9)
        ok = bool(a) and bool(b)
        worst = 0.0
        for ca in a:
            for cb in b:
                gap = abs(float(scores[r, ca] - scores[r, cb]))
                bound = float(ds[ca - c0] + ds[cb - c0])
                worst = max(worst, gap / bound if bound > 0 else float("inf"))
                if gap > bound:
                    ok = False
                    out["unexplained"].append((r, c0, gap, bound))
        out["worst_gap_over_bound"] = max(out["worst_gap_over_bound"], worst)
        out["explained"] += int(ok)
    return out
