"""The SparseGPT tie audit itself (tests/sgpt_audit.py) must have teeth: a mask that differs from the reference at a
group whose scores are clearly apart is reported, a flip between two (almost) equal scores is accepted, and whatever
follows a row's first disagreement is not looked at."""

import torch

import sgpt_audit


def _case():
    gen = torch.Generator().manual_seed(3)
    rows, cols = 32, 512
    w = torch.randn(rows, cols, generator=gen) * 0.05
    a = torch.randn(cols, 2 * cols, generator=gen)
    h = a @ a.t() / (2 * cols) + 0.1 * torch.eye(cols)
    hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(h)), upper=True).contiguous()
    return w, hinv


def test_trace_produces_a_2_of_4_mask_and_is_its_own_fixed_point():
    w, hinv = _case()
    ref = sgpt_audit.trace(w, hinv)
    assert (ref[0].view(32, -1, 4).sum(-1) == 2).all()
    report = sgpt_audit.audit(ref[0], ref, hinv)
    assert report["rows_differing"] == 0 and not report["unexplained"]


def test_a_real_divergence_is_reported_and_a_tie_is_not():
    w, hinv = _case()
    ref = sgpt_audit.trace(w, hinv)
    mask, scores = ref[0], ref[1]
    # a clear divergence: in row 5, second column block (trailing updates have happened), swap the kept element with the
    # LARGEST score against the pruned element with the SMALLEST one
    bad = mask.clone()
    g0 = 128 + 8
    s = scores[5, g0:g0 + 4]
    kept = [c for c in range(4) if mask[5, g0 + c]]
    pruned = [c for c in range(4) if not mask[5, g0 + c]]
    k = max(kept, key=lambda c: s[c])
    p = min(pruned, key=lambda c: s[c])
    bad[5, g0 + k], bad[5, g0 + p] = False, True
    bad[5, g0 + 64:] = ~bad[5, g0 + 64:]  # garbage AFTER the first disagreement must not matter
    report = sgpt_audit.audit(bad, ref, hinv)
    assert report["rows_differing"] == 1 and report["explained"] == 0 and len(report["unexplained"]) == 1
    assert report["unexplained"][0][:2] == (5, g0)
    # a tie: make two scores of a group equal in the trace and flip exactly them
    tied = (mask.clone(), scores.clone(), ref[2], ref[3])
    tied[1][7, g0 + kept[0]] = tied[1][7, g0 + pruned[0]] = 1.0
    kept7 = [c for c in range(4) if mask[7, g0 + c]]
    pruned7 = [c for c in range(4) if not mask[7, g0 + c]]
    tied[1][7, g0 + kept7[0]] = tied[1][7, g0 + pruned7[0]]
    flip = mask.clone()
    flip[7, g0 + kept7[0]], flip[7, g0 + pruned7[0]] = False, True
    report = sgpt_audit.audit(flip, tied, hinv)
    assert report["rows_differing"] == 1 and report["explained"] == 1 and not report["unexplained"]
    # in the FIRST column block no trailing update has touched the weights: there is no order noise to explain anything
    first = mask.clone()
    kept0 = [c for c in range(4) if mask[2, c]]
    pruned0 = [c for c in range(4) if not mask[2, c]]
    first[2, kept0[0]], first[2, pruned0[0]] = False, True
    report = sgpt_audit.audit(first, ref, hinv)
    assert len(report["unexplained"]) == 1
