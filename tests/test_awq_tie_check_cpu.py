"""The self-check of the AWQ re-scoring margin (model_calib.tie_margin_check), replayed on MEASURED score tables.

tests/golden/awq_hf_tables.json holds, for the 224 linears of a random-init Hugging Face Llama-3-8B (bf16, 64 x 4096
calibration tokens, one MI355X, round 3), the Gram score and the error-GEMM ("exact": the reference's arithmetic,
model_calib.py:1489-1495, :1548-1556) score of ALL 11 candidates.  It is the adversarial case for the Gram screen: all
candidates of a linear lie within 0.1-0.9 % of each other, the two engines disagree by up to 8e-4 between candidates of
one linear and the plain Gram minimum is not the exact minimum on 30 linears.  The policy under test decides which
candidates to re-score from the Gram scores alone, sees the exact scores only of those, and must end on the exact
minimum (first minimum in ascending alpha, :1637)."""

import json
import os

import pytest

import _moa_import

moa = _moa_import.load()
from model_optimizer_amd import model_calib  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def tables():
    with open(os.path.join(HERE, "golden", "awq_hf_tables.json")) as f:
        return [(lin["gram"], lin["exact"]) for lin in json.load(f)["linears"]]


def _first_min(values):
    return min(range(len(values)), key=values.__getitem__)


def _search(gram, exact, margin, check):
    """awq_lite's selection for one linear: contenders within `margin` of the best Gram score are re-scored; with
    `check` the margin verifies itself and widens (every widening is one more pass over the calibration data)."""
    best = min(gram)
    pending = [i for i, v in enumerate(gram) if v <= best * (1.0 + margin)]
    if len(pending) < 2:
        return _first_min(gram), 0, 0
    scored, rounds, passes = {}, 0, 0
    need = None
    while pending:
        passes += 1
        scored.update({i: exact[i] for i in pending})
        if not check:
            break
        need, margin, pending = model_calib.tie_margin_check(gram, scored, margin, rounds)
        if pending:
            rounds += 1
    idx = sorted(scored)
    _search.last = (need, margin)  # (the requirement and the margin the linear was settled at)
    return idx[_first_min([scored[i] for i in idx])], len(scored), passes


def test_plain_gram_minimum_is_not_enough_on_this_model(tables):
    assert sum(_first_min(g) != _first_min(e) for g, e in tables) >= 20


def test_default_margin_with_self_check_finds_every_exact_minimum(tables):
    margin = model_calib.GRAM_TIE_MARGIN[moa.ops.torch.bfloat16] + model_calib.GRAM_PLANES_SLACK
    wrong = cands = extra = 0
    for g, e in tables:
        w, n, passes = _search(g, e, margin, check=True)
        wrong += w != _first_min(e)
        cands += n
        extra += passes > 1
    assert wrong == 0
    assert cands < 0.6 * 11 * len(tables)  # about half of the full search, even on this all-ties model
    assert extra <= 40  # widening (a handful of candidates re-scored from stored activations) is not the common case: 31 of 224


def test_every_linear_is_settled_with_room_to_spare(tables):
    """rounds 4 and 5 settled this model's tightest linear with requirement / margin 0.984.  With
    TIE_HEADROOM a linear that passes by less than 30 % is widened like one that fails: every linear of the measured
    full-size tables ends at requirement / margin <= 0.7 (or with all of its candidates scored), on the exact minimum, and
    what that costs is counted here."""
    margin = model_calib.GRAM_TIE_MARGIN[moa.ops.torch.bfloat16] + model_calib.GRAM_PLANES_SLACK
    worst, widened, cands, tight = 0.0, 0, 0, 0
    for g, e in tables:
        w, n, passes = _search(g, e, margin, check=True)
        assert w == _first_min(e)
        cands += n
        widened += passes > 1
        need, settled_at = _search.last
        if need is not None and n < len(g) and settled_at != float("inf"):
            worst = max(worst, need / settled_at)
            tight += need > model_calib.TIE_HEADROOM * margin  # would have passed the bare rule (need <= margin) or failed it
    assert worst <= model_calib.TIE_HEADROOM + 1e-12, worst
    # measured on these tables: bare rule (headroom 1.0) 5 linears widened, 1199 of 2464 candidates re-scored, worst ratio 0.973;
    # headroom 0.7: 31 widened, 1250 re-scored (+4 %), worst ratio 0.691 -- the same 224 exact minima either way
    assert widened <= 40 and cands < 0.6 * 11 * len(tables), (widened, cands)


def test_a_requirement_above_the_margin_is_widened_until_the_exact_minimum_is_in():
    """A linear built to break a fixed margin: the Gram screen ranks candidate 4 best by 3e-3 while the exact engine prefers
    candidate 7, which lies OUTSIDE the 1.3e-3 margin; the two engines disagree by 2e-3 among the candidates that are inside.
    need > margin on the first check, the margin widens, candidate 7 is admitted and wins -- the exact engine's argmin."""
    gram = [1.05, 1.04, 1.03, 1.0008, 1.0, 1.0005, 1.001, 1.003, 1.02, 1.04, 1.06]
    exact = [1.05, 1.04, 1.03, 1.0030, 1.0012, 1.0002, 1.0004, 1.0001, 1.02, 1.04, 1.06]
    assert _first_min(gram) == 4 and _first_min(exact) == 7
    assert _search(gram, exact, 1.3e-3, check=False)[0] != 7  # a fixed margin never sees candidate 7
    w, n, passes = _search(gram, exact, 1.3e-3, check=True)
    assert w == 7 and passes >= 2 and n < len(gram)
    need, settled_at = _search.last
    assert need <= model_calib.TIE_HEADROOM * settled_at


def test_a_stricter_factor_widens_some_linears_of_this_model_and_changes_nothing(tables, monkeypatch):
    monkeypatch.setattr(model_calib, "TIE_SPREAD_FACTOR", 2.0)
    margin = model_calib.GRAM_TIE_MARGIN[moa.ops.torch.bfloat16] + model_calib.GRAM_PLANES_SLACK
    results = [_search(g, e, margin, check=True) for g, e in tables]
    assert all(w == _first_min(e) for (w, _, _), (_, e) in zip(results, tables))
    assert 0 < sum(p > 1 for _, _, p in results) <= 40  # the check has teeth on this data


def test_too_small_a_fixed_margin_flips_linears_and_the_self_check_repairs_them(tables):
    fixed = sum(_search(g, e, 1e-4, check=False)[0] != _first_min(e) for g, e in tables)
    assert fixed >= 1, "the fixture lost its teeth"
    checked = sum(_search(g, e, 1e-4, check=True)[0] != _first_min(e) for g, e in tables)
    # the check repairs what it can measure; a margin an order of magnitude too small is not fully recoverable (a linear
    # with one candidate inside it re-scores nothing), which is why the default margin stays at 1e-3 and the check is
    # the safeguard on top of it, not a replacement for it
    assert checked < fixed


def test_tie_margin_check_rounds_and_nan():
    gram = [1.0, 1.0005, 1.002, 1.01, 1.2]
    # engines agree -> settled
    need, margin, new = model_calib.tie_margin_check(gram, {0: 1.0, 1: 1.0005}, 1e-3, 0)
    assert new == [] and margin == 1e-3 and need == pytest.approx(0.0, abs=1e-12)
    # they disagree by 1e-3 between the two scored candidates -> need = factor * 1e-3 (+ the winner's gap), margin doubles that
    need, margin, new = model_calib.tie_margin_check(gram, {0: 1.001, 1: 1.0005}, 1e-3, 0)
    assert need == pytest.approx(model_calib.TIE_SPREAD_FACTOR * 1e-3 + 5e-4) and margin == pytest.approx(2 * need)
    assert new == [2]
    # inside the rule but without room (need = 0.9 x margin): widened as well
    need, margin, new = model_calib.tie_margin_check(gram, {0: 1.0002, 1: 1.0005}, 4.4e-4, 0)
    assert need == pytest.approx(4e-4) and need > model_calib.TIE_HEADROOM * 4.4e-4 and margin == pytest.approx(8.8e-4) and new == []
    # last round: everything that is left
    _, margin, new = model_calib.tie_margin_check(gram, {0: 1.001, 1: 1.0005}, 1e-3, model_calib.TIE_CHECK_MAX_ROUNDS - 1)
    assert margin == float("inf") and new == [2, 3, 4]
    # a NaN exact score: score every candidate
    _, _, new = model_calib.tie_margin_check(gram, {0: float("nan"), 1: 1.0}, 1e-3, 0)
    assert new == [2, 3, 4]
    # all scored: nothing to add
    assert model_calib.tie_margin_check(gram, dict(enumerate(gram)), 1e-3, 0)[2] == []
