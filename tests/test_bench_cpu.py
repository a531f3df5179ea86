"""bench.py's host logic without a GPU: how the pool of per-layer weight tensors is partitioned over the ranks in the
two scaling modes, the model shape tables behind the metric's byte counts, and the committed-PMC lookup."""

import importlib.util
import os

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("moq_bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def test_model_tables_give_the_survey_sizes():
    # Llama-3-8B 6.98 G linear parameters in 224 tensors, Llama-3-70B 68.45 G in 560
    for model, n_tensors, elems in [("llama3-8b", 224, 6_979_321_856), ("llama3-70b", 560, 68_451_041_280)]:
        shapes = [s for _ in range(bench.MODELS[model][2]) for s in bench.layer_shapes(model)]
        assert len(shapes) == n_tensors
        assert sum(r * c for r, c in shapes) == elems
    assert len(bench.layer_shapes("mixtral-8x7b")) == 4 + 3 * 8


@pytest.mark.parametrize("world", [1, 2, 4, 8])
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_pool_partition(world, scaling, monkeypatch):
    """Every pool tensor lives on exactly one rank; weak: every rank holds one model's worth (same shapes in the same
    order, different values), strong: one model's list dealt round-robin; a tensor's values depend on its pool index only."""
    monkeypatch.setitem(bench.MODELS, "tiny", (16, 24, 3, 8))
    per_rank = [bench.make_weights("tiny", 3, "cpu", rank=r, world=world, scaling=scaling) for r in range(world)]
    n_model = 3 * 7
    pool = n_model * (world if scaling == "weak" else 1)
    assert all(n == pool for _, _, n in per_rank)
    owned = sorted(i for _, idx, _ in per_rank for i in idx)
    assert owned == list(range(pool))
    shapes = [s for _ in range(3) for s in bench.layer_shapes("tiny")]
    for r, (ws, idx, _) in enumerate(per_rank):
        assert [tuple(w.shape) for w in ws] == [shapes[i % n_model] for i in idx]
        assert all(w.dtype == torch.bfloat16 for w in ws)
        if scaling == "weak":
            assert idx == list(range(r * n_model, (r + 1) * n_model))
        else:
            assert idx == list(range(r, n_model, world))
    if world > 1:
        # the same pool index gives the same tensor whoever generates it; different indices differ
        again, idx, _ = bench.make_weights("tiny", 3, "cpu", rank=1, world=world, scaling=scaling)
        assert all(torch.equal(a, b) for a, b in zip(again, per_rank[1][0]))
        if scaling == "weak":
            assert not torch.equal(per_rank[0][0][0], per_rank[1][0][0])
    if scaling == "strong" or world == 1:
        single, _, _ = bench.make_weights("tiny", 3, "cpu")
        for ws, idx, _ in per_rank:
            assert all(torch.equal(w, single[i]) for w, i in zip(ws, idx))


def test_committed_pmc_traffic_matches_the_algorithmic_bytes():
    """roofline.traffic of the default line comes from the committed PMC profile of the same launch: present, and within
    0.1 % of the 4 B/element the FP8 QDQ launch must move (anything above would be wasted re-reads)."""
    traffic, src = bench.pmc_traffic("fp8", "llama3-8b", 32)
    assert traffic is not None and os.path.exists(os.path.join(ROOT, src.split(",")[0].split(" ")[0]))
    alg = 6_979_321_856 * 4
    assert abs(traffic - alg) / alg < 1e-3
    assert bench.pmc_traffic("fp8", "llama3-8b", 4)[0] is None  # another workload size: no committed counter run
