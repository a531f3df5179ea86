"""AWQ-lite search modes on the GPU: the Gram-matrix search (one forward pass, trace(E G E^T)) against the error-GEMM
search (the reference's structure: one fused MFMA GEMM per alpha and batch) on Llama-shaped random linears -- same
best alpha, losses within the reference's own bf16 rounding floor."""

import copy

import pytest
import torch

import _moa_import

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import model_quant  # noqa: E402

DEV = "cuda:0"


class Stack(torch.nn.Module):
    def __init__(self, dims, dtype):
        super().__init__()
        g = torch.Generator().manual_seed(7)
        self.linears = torch.nn.ModuleList()
        for co, ci in dims:
            lin = torch.nn.Linear(ci, co, bias=False)
            with torch.no_grad():
                w = torch.randn(co, ci, generator=g) * 0.02
                lin.weight.copy_(torch.where(torch.rand(co, ci, generator=g) < 0.001, w * 8, w))
            self.linears.append(lin)
        self.to(dtype)

    def forward(self, xs):
        return [lin(x) for lin, x in zip(self.linears, xs)]


def _batches(dims, dtype, n, tokens):
    g = torch.Generator().manual_seed(11)
    out = []
    for _ in range(n):
        xs = []
        for _, ci in dims:
            ch = torch.exp(torch.randn(ci, generator=g))
            ch[:4] *= 50
            xs.append((torch.randn(tokens, ci, generator=g) * ch).to(dtype).to(DEV))
        out.append(xs)
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_gram_search_equals_gemm_search(dtype):
    dims = [(512, 1024), (1024, 512), (256, 1536)]
    batches = _batches(dims, dtype, 3, 200)  # 200 tokens: exercises the pad-to-8 of the transpose + MFMA path
    results = {}
    for mode in ("gemm", "gram"):
        model = Stack(dims, dtype).to(DEV)
        cfg = copy.deepcopy(model_quant.INT4_AWQ_CFG)
        cfg["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "search": mode}
        q = moa.quantize(model, cfg, lambda m: [m(b) for b in batches])
        results[mode] = [(lin.awq_lite.best_alpha, lin.awq_lite.loss_buf.cpu().clone(), lin.weight.detach().cpu().clone(),
                          lin.awq_lite.num_search_steps) for lin in q.linears]
    for (a_gemm, l_gemm, w_gemm, _), (a_gram, l_gram, w_gram, steps) in zip(results["gemm"], results["gram"]):
        rel = ((l_gram - l_gemm).abs() / l_gemm).max().item()
        assert rel <= (1e-4 if dtype == torch.float32 else 2e-2), f"{dtype}: Gram vs GEMM loss differs by {rel:.3e}"
        assert a_gram == a_gemm, f"{dtype}: best alpha {a_gram} (gram) vs {a_gemm} (gemm)"
        assert torch.equal(w_gram, w_gemm)  # same alpha -> identical folded weights
        assert steps == 3


def test_quadform_split_precision_vs_fp64():
    """<E G, E> on the matrix cores in split bf16 precision against fp64, on shapes with ragged tiles."""
    from model_optimizer_amd import ops

    gen = torch.Generator().manual_seed(3)
    for cout, cin, tokens in [(512, 1024, 700), (300, 520, 2000), (64, 128, 50)]:
        ch = torch.exp(torch.randn(cin, generator=gen))
        ch[:4] *= 50
        x = torch.randn(tokens, cin, generator=gen) * ch
        g = (x.t() @ x / tokens).float()
        e = (torch.randn(cout, cin, generator=gen) * 1e-3).float()
        want = ((e.double() @ g.double()) * e.double()).sum().item() / cout
        acc = torch.zeros(1, device=DEV)
        ops.awq_quadform(e.to(DEV), ops.gram_operand(g.to(DEV)), acc, 1.0 / cout)
        got = acc.item()
        assert abs(got - want) <= 5e-5 * abs(want), f"{cout}x{cin}: {got} vs {want}"
        ops.awq_quadform(e.to(DEV), ops.gram_operand(g.to(DEV)), acc, 1.0 / cout)  # accumulates
        assert abs(acc.item() - 2 * want) <= 1e-4 * abs(want)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_err_weight_kernel_equals_composed_ops(dtype):
    """moq_awq_err_weight (one read of W) == awq_scale_qdq -> float -> * r -> - W and the bf16 split, bit for bit."""
    from model_optimizer_amd import ops

    gen = torch.Generator().manual_seed(5)
    for cout, cin, g in [(96, 512, 128), (33, 256, 64), (260, 8192 + 128, 128)]:
        w = (torch.randn(cout, cin, generator=gen) * 0.02).to(dtype).to(DEV)
        s = torch.exp(torch.randn(cin, generator=gen) * 0.5).to(dtype).to(DEV)
        r = (1 / s.float()).to(dtype).float()
        err, a = ops.awq_err_weight(w, s, r, g, 4)
        want = ops.awq_scale_qdq(w, s, g, 4).float().mul_(r).sub_(w.float())
        assert torch.equal(err, want), f"E differs ({cout}x{cin} g={g})"
        hi, lo = ops.split_bf16(want)
        assert torch.equal(a, torch.cat([hi, hi, lo], dim=1)), "split-precision operand differs"


class Experts(torch.nn.Module):
    """Three "experts"; the router sends tokens to experts 0 and 1 only, expert 1's input carries a NaN."""

    def __init__(self, dtype):
        super().__init__()
        self.stack = Stack([(256, 512), (256, 512), (256, 512)], dtype)

    def forward(self, xs):
        return [self.stack.linears[0](xs[0]), self.stack.linears[1](xs[1])]  # expert 2 never runs


@pytest.mark.parametrize("mode", ["gram", "gemm"])
def test_unexercised_and_nan_linears_fall_back_to_max_calibration(mode):
    """awq_lite's guard rails (model_calib.py:1605-1700): a linear that saw no tokens, or whose act scale holds a
    NaN, leaves the search, gets max-calibrated weights and a neutral pre_quant_scale; the others are searched as
    usual."""
    dtype = torch.bfloat16
    dims = [(256, 512)] * 3
    batches = _batches(dims, dtype, 2, 96)
    for b in batches:
        b[1][5, 7] = float("nan")
    model = Experts(dtype).to(DEV)
    w_before = [lin.weight.detach().clone() for lin in model.stack.linears]
    cfg = copy.deepcopy(model_quant.INT4_AWQ_CFG)
    cfg["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "search": mode}
    with pytest.warns(UserWarning, match="Forcing pre_quant_scale=1"):
        q = moa.quantize(model, cfg, lambda m: [m(b) for b in batches])
    l0, l1, l2 = q.stack.linears
    assert l0.awq_lite.is_enabled and l0.awq_lite.best_alpha is not None
    assert not torch.equal(l0.weight, w_before[0])  # scale folded into the weight
    for lin, w0 in ((l1, w_before[1]), (l2, w_before[2])):
        assert not lin.awq_lite.is_enabled and lin.awq_lite.best_scale is None
        assert torch.equal(lin.weight, w0)  # untouched
        pqs = lin.input_quantizer.pre_quant_scale
        assert pqs.dtype == dtype and torch.equal(pqs, torch.ones_like(pqs))
        want = w0.view(-1, 128).abs().amax(dim=1).float()
        assert torch.equal(lin.weight_quantizer.amax.float().reshape(-1), want)  # plain max calibration
    # the searched linear is what a run without the broken experts gives
    ref = Stack(dims[:1], dtype).to(DEV)
    qr = moa.quantize(ref, cfg, lambda m: [m(b[:1]) for b in batches])
    assert qr.linears[0].awq_lite.best_alpha == l0.awq_lite.best_alpha
    assert torch.equal(qr.linears[0].weight, l0.weight)


class SharedInputs(torch.nn.Module):
    """q / k / v read ONE tensor object (as in an attention block), o reads another of the same width."""

    def __init__(self, dtype):
        super().__init__()
        self.stack = Stack([(256, 512), (128, 512), (128, 512), (512, 512)], dtype)

    def forward(self, xs):
        q, k, v, o = self.stack.linears
        return [q(xs[0]), k(xs[0]), v(xs[0]), o(xs[1])]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_gram_matrix_is_shared_between_linears_with_the_same_input(dtype):
    """The Gram search accumulates X^T X once per distinct input tensor: k and v alias q's matrix, o (a different
    tensor of the same width) owns its own; the result equals a run where every linear gets its own copy."""
    dims = [(256, 512)] * 2
    batches = _batches(dims, dtype, 3, 96)
    cfg = copy.deepcopy(model_quant.INT4_AWQ_CFG)
    cfg["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "search": "gram"}
    shared = SharedInputs(dtype).to(DEV)
    calls = []
    from model_optimizer_amd import ops
    orig_add = ops.GramStage.add  # 16-bit inputs go through the staging buffer (several batches per Gram launch)
    ops.GramStage.add = lambda self, x2: (calls.append(1), orig_add(self, x2))[1]
    try:
        qs = moa.quantize(shared, cfg, lambda m: [m(b) for b in batches])
    finally:
        ops.GramStage.add = orig_add
    lin = qs.stack.linears
    assert lin[1].awq_lite.gram_owner is lin[0].awq_lite and lin[2].awq_lite.gram_owner is lin[0].awq_lite
    assert lin[0].awq_lite.gram_owner is None and lin[3].awq_lite.gram_owner is None
    if dtype != torch.float32:
        assert len(calls) == 2 * len(batches)  # q and o only
    # reference: the same linears, every one with a private clone of its input
    private = SharedInputs(dtype).to(DEV)
    private.forward = lambda xs: [l(x.clone()) for l, x in zip(private.stack.linears, (xs[0], xs[0], xs[0], xs[1]))]
    qp = moa.quantize(private, cfg, lambda m: [m(b) for b in batches])
    for a, b in zip(qs.stack.linears, qp.stack.linears):
        assert b.awq_lite.gram_owner is None
        assert a.awq_lite.best_alpha == b.awq_lite.best_alpha
        assert torch.equal(a.awq_lite.loss_buf, b.awq_lite.loss_buf)
        assert torch.equal(a.weight, b.weight)


class FlatStack(torch.nn.Module):
    """Plain Gaussian weights; with nearly uniform activation channels the 11 candidates score almost alike, so the
    roundings the Gram formulation leaves out decide the order: the near-tie case."""

    def __init__(self, dims, dtype, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.linears = torch.nn.ModuleList()
        for co, ci in dims:
            lin = torch.nn.Linear(ci, co, bias=False)
            with torch.no_grad():
                lin.weight.copy_(torch.randn(co, ci, generator=g) * 0.02)
            self.linears.append(lin)
        self.to(dtype)

    def forward(self, xs):
        return [lin(x) for lin, x in zip(self.linears, xs)]


def _flat_batches(dims, dtype, n, tokens, seed, spread):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        xs = []
        for _, ci in dims:
            ch = torch.exp(spread * torch.randn(ci, generator=g))
            xs.append((torch.randn(tokens, ci, generator=g) * ch).to(dtype).to(DEV))
        out.append(xs)
    return out


def _run_flat(dims, dtype, seed, spread, search, **kw):
    model = FlatStack(dims, dtype, seed).to(DEV)
    cfg = copy.deepcopy(model_quant.INT4_AWQ_CFG)
    cfg["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "search": search, **kw}
    b = _flat_batches(dims, dtype, 2, 200, seed + 100, spread)
    q = moa.quantize(model, cfg, lambda m: [m(x) for x in b])
    return [lin.awq_lite for lin in q.linears], [lin.weight.detach().clone() for lin in q.linears]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_auto_search_rescoring_near_ties_equals_gemm_search(dtype):
    """search="auto" (the default): near-ties of the Gram scores are re-scored by the exact-rounding error-GEMM engine
    and the selection equals search="gemm" on every linear (model_calib.py:1489-1495, :1548-1556, :1637) -- over seeds
    whose loss curves are flat enough that the plain Gram search flips some of them."""
    dims = [(256, 512), (512, 256), (128, 1024)]
    flipped = rescored = 0
    for seed in range(6):
        for spread in (0.02, 0.1):
            gemm, w_gemm = _run_flat(dims, dtype, seed, spread, "gemm")
            gram, _ = _run_flat(dims, dtype, seed, spread, "gram")
            auto, w_auto = _run_flat(dims, dtype, seed, spread, "auto")
            for hg, hm, ha, wg, wa in zip(gemm, gram, auto, w_gemm, w_auto):
                flipped += hm.best_alpha != hg.best_alpha
                assert ha.best_alpha == hg.best_alpha, f"seed {seed} spread {spread}: auto {ha.best_alpha} vs gemm {hg.best_alpha}"
                assert torch.equal(wa, wg)
                if ha.contenders is not None:
                    rescored += 1
                    for j, i in enumerate(ha.contenders):  # same kernel, same operands -> same number
                        assert float(ha.exact_buf[j]) == float(hg.loss_buf[i]) == float(ha.loss_buf[i])
    assert rescored > 0
    if dtype == torch.bfloat16:
        assert flipped > 0, "no seed where the plain Gram search disagrees: the test lost its teeth"


class AdversarialStack(torch.nn.Module):
    """Heavy-tailed weights (Student-t, 3 degrees of freedom) or plain Gaussian ones."""

    def __init__(self, dims, dtype, seed, student):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.linears = torch.nn.ModuleList()
        for co, ci in dims:
            lin = torch.nn.Linear(ci, co, bias=False)
            with torch.no_grad():
                w = torch.randn(co, ci, generator=g)
                if student:
                    w = w / torch.sqrt((torch.randn(3, co, ci, generator=g) ** 2).sum(0) / 3)
                lin.weight.copy_(w * 0.02)
            self.linears.append(lin)
        self.to(dtype)

    def forward(self, xs):
        return [lin(x) for lin, x in zip(self.linears, xs)]


def _adversarial_batches(dims, dtype, n, tokens, seed, massive):
    g = torch.Generator().manual_seed(seed)
    out = []
    for _ in range(n):
        xs = []
        for _, ci in dims:
            x = torch.randn(tokens, ci, generator=g) * torch.exp(0.02 * torch.randn(ci, generator=g))
            if massive:  # 1 % of the tokens are 60x larger than the rest
                x[torch.randint(0, tokens, (max(1, tokens // 100),), generator=g)] *= 60.0
            xs.append(x.to(dtype).to(DEV))
        out.append(xs)
    return out


def _run_adversarial(kind, seed, search, **kw):
    dims, dtype = [(256, 512), (512, 256), (128, 1024)], torch.bfloat16
    model = AdversarialStack(dims, dtype, seed, student=kind == "student").to(DEV)
    cfg = copy.deepcopy(model_quant.INT4_AWQ_CFG)
    cfg["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "search": search, **kw}
    b = _adversarial_batches(dims, dtype, 2, 200, seed + 100, massive=kind == "massive")
    q = moa.quantize(model, cfg, lambda m: [m(x) for x in b])
    from model_optimizer_amd import model_calib

    return [lin.awq_lite for lin in q.linears], dict(model_calib.AWQ_LITE_STATS)


@pytest.mark.parametrize("kind", ["student", "massive"])
def test_self_checking_margin_on_adversarial_distributions(kind, monkeypatch):
    """Heavy-tailed weights / activations with 1 % massive tokens, flat loss curves: the default search equals
    search="gemm" on every linear, the margin's self-check reports requirement / margin <= 1; and with the check forced to
    distrust every margin (an absurd TIE_SPREAD_FACTOR) the widening loop -- more (replayed) passes over the data, more candidates
    per pass, the last round scoring everything -- still ends on the same selection with the same exact scores."""
    from model_optimizer_amd import model_calib

    monkeypatch.setattr(model_calib._WeightCacheBudget, "host_bytes", 1 << 30)  # CPU re-run tier: Gram matrices on the host
    for seed in range(3):
        gemm, _ = _run_adversarial(kind, seed, "gemm")
        auto, st = _run_adversarial(kind, seed, "auto")
        assert [h.best_alpha for h in auto] == [h.best_alpha for h in gemm], f"{kind} seed {seed}"
        tc = st["tie_check"]
        assert tc["enabled"] and tc["max_need_over_margin"] <= 1.0 + 1e-9 or tc["widened_linears"] > 0
    monkeypatch.setattr(model_calib, "TIE_SPREAD_FACTOR", 1e6)
    gemm, _ = _run_adversarial(kind, 1, "gemm")
    forced, st = _run_adversarial(kind, 1, "auto", tie_margin=0.01)  # wide enough for near-ties on most linears
    assert [h.best_alpha for h in forced] == [h.best_alpha for h in gemm]
    widened = [h for h in forced if h.tie_rounds]
    assert widened and st["tie_check"]["widened_linears"] == len(widened) and st["passes"] + st.get("replayed_passes", 0) >= 3
    for h, hg in zip(forced, gemm):
        if h.tie_rounds:
            assert h.contenders == list(range(11))  # widened until every candidate was scored
            assert torch.equal(h.loss_buf, hg.loss_buf)  # each candidate scored exactly once, by the same kernel
    off, _ = _run_adversarial(kind, 1, "auto", tie_margin=0.01, tie_check=False)
    assert all(h.tie_rounds == 0 for h in off) and any(h.contenders is not None for h in off)
    assert sum(len(h.contenders or []) for h in off) < sum(len(h.contenders or []) for h in forced)


def test_auto_search_margins():
    dims, dt = [(256, 512), (128, 1024)], torch.bfloat16
    gemm, _ = _run_flat(dims, dt, 1, 0.02, "gemm")
    gram, _ = _run_flat(dims, dt, 1, 0.02, "gram")
    zero, _ = _run_flat(dims, dt, 1, 0.02, "auto", tie_margin=0.0)
    assert all(h.contenders is None for h in zero)
    assert [h.best_alpha for h in zero] == [h.best_alpha for h in gram]
    inf, _ = _run_flat(dims, dt, 1, 0.02, "auto", tie_margin=float("inf"))
    for hg, hi, hm in zip(gemm, inf, gram):
        assert hi.contenders == list(range(11)) and torch.equal(hi.loss_buf, hg.loss_buf)
        assert hi.gram_loss == [float(v) for v in hm.loss_buf.tolist()]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_col_abs_mean_accum_equals_host_get_act_scale(dtype):
    """acc += x.abs().mean(0).to(float32) exactly as the reference computes it on the host (model_calib.py:1471-1472,
    :1527): token counts that are not powers of two (torch's GPU tensor / scalar is a reciprocal multiply and differs),
    several batches accumulated."""
    from model_optimizer_amd import ops

    g = torch.Generator().manual_seed(5)
    cols = 1024
    acc = torch.zeros(cols, dtype=torch.float32, device=DEV)
    want = torch.zeros(cols, dtype=torch.float32)
    for tokens in (200, 77, 4099, 1):
        x = (torch.randn(tokens, cols, generator=g) * torch.exp(torch.randn(cols, generator=g))).to(dtype)
        ops.col_abs_mean_accum(x.to(DEV), acc)
        want += x.abs().contiguous().view(-1, cols).mean(0).to(torch.float32)
    got = acc.cpu()
    # the mean is an fp32 sum of <= 4099 values rounded ONCE to the dtype: a different summation order can only matter
    # when the sum sits within ~1e-6 of a rounding boundary -- none of these columns does (seeded); fp32 needs no luck
    # only up to the summation order itself
    if dtype == torch.float32:
        assert torch.allclose(got, want, rtol=2e-6, atol=0)
    else:
        assert torch.equal(got, want), f"{int((got != want).sum())} of {cols} columns differ"


def test_awq_lite_ragged_input_width_equals_the_reference_run(golden):
    """Cin = 192 with INT4 blocks of 128 on the GPU: the helper works on a zero-padded copy of the weight (whole blocks for
    the kernels) like the reference pads its last block; alpha, scales, folded weight, per-block amax and the
    fake-quantized output equal the reference run bit for bit (tests/golden/awq_ragged.npz)."""
    import replay_common

    replay_common.awq_ragged_check(moa, golden, DEV)


def test_the_hbm_budget_counts_what_torchs_allocator_holds_without_using():
    """A process that has freed something large sits on it in torch's caching allocator: the driver reports that memory as
    taken, yet the flow gets it on request.  Counting the driver's share alone sent tools/awq_bench.py's run -- which parks the
    flow's memory there before its clock starts -- into a second pass over the calibration data."""
    dev = torch.device(DEV)
    torch.cuda.empty_cache()
    before = moa.model_calib._WeightCacheBudget(dev).left
    held = torch.empty(8 << 30, dtype=torch.uint8, device=dev)
    during = moa.model_calib._WeightCacheBudget(dev).left
    del held  # back to torch's cache, not to the driver
    after = moa.model_calib._WeightCacheBudget(dev).left
    torch.cuda.empty_cache()
    assert during <= before - int(0.55 * (8 << 30))
    assert after >= before - (64 << 20), (before, during, after)
