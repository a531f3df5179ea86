"""GPU parity of the AWQ-lite search error GEMM (moq_awq_err_gemm / moq_gemm_nt, MFMA) against the CPU oracle.

The contraction is floating point with a hardware-defined accumulation order, so this is the one kernel of the
path that is compared with a stated tolerance instead of bit-exactly:
  * exact-arithmetic inputs (small integers: every partial sum is representable) must match bit for bit --
    this pins the MFMA fragment / C-layout mapping, the LDS swizzle and the ragged-edge handling;
  * random inputs: fp32 accumulation-order noise is <= ~K * 2^-24 relative to sum|x*w|, which can flip the
    final rounding to the model dtype of a few outputs by one ulp.  Stated tolerances: stored outputs within
    one model-dtype ulp of the oracle and >= 99% bit-identical; loss within rtol 2e-3.
"""

import pytest
import torch

import _moa_import

pytestmark = pytest.mark.gpu

moa = _moa_import.load()
ops = moa.ops
from oracle import oracle  # noqa: E402  (the checker)

DEV = "cuda:0"
SHAPES = [  # (tokens, cout, cin): ragged tokens / cout, K tail (cin % 64 != 0), multi-tile, single tiny tile
    (1, 4, 8), (5, 12, 24), (128, 128, 64), (130, 132, 72), (257, 384, 200), (64, 256, 1024), (300, 260, 136),
    # cout % 8 == 0 (the loss epilogue stages out_actual's tile through the LDS) with ragged tokens and partial column tiles
    (300, 264, 136), (513, 520, 72)]


def _ints(shape, dtype, seed, lo=-3, hi=4):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).to(dtype)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", SHAPES)
def test_gemm_nt_exact_on_integer_inputs(dtype, shape):
    t, n, k = shape
    x, w, b = _ints((t, k), dtype, 1), _ints((n, k), dtype, 2), _ints((n,), dtype, 3)
    for bias in (None, b):
        _, want = oracle.awq_err_gemm(x, w, None, bias, return_out=True)
        got = ops.gemm_nt(x.to(DEV), w.to(DEV), None if bias is None else bias.to(DEV)).cpu()
        assert got.shape == want.shape and got.dtype == dtype
        bad = (got.view(torch.int16) != want.view(torch.int16)) & ~((got == 0) & (want == 0))
        assert not bad.any(), f"{shape} {dtype} bias={bias is not None}: {int(bad.sum())} of {got.numel()} outputs differ; " \
                              f"first at {bad.nonzero()[0].tolist()}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_err_gemm_out_actual_staged_or_direct_is_the_same(dtype):
    """out_actual 16-byte aligned: its tile goes through the LDS; 8-byte aligned only: read straight from memory.  Same loss,
    bit for bit (same cells, same order), on shapes with ragged tokens and a partial column tile."""
    for t, n, k in [(300, 264, 136), (257, 384, 200), (64, 256, 1024)]:
        x, w = _ints((t, k), dtype, 14), _ints((n, k), dtype, 15)
        ref = _ints((t, n), dtype, 16, -8, 9)
        xd, wd = x.to(DEV), w.to(DEV)
        buf = torch.zeros(t * n + 8, dtype=dtype, device=DEV)
        losses = []
        for shift in (0, 4):  # elements: 0 -> 16-byte aligned (staged), 4 -> 8-byte aligned (direct)
            rv = buf[shift:shift + t * n].view(t, n)
            rv.copy_(ref)
            assert rv.data_ptr() % 16 == (8 if shift else 0)
            acc = torch.zeros(1, dtype=torch.float32, device=DEV)
            ops.awq_err_gemm(xd, wd, rv, None, acc)
            losses.append(acc.item())
        assert losses[0] == losses[1], (t, n, k, losses)
        want = oracle.awq_err_gemm(x, w, ref)
        assert abs(losses[0] - want) <= 2e-6 * abs(want) + 1e-30


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", SHAPES)
def test_err_gemm_exact_on_integer_inputs(dtype, shape):
    """Integer data: out and the differences are exact, so the fused loss must equal the oracle's up to the
    final fp32 rounding of the mean."""
    t, n, k = shape
    x, w = _ints((t, k), dtype, 4), _ints((n, k), dtype, 5)
    ref = _ints((t, n), dtype, 6, -8, 9)
    want = oracle.awq_err_gemm(x, w, ref)
    acc = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.awq_err_gemm(x.to(DEV), w.to(DEV), ref.to(DEV), None, acc)
    ops.awq_err_gemm(x.to(DEV), w.to(DEV), ref.to(DEV), None, acc)  # accumulates: loss[alpha] += ...
    got = acc.item()
    assert abs(got - 2 * want) <= 2e-6 * abs(2 * want) + 1e-30, f"{shape} {dtype}: {got} vs {2 * want}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_random_within_one_ulp(dtype):
    torch.manual_seed(7)
    t, n, k = 192, 320, 4096
    x = torch.randn(t, k).to(dtype)
    w = (torch.randn(n, k) * 0.02).to(dtype)
    _, want = oracle.awq_err_gemm(x, w, None, None, return_out=True)
    got = ops.gemm_nt(x.to(DEV), w.to(DEV)).cpu()
    same = (got.view(torch.int16) == want.view(torch.int16)).float().mean().item()
    assert same >= 0.99, f"only {same:.4f} of outputs bit-identical"
    # one ulp of the model dtype at the output's magnitude, plus the fp32 accumulation-order noise that
    # dominates where the sum cancels to ~0 (bound K * 2^-24 * sum|x*w|; 2e-6 * sum|x*w| stated here --
    # the library GEMM shows the same few near-zero outliers against the fp64 oracle)
    ulp = torch.finfo(dtype).eps * want.float().abs()
    noise = 2e-6 * (x.float().abs() @ w.float().abs().T)
    assert ((got.float() - want.float()).abs() <= ulp * 1.001 + noise).all()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_err_gemm_awq_like_loss(dtype):
    """The search's actual use: w_hat = QDQ(W * s), xs = x / s, out_actual = x @ W^T; loss vs the oracle."""
    torch.manual_seed(11)
    t, n, k, g = 256, 256, 512, 128
    x = (torch.randn(t, k) * torch.exp(torch.randn(k))).to(dtype)
    w = (torch.randn(n, k) * 0.02).to(dtype)
    s = torch.exp(torch.randn(k) * 0.3).to(dtype)
    _, out_actual = oracle.awq_err_gemm(x, w, None, None, return_out=True)
    w_hat = oracle.awq_scale_qdq(w, s, g, 4)
    xs = oracle.scale_cols(x, (1 / s.float()).to(dtype).float())
    want = oracle.awq_err_gemm(xs, w_hat, out_actual)
    acc = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.awq_err_gemm(xs.to(DEV), w_hat.to(DEV), out_actual.to(DEV), None, acc)
    got = acc.item()
    assert want > 0 and abs(got - want) <= 2e-3 * want, f"{dtype}: loss {got} vs oracle {want}"


def test_err_gemm_rejects_unsupported():
    x = torch.randn(8, 16, device=DEV)
    w = torch.randn(8, 16, device=DEV)
    acc = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.awq_err_gemm(x, w, torch.randn(8, 8, device=DEV), None, acc)  # fp32 operands: the fp32 matrix-core kernel (round 3)
    with pytest.raises(ValueError):  # float64: MOQ_ERR_UNSUPPORTED -> ValueError, loud
        ops.awq_err_gemm(x.double(), w.double(), torch.randn(8, 8, device=DEV).double(), None, acc)
    xb = x.to(torch.bfloat16)
    with pytest.raises(ValueError):  # cin % 8 != 0
        ops.gemm_nt(xb[:, :12].contiguous(), w.to(torch.bfloat16)[:, :12].contiguous())


def test_err_gemm_full_size_linearity_property():
    """Llama-3-8B gate_proj shape at one calibration batch (4096 tokens): scaling out_actual and w_hat by 2
    scales every difference by exactly 2 (power-of-two scaling commutes with every rounding), so the loss
    must be exactly 4x -- a size-independent check at full size where the oracle is too slow."""
    torch.manual_seed(3)
    t, n, k = 4096, 14336, 4096
    x = torch.randn(t, k, device=DEV).to(torch.bfloat16)
    w = (torch.randn(n, k, device=DEV) * 0.02).to(torch.bfloat16)
    ref = ops.gemm_nt(x, (w.float() * 1.03).to(torch.bfloat16))
    a1 = torch.zeros(1, dtype=torch.float32, device=DEV)
    a2 = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.awq_err_gemm(x, w, ref, None, a1)
    ops.awq_err_gemm(x, w * 2, ref * 2, None, a2)
    assert a1.item() > 0 and abs(a2.item() - 4 * a1.item()) <= 1e-6 * a2.item()
    # and the deterministic reduction: same inputs, same bits
    a3 = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.awq_err_gemm(x, w, ref, None, a3)
    assert a3.item() == a1.item()


def test_err_gemm_multi_equals_single_launches():
    """The batched launch (all AWQ candidates of a linear in one grid) must give exactly the per-candidate
    results of single launches -- same tiles, same reduction order -- for strided and shared operands."""
    torch.manual_seed(5)
    a, t, n, k = 5, 200, 136, 328
    x = torch.randn(t, k, device=DEV).to(torch.bfloat16)
    w = (torch.randn(n, k, device=DEV) * 0.05).to(torch.bfloat16)
    ref = ops.gemm_nt(x, w)
    inv_s = torch.exp(torch.randn(a, k, device=DEV) * 0.2)
    xs = ops.scale_cols_multi(x, inv_s)
    for i in range(a):  # one read / A writes == A single-scale passes
        assert torch.equal(xs[i], ops.scale_cols(x, inv_s[i]))
    w_hat = torch.stack([ops.awq_scale_qdq(w, (1 / inv_s[i]).to(torch.bfloat16), 8, 4) for i in range(a)])
    multi = torch.zeros(a, dtype=torch.float32, device=DEV)
    ops.awq_err_gemm_multi(xs, w_hat, ref, None, multi)
    single = torch.zeros(a, dtype=torch.float32, device=DEV)
    for i in range(a):
        ops.awq_err_gemm(xs[i], w_hat[i], ref, None, single[i:i + 1])
    assert torch.equal(multi, single) and (multi > 0).all()
    shared = torch.zeros(a, dtype=torch.float32, device=DEV)
    ops.awq_err_gemm_multi(x, w_hat, ref, None, shared)  # x shared by all candidates (stride 0)
    for i in range(a):
        one = torch.zeros(1, dtype=torch.float32, device=DEV)
        ops.awq_err_gemm(x, w_hat[i], ref, None, one)
        assert shared[i].item() == one.item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("rows,cols", [(4096, 4096), (200, 328), (33, 8200), (1, 8), (8, 1024)])
def test_scale_cols_multi_is_the_rounded_product(dtype, rows, cols):
    """y[a] = dtype(x * s[a]) (one fp32 multiply, one rounding) for every candidate, on whole chunks, a ragged last chunk
    and tensors smaller than a chunk -- the pre-scaled activations x / s_alpha of the AWQ search (model_calib.py:1489-1495)."""
    g = torch.Generator().manual_seed(rows * 31 + cols)
    x = torch.randn(rows, cols, generator=g).to(dtype)
    s = torch.exp(torch.randn(7, cols, generator=g) * 0.3)
    got = ops.scale_cols_multi(x.to(DEV), s.to(DEV)).cpu()
    want = (x.float().unsqueeze(0) * s.unsqueeze(1)).to(dtype)
    assert got.shape == want.shape and got.dtype == dtype
    assert torch.equal(got.view(torch.int16 if dtype != torch.float32 else torch.int32),
                       want.view(torch.int16 if dtype != torch.float32 else torch.int32))


_GEO_PROBE = r"""
import hashlib, sys, torch
sys.path.insert(0, sys.argv[1])
import _moa_import
moa = _moa_import.load(); ops = moa.ops
g = torch.Generator().manual_seed(11)
h = hashlib.sha256()
for t, n, k in ((300, 260, 136), (512, 768, 1024), (257, 384, 200)):
    for dt in (torch.bfloat16, torch.float16):
        x = torch.randn(t, k, generator=g).to(dt).cuda(); w = (torch.randn(n, k, generator=g) * 0.05).to(dt).cuda()
        ref = torch.randn(t, n, generator=g).to(dt).cuda()
        h.update(ops.gemm_nt(x, w).cpu().view(torch.int16).numpy().tobytes())
        acc = torch.zeros(1, dtype=torch.float32, device="cuda"); ops.awq_err_gemm(x, w, ref, None, acc)
        h.update(acc.cpu().numpy().tobytes())
        hs = torch.zeros(k, k, dtype=torch.float32, device="cuda"); ops.hessian_accum(hs, x, 0.0, 1.0 / t)
        h.update(hs.cpu().numpy().tobytes())
print(h.hexdigest())
"""


def test_the_two_release_loop_structures_are_bit_identical():
    """MOQ_TUNE_GEMM_GEO is the one knob the release contraction reads: 4 selects the block-issue loop, anything else the
    default GEO 10 stream.  Same tile, same k order per accumulator: stored outputs, fused losses and Gram matrices must be
    the same bits (the variants that are NOT are no longer in the library: csrc/exp/)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for geo in ("4", "10", "9", "11"):  # 9: was a wrong-by-construction diagnostic; now means the default
        env = dict(os.environ, MOQ_TUNE_GEMM_GEO=geo)
        p = subprocess.run([sys.executable, "-c", _GEO_PROBE, root], capture_output=True, text=True, timeout=300, env=env)
        assert p.returncode == 0, p.stderr[-2000:]
        digests[geo] = p.stdout.strip().splitlines()[-1]
    assert len(set(digests.values())) == 1, digests


F32_SHAPES = [(1, 4, 8), (5, 12, 24), (130, 132, 72), (257, 384, 200), (64, 256, 1024), (300, 260, 136), (512, 128, 4096)]


@pytest.mark.parametrize("shape", F32_SHAPES)
def test_f32_gemm_on_the_fp32_matrix_cores(shape):
    """fp32 operands (fp32 models' AWQ search; moq_gemm_f32.hip, v_mfma_f32_32x32x2_f32): exact on integer data (tile /
    fragment / edge mapping), and on random data within the fp32 reordering bound of a double-precision reference --
    store epilogue, fused squared-error loss (two accumulating calls, bias) and the <E G, E> dot of the Gram search."""
    t, n, k = shape
    x, w, b = _ints((t, k), torch.float32, 1), _ints((n, k), torch.float32, 2), _ints((n,), torch.float32, 3)
    for bias in (None, b):
        want = x @ w.t() + (0 if bias is None else bias)
        got = ops.gemm_nt(x.to(DEV), w.to(DEV), None if bias is None else bias.to(DEV)).cpu()
        assert torch.equal(got, want), f"{shape}: integer fp32 GEMM not exact (bias={bias is not None})"
    g = torch.Generator().manual_seed(t * 31 + k)
    x, w = torch.randn(t, k, generator=g), torch.randn(n, k, generator=g) * 0.05
    bias, ref = torch.randn(n, generator=g) * 0.1, torch.randn(t, n, generator=g) * 0.1
    exact = x.double() @ w.double().t()
    bound = (k + 4) * 2.0 ** -24 * (x.abs().double() @ w.abs().double().t()) + 1e-30
    got = ops.gemm_nt(x.to(DEV), w.to(DEV)).cpu()
    assert ((got.double() - exact).abs() <= bound).all(), f"{shape}: fp32 GEMM outside the summation bound"
    acc = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.awq_err_gemm(x.to(DEV), w.to(DEV), ref.to(DEV), bias.to(DEV), acc)
    ops.awq_err_gemm(x.to(DEV), w.to(DEV), ref.to(DEV), bias.to(DEV), acc)
    want = 2 * ((exact + bias.double() - ref.double()) ** 2).mean().item()
    assert abs(acc.item() - want) <= 2e-5 * abs(want), f"{shape}: fused loss {acc.item()} vs {want}"
    # batched candidates: per-candidate operands, shared reference
    xs = torch.stack([x, x * 0.5, x * 2.0]).to(DEV)
    ws = torch.stack([w, w * 2.0, w * 0.25]).to(DEV)
    accs = torch.zeros(3, dtype=torch.float32, device=DEV)
    ops.awq_err_gemm_multi(xs, ws, ref.to(DEV), None, accs)
    for a in range(3):
        want = ((xs[a].cpu().double() @ ws[a].cpu().double().t() - ref.double()) ** 2).mean().item()
        assert abs(accs[a].item() - want) <= 2e-5 * abs(want)


@pytest.mark.parametrize("cout,cin", [(64, 128), (132, 264), (256, 1024)])
def test_f32_quadform_on_the_fp32_matrix_cores(cout, cin):
    g = torch.Generator().manual_seed(cin)
    err = torch.randn(cout, cin, generator=g) * 0.01
    a = torch.randn(cin, 2 * cin, generator=g)
    gram = (a @ a.t() / cin).contiguous()
    gram = ((gram + gram.t()) * 0.5).contiguous()
    acc = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.awq_quadform(err.to(DEV), gram.to(DEV), acc, 1.0 / cout)
    want = ((err.double() @ gram.double()) * err.double()).sum().item() / cout
    assert abs(acc.item() - want) <= 2e-5 * abs(want), f"{acc.item()} vs {want}"


def test_f32_gemm_full_size_scaling_property():
    """4096 x 4096 x 4096 in fp32 (one calibration batch of an fp32 Llama-3-8B attention projection): scaling w by 2 and
    out_actual by 2 scales every difference by exactly 2, so the fused loss must be exactly 4x; the stored product is
    within the fp32 summation bound of the library's."""
    torch.manual_seed(9)
    t = n = k = 4096
    x = torch.randn(t, k, device=DEV)
    w = torch.randn(n, k, device=DEV) * 0.02
    ref = ops.gemm_nt(x, w * 1.03)
    a1 = torch.zeros(1, dtype=torch.float32, device=DEV)
    a2 = torch.zeros(1, dtype=torch.float32, device=DEV)
    ops.awq_err_gemm(x, w, ref, None, a1)
    ops.awq_err_gemm(x, w * 2.0, ref * 2.0, None, a2)
    assert a2.item() == 4.0 * a1.item()
    lib = torch.nn.functional.linear(x, w * 1.03)
    bound = (k + 4) * 2.0 ** -23 * torch.nn.functional.linear(x.abs(), w.abs() * 1.03)
    assert ((ref - lib).abs() <= bound).all()
