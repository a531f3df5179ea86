"""SparseGPT on the GPU: MFMA Hessian accumulation, the column-sweep kernel and create_sgpt_mask
against the oracle and against the reference's own run (tests/golden/sgpt.npz: hook-accumulated Hessian, prepared
inverse factor and final mask, all produced by sparsity/weight_sparsity/sparsegpt.py on CPU)."""

import pytest
import torch

import _moa_import

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import ops, sparsity  # noqa: E402
from oracle import oracle  # noqa: E402

DEV = "cuda:0"
CFG = {"pattern": "2:4 sparsity", "col_block_size": 128, "row_block_size": -1, "hessian_damp": 0.1}


def test_transpose16():
    for shape in [(64, 64), (100, 264), (4096, 1000), (7, 9)]:
        x = torch.randn(*shape).to(torch.bfloat16)
        assert torch.equal(ops.transpose16(x.to(DEV)).cpu(), x.t().contiguous())


@pytest.mark.parametrize("name", ["sgpt_f32", "sgpt_bf16"])
def test_hessian_matches_reference_hook(golden, name):
    g = golden("sgpt")
    c = g.cases[name]
    dt = getattr(torch, c["dtype"])
    want = g.t(f"{name}_hessian")
    st = sparsity.HessianState(want.shape[0], DEV)
    for i in range(c["n_batches"]):
        st.update(g.t(f"{name}_x{i}", dt).to(DEV).unsqueeze(0))
    assert st.samples == c["samples"]
    got = st.hessian.cpu()
    # fp32 accumulation in a different order (MFMA tiles vs BLAS), scale applied after instead of before the product
    err = (got - want).abs().max().item() / want.abs().max().item()
    assert err <= 2e-6, f"{name}: Hessian differs by {err:.2e} of its range"
    assert torch.allclose(got, got.t(), rtol=0, atol=1e-6 * want.abs().max().item())


def test_hessian_mfma_exact_on_integer_inputs():
    """Small-integer activations: every product and partial sum is exact, so X^T X must be exact whatever the order."""
    gen = torch.Generator().manual_seed(0)
    x = torch.randint(-3, 4, (200, 320), generator=gen).to(torch.bfloat16)
    h = torch.zeros(320, 320, device=DEV)
    ops.hessian_accum(h, x.to(DEV), 0.0, 1.0)
    assert torch.equal(h.cpu(), x.float().t() @ x.float())
    ops.hessian_accum(h, x.to(DEV), 0.5, 2.0)
    assert torch.equal(h.cpu(), 2.5 * (x.float().t() @ x.float()))
    # upper-tile accumulation + one mirror pass at the end gives the same matrix
    for n in (320, 700, 64):
        xs = [torch.randint(-3, 4, (96, n), generator=gen).to(torch.bfloat16) for _ in range(2)]
        hu = torch.zeros(n, n, device=DEV)
        for xb in xs:
            ops.hessian_accum(hu, xb.to(DEV), 1.0, 1.0, upper_only=True)
        ops.symmetrize(hu)
        assert torch.equal(hu.cpu(), sum(xb.float().t() @ xb.float() for xb in xs)), n


@pytest.mark.parametrize("m,n", [(4, 2), (2, 1), (8, 4)])
def test_block_sweep_bit_exact_vs_oracle(m, n):
    gen = torch.Generator().manual_seed(m)
    for rows, ld, i1, bs in [(70, 256, 0, 128), (33, 384, 128, 128), (16, 200, 128, 72), (5, 64, 0, 64)]:
        w = (torch.randn(rows, ld, generator=gen) * 0.05).float()
        a = torch.randn(ld, ld, generator=gen)
        hinv = torch.linalg.cholesky(torch.linalg.inv(a @ a.t() / ld + 0.1 * torch.eye(ld)), upper=True).contiguous()
        w_ref = w.clone()
        d_ref = oracle.sgpt_block_sweep(w_ref, i1, bs, hinv, n, m)
        w_gpu = w.to(DEV)
        d_gpu = ops.sgpt_block_sweep(w_gpu, i1, bs, hinv.to(DEV), n, m)
        assert torch.equal(w_gpu.cpu(), w_ref), f"pruned block differs ({rows}x{ld} block {i1}+{bs}, {n}:{m})"
        assert torch.equal(d_gpu.cpu(), d_ref), f"delta differs ({rows}x{ld} block {i1}+{bs}, {n}:{m})"
        blk = w_ref[:, i1:i1 + bs].reshape(rows, -1, m)
        assert ((blk == 0).sum(-1) >= n).all()


@pytest.mark.parametrize("name", ["sgpt_f32", "sgpt_bf16"])
def test_create_sgpt_mask_matches_reference(golden, name):
    g = golden("sgpt")
    c = g.cases[name]
    dt = getattr(torch, c["dtype"])
    w = g.t(f"{name}_w", dt).to(DEV)
    want = torch.from_numpy(g.raw(f"{name}_mask").astype(bool))
    # (1) from the reference's prepared inverse factor: the sweep is bit-exact, the trailing fp32 GEMM differs in
    # summation order only -> (almost) every decision identical
    m1 = sparsity.create_sgpt_mask(w, None, CFG, hessian_inv=g.t(f"{name}_hinv").to(DEV)).cpu()
    agree = (m1 == want).float().mean().item()
    assert agree >= 0.995, f"{name}: {agree:.4f} of the mask equals the reference (given its Hinv)"
    # (2) end to end from the reference's Hessian (Cholesky inverse by the GPU library)
    m2 = sparsity.create_sgpt_mask(w, g.t(f"{name}_hessian").to(DEV), CFG).cpu()
    agree2 = (m2 == want).float().mean().item()
    assert agree2 >= 0.98, f"{name}: {agree2:.4f} of the mask equals the reference (from its Hessian)"
    for m in (m1, m2):
        assert m.dtype == torch.bool and m.shape == want.shape
        assert (m.view(m.shape[0], -1, 4).sum(-1) <= 2).all() and abs(m.float().mean().item() - 0.5) < 0.01


def test_sparsify_sparsegpt_flow():
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(256, 128, bias=False), torch.nn.GELU(), torch.nn.Linear(128, 64)).to(DEV).to(torch.bfloat16)
    batches = [torch.randn(2, 40, 256, device=DEV).to(torch.bfloat16) for _ in range(3)]
    ref_out = model(batches[0]).float()
    sparsity.sparsify(model, "sparsegpt", lambda m: [m(b) for b in batches])
    for lin in (model[0], model[2]):
        assert lin._weight_mask.dtype == torch.bool
        assert ((lin.weight != 0) <= lin._weight_mask).all()
        assert (lin._weight_mask.view(lin.weight.shape[0], -1, 4).sum(-1) <= 2).all()
    out = model(batches[0]).float()
    assert torch.isfinite(out).all() and (out - ref_out).abs().mean() < ref_out.abs().mean()
    model2 = torch.nn.Sequential(torch.nn.Linear(256, 128, bias=False)).to(DEV).to(torch.bfloat16)
    w_before = model2[0].weight.detach().clone()
    sparsity.sparsify(model2, "sparse_magnitude")
    assert torch.equal(model2[0]._weight_mask.cpu(), oracle.mask_2to4(w_before.cpu()))


class _Attn(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.q, self.k, self.v = (torch.nn.Linear(256, 128, bias=False) for _ in range(3))
        self.o = torch.nn.Linear(256, 64, bias=False)
        self.private = False

    def forward(self, x, y):
        if self.private:
            return self.q(x.clone()) + self.k(x.clone()) + self.v(x.clone()), self.o(y)
        return self.q(x) + self.k(x) + self.v(x), self.o(y)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_sparsegpt_hessian_shared_between_linears_with_the_same_input(dtype):
    """q / k / v read one tensor object: one Hessian accumulation and one inverse factor for the three; the masks equal
    those of a run where every linear gets a private copy of its input."""
    torch.manual_seed(3)
    batches = [(torch.randn(2, 48, 256, device=DEV).to(dtype), torch.randn(2, 48, 256, device=DEV).to(dtype)) for _ in range(3)]
    results = []
    for private in (False, True):
        torch.manual_seed(11)
        model = _Attn().to(DEV).to(dtype)
        model.private = private
        updates, inverts = [], []
        orig_u, orig_i = sparsity.HessianState.update, sparsity.invert
        sparsity.HessianState.update = lambda self, x: (updates.append(1), orig_u(self, x))[1]
        sparsity.invert = lambda h: (inverts.append(1), orig_i(h))[1]
        try:
            sparsity.sparsify(model, "sparsegpt", lambda m: [m(*b) for b in batches])
        finally:
            sparsity.HessianState.update, sparsity.invert = orig_u, orig_i
        assert (len(updates), len(inverts)) == ((4 * 3, 4) if private else (2 * 3, 2))
        results.append({n: m._weight_mask.clone() for n, m in model.named_modules() if hasattr(m, "_weight_mask")})
    assert set(results[0]) == {"q", "k", "v", "o"}
    for n in results[0]:
        assert torch.equal(results[0][n], results[1][n]), n
