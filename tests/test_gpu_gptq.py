"""GPTQ on the GPU: the column-sweep kernel against the oracle bit for bit, the blockwise update against the
reference-run fixtures (tests/golden/gptq.npz), and the whole flow through mtq.quantize(algorithm = gptq)."""

import copy

import pytest
import torch

import _moa_import
from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import gptq, ops  # noqa: E402
from oracle import oracle  # noqa: E402

DEV = "cuda:0"
DT = {"float32": torch.float32, "bfloat16": torch.bfloat16}


def _factor(cols, seed):
    """An upper Cholesky factor of a damped inverse Hessian, like compute_hessian_inverse returns (on the host: the
    sweep is what is under test)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(3 * cols, cols, generator=g) * torch.exp(torch.randn(cols, generator=g) * 0.5)
    h = (2.0 / x.shape[0]) * x.t() @ x
    h += 0.01 * h.diag().mean() * torch.eye(cols)
    return torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(h)), upper=True).contiguous()


@pytest.mark.parametrize("fmt,bits,layout", [(1, 4, "g32"), (1, 4, "g128"), (1, 8, "row"), (1, 8, "tensor"), (2, 8, "tensor"),
                                             (2, 8, "row"), (1, 3, "g64")])
@pytest.mark.parametrize("rows,cols,bs", [(37, 256, 128), (64, 192, 96), (5, 128, 128), (130, 320, 64)])
def test_gptq_block_sweep_equals_the_oracle(fmt, bits, layout, rows, cols, bs):
    """Every block of a weight, swept by the kernel and by the oracle from the same state: quantized columns, errors and
    the running weights of the block agree bit for bit (same fp32 steps in the same order); groups with amax at or
    below 2^-24 (scale 0), values far outside the amax, ragged last blocks."""
    if layout.startswith("g") and cols % int(layout[1:]):
        pytest.skip("group does not divide the width")
    gen = torch.Generator().manual_seed(rows + cols + bits)
    w = (torch.randn(rows, cols, generator=gen) * 0.05).float()
    w[0, :8] = torch.tensor([0.0, -0.0, 5.0, -5.0, 1e-9, 0.33, -0.21, 1e4])
    hinv = _factor(cols, 7 + cols)
    if layout == "tensor":
        amax, stride, g = w.abs().amax().reshape(1), 0, cols
    elif layout == "row":
        amax, stride, g = w.abs().amax(dim=1), 1, cols
    else:
        g = int(layout[1:])
        amax, stride = w.reshape(rows, cols // g, g).abs().amax(dim=-1).reshape(-1), cols // g
    amax = amax.clone()
    amax.view(-1)[-1] = 1e-9  # at or below 2^-24: the quantizer's zero-scale branch
    wg, wo = w.clone().to(DEV), w.clone()
    hg = hinv.to(DEV)
    for i1 in range(0, cols, bs):
        b = min(bs, cols - i1)
        dg = ops.gptq_block_sweep(wg, i1, b, hg, amax.to(DEV), stride, g, fmt, bits)
        do = oracle.gptq_block_sweep(wo, i1, b, hinv, amax, stride, g, fmt, bits)
        assert_bits_equal(dg, do, f"errors of block {i1}")
        assert_bits_equal(wg[:, i1:i1 + b], wo[:, i1:i1 + b], f"quantized columns of block {i1}")
        if i1 + b < cols:
            ops.sgpt_trailing_update(wg, i1, dg, hg)
            oracle.sgpt_trailing_update(wo, i1, do, hinv)
            assert_bits_equal(wg, wo, f"weights after the trailing update of block {i1}")


@pytest.mark.parametrize("elem,g", [("E2M1", 32), ("E2M1", 16), ("E4M3", 32), ("E2M3", 64), ("INT8", 8)])
@pytest.mark.parametrize("rows,cols,bs", [(37, 256, 128), (64, 192, 64), (130, 320, 64)])
def test_gptq_block_sweep_mx_equals_the_oracle(elem, g, rows, cols, bs):
    """fmt 3: MX blocks whose E8M0 scale is recomputed from the block's current values at every column -- kernel and oracle
    from the same state, bit for bit; an all-zero block (scale 1), an inf (abs-max clamps to FLT_MAX) and a NaN among the
    values."""
    gen = torch.Generator().manual_seed(rows + cols + g)
    w = (torch.randn(rows, cols, generator=gen) * torch.exp(torch.randn(rows, 1, generator=gen))).float()
    w[1, :g] = 0
    w[2, 5] = float("inf")
    w[3, 70] = float("nan")
    hinv = _factor(cols, 11 + cols)
    wg, wo, hg = w.clone().to(DEV), w.clone(), hinv.to(DEV)
    for i1 in range(0, cols, bs):
        b = min(bs, cols - i1)
        dg = ops.gptq_block_sweep(wg, i1, b, hg, None, 0, g, 3, elem)
        do = oracle.gptq_block_sweep(wo, i1, b, hinv, None, 0, g, 3, elem)
        assert_bits_equal(dg, do, f"errors of block {i1}")
        assert_bits_equal(wg[:, i1:i1 + b], wo[:, i1:i1 + b], f"quantized columns of block {i1}")
        if i1 + b < cols:
            ops.sgpt_trailing_update(wg, i1, dg, hg)
            oracle.sgpt_trailing_update(wo, i1, do, hinv)


@pytest.mark.parametrize("elem,sfmt,g", [("E2M1", "E4M3", 16), ("E2M1", "E5M2", 32), ("E4M3", "E4M3", 32)])
def test_gptq_block_sweep_two_level_equals_the_oracle(elem, sfmt, g):
    """fmt 4: block scales in an element format relative to a tensor-wide amax (NVFP4-style), recomputed per column --
    kernel and oracle bit for bit, tensor-wide values below, at and far above the data's own maximum."""
    rows, cols, bs = 50, 256, 64
    gen = torch.Generator().manual_seed(77 + g)
    w = (torch.randn(rows, cols, generator=gen) * torch.exp(torch.randn(rows, 1, generator=gen))).float()
    w[1, :g] = 0
    hinv = _factor(cols, 5 + cols)
    for glob in (float(w.abs().max()), 0.5, 1e4):
        amax = torch.tensor([glob])
        wg, wo, hg = w.clone().to(DEV), w.clone(), hinv.to(DEV)
        for i1 in range(0, cols, bs):
            dg = ops.gptq_block_sweep(wg, i1, bs, hg, amax.to(DEV), 0, g, 4, elem, sfmt)
            do = oracle.gptq_block_sweep(wo, i1, bs, hinv, amax, 0, g, 4, elem, sfmt)
            assert_bits_equal(dg, do, f"errors of block {i1} (tensor-wide amax {glob})")
            assert_bits_equal(wg[:, i1:i1 + bs], wo[:, i1:i1 + bs], f"quantized columns of block {i1}")
            if i1 + bs < cols:
                ops.sgpt_trailing_update(wg, i1, dg, hg)
                oracle.sgpt_trailing_update(wo, i1, do, hinv)


@pytest.mark.parametrize("ones,expect_change", [(False, True), (True, False)])
def test_gptq_updates_like_the_reference_test(ones, expect_change):
    """tests/gpu/torch/quantization/test_gptq.py::test_gptq_updates of the reference, on the NVFP4-style preset it uses:
    after max calibration the weight is restored and GPTQ'd (perc_damp 0.1, block_size 16) -- a random weight must end
    away from its plain quantize-dequantize, an all-equal weight (no quantization error, nothing to compensate) on it."""
    torch.manual_seed(42)
    dim, block_size = 128, 16
    model = torch.nn.Sequential(torch.nn.Linear(dim, dim)).to(DEV)
    w = torch.ones(dim, dim, device=DEV) if ones else torch.randn(dim, dim, device=DEV)
    model[0].weight.data = w.clone()
    x = torch.randn(2, 16, dim, device=DEV)
    moa.quantize(model, copy.deepcopy(moa.model_quant.NVFP4_DEFAULT_CFG), lambda m: m(x))
    lin = model[0]
    q_dq = lin.weight_quantizer(lin.weight.data).clone()
    lin.weight.data = w.clone()
    gptq.gptq(model, lambda m: m(x), perc_damp=0.1, block_size=block_size)
    assert gptq.GPTQ_STATS["kernel_linears"] == 1  # two-level dynamic blocks under the calibrated tensor-wide amax: fmt 4
    if expect_change:
        assert not torch.allclose(lin.weight.data, q_dq), "Weight should not be equal"
    else:
        assert torch.allclose(lin.weight.data, q_dq), "Weight should be equal"


def test_mxfp4_gptq_flow_and_the_loop_through_the_quantizer_agree():
    """MXFP4 (E2M1, blocks of 32, E8M0 scales) weights through gptq(): the kernel path against the reference's column loop
    through the quantizer's own forward on the GPU, and against round-to-nearest on held-out inputs."""
    import model_optimizer_amd.gptq as G

    torch.manual_seed(9)
    d = 512
    lin = torch.nn.Linear(d, 384, bias=False).to(torch.bfloat16).to(DEV)
    mix = torch.randn(d, d, device=DEV) / d ** 0.5 * torch.linspace(3.0, 0.05, d, device=DEV)[:, None]
    batches = [(torch.randn(2, 256, d, device=DEV) @ mix).to(torch.bfloat16) for _ in range(4)]
    held = (torch.randn(512, d, device=DEV) @ mix).to(torch.bfloat16)
    cfg = copy.deepcopy(moa.model_quant.MXFP4_DEFAULT_CFG)
    cfg["quant_cfg"]["*input_quantizer"] = {"enable": False}
    with torch.no_grad():
        ref = lin(held).float()
    rtn = moa.quantize(torch.nn.Sequential(copy.deepcopy(lin)), copy.deepcopy(cfg), lambda m: [m(b) for b in batches])
    holder = torch.nn.Sequential(copy.deepcopy(lin))
    cfg_g = copy.deepcopy(cfg)
    cfg_g["algorithm"] = {"method": "gptq"}
    moa.quantize(holder, cfg_g, lambda m: [m(b) for b in batches])
    assert G.GPTQ_STATS["kernel_linears"] == 1
    w_kernel = holder[0].weight.data.float().clone()
    holder2 = torch.nn.Sequential(copy.deepcopy(lin))
    orig, G._static_layout = G._static_layout, (lambda q_, w_: None)
    try:
        moa.quantize(holder2, copy.deepcopy(cfg_g), lambda m: [m(b) for b in batches])
        assert G.GPTQ_STATS["kernel_linears"] == 0
    finally:
        G._static_layout = orig
    same = (w_kernel == holder2[0].weight.data.float()).float().mean().item()
    assert same >= 0.99, f"kernel path vs the loop through the quantizer: {same:.4f} identical"
    with torch.no_grad():
        e_rtn = float((rtn(held).float() - ref).pow(2).mean())
        e_gptq = float((holder(held).float() - ref).pow(2).mean())
    assert e_gptq < 0.8 * e_rtn, (e_gptq, e_rtn)


@pytest.mark.parametrize("name", ["int4_g128_bf16", "int4_g32_f32", "fp8_bf16", "int8_pc_f32"])
def test_blockwise_update_equals_the_reference_run(golden, name):
    """gptq_blockwise_update on the GPU from the reference's inverse factor: the updated weight is the reference's."""
    g = golden("gptq")
    c = g.cases[name]
    dt = DT[c["dtype"]]
    preset = {"int4_g128_bf16": "INT4_BLOCKWISE_WEIGHT_ONLY_CFG", "int4_g32_f32": "INT4_BLOCKWISE_WEIGHT_ONLY_CFG",
              "fp8_bf16": "FP8_DEFAULT_CFG", "int8_pc_f32": "INT8_DEFAULT_CFG"}[name]
    cfg = copy.deepcopy(getattr(moa.model_quant, preset))
    if name == "int4_g32_f32":
        for v in cfg["quant_cfg"].values():
            if isinstance(v, dict) and "block_sizes" in v:
                v["block_sizes"] = {-1: 32, "type": "static"}
    w = g.t(f"{name}_w", dt)
    lin = torch.nn.Linear(w.shape[1], w.shape[0], bias=False).to(dt)
    lin.weight.data.copy_(w)
    holder = torch.nn.Sequential(lin).to(DEV)
    batches = [g.t(f"{name}_x{i}", dt).to(DEV) for i in range(c["n_batches"])]
    moa.quantize(holder, cfg, lambda m: [m(b) for b in batches])
    q = holder[0]
    assert_bits_equal(q.weight_quantizer._amax.float().reshape(-1).cpu(), g.t(f"{name}_w_amax").reshape(-1), "weight amax")
    weight = q.weight.data.float().clone()
    info = gptq.gptq_blockwise_update(weight, g.t(f"{name}_hinv").to(DEV), c["block_size"], q.weight_quantizer)
    assert info["kernel"]
    assert_bits_equal(weight.to(dt).cpu(), g.t(f"{name}_wfinal", dt), f"updated weight {name}")
    # and the whole algorithm from the original weights: own Hessian (matrix cores for 16-bit inputs), own inverse
    q.weight.data.copy_(w.to(DEV))
    gptq.gptq(holder, lambda m: [m(b) for b in batches], perc_damp=c["perc_damp"], block_size=c["block_size"])
    got, want = q.weight.data.float().cpu(), g.t(f"{name}_wfinal", dt).float()
    assert (got != want).float().mean().item() <= 0.01, f"{int((got != want).sum())} of {got.numel()} weights differ"
    assert gptq.GPTQ_STATS["kernel_linears"] == 1


def test_gptq_on_a_decoder_sized_linear_shares_hessians_and_lowers_the_output_error():
    """q / k / v-like linears fed by ONE tensor: one Hessian for the three; the GPTQ'd weights reproduce the layer's
    outputs better than round-to-nearest (the point of the algorithm), checked on held-out inputs."""
    torch.manual_seed(5)
    d = 1024

    class Block(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q = torch.nn.Linear(d, d, bias=False)
            self.k = torch.nn.Linear(d, d // 4, bias=False)
            self.v = torch.nn.Linear(d, d // 4, bias=False)

        def forward(self, x):
            return self.q(x), self.k(x), self.v(x)

    model = Block().to(torch.bfloat16).to(DEV)
    # correlated input features (a decaying spectrum, as hidden states have): with independent channels the Hessian is
    # diagonal and there is nothing for the update to compensate with
    mix = torch.randn(d, d, device=DEV) / d ** 0.5 * torch.linspace(3.0, 0.05, d, device=DEV)[:, None]
    batches = [(torch.randn(4, 256, d, device=DEV) @ mix).to(torch.bfloat16) for _ in range(4)]
    held = (torch.randn(512, d, device=DEV) @ mix).to(torch.bfloat16)
    with torch.no_grad():
        ref = [o.float() for o in model(held)]
    cfg = copy.deepcopy(moa.model_quant.INT4_BLOCKWISE_WEIGHT_ONLY_CFG)
    rtn = moa.quantize(copy.deepcopy(model), copy.deepcopy(cfg), lambda m: [m(b) for b in batches])
    with torch.no_grad():
        err_rtn = [float((o.float() - r).pow(2).mean()) for o, r in zip(rtn(held), ref)]
    cfg["algorithm"] = {"method": "gptq"}
    moa.quantize(model, cfg, lambda m: [m(b) for b in batches])
    st = gptq.GPTQ_STATS
    assert st["linears"] == 3 and st["kernel_linears"] == 3 and st["shared_hessians"] == 2
    with torch.no_grad():
        err_gptq = [float((o.float() - r).pow(2).mean()) for o, r in zip(model(held), ref)]
    assert all(g < 0.8 * r for g, r in zip(err_gptq, err_rtn)), (err_gptq, err_rtn)
