"""Real quantisation (a15) on the GPU: FP8 / MXFP4 pack + unpack kernels and the QTensor mirrors, byte-exact against
the oracle and against FP8QTensor / MXFP4QTensor / INT4QTensor run by the reference on CPU
(tests/golden/qtensor.npz, tests/golden/int4.npz)."""

import pytest
import torch

import _moa_import
from conftest import DT, assert_bits_equal

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import ops, qtensor  # noqa: E402
from oracle import oracle  # noqa: E402

DEV = "cuda:0"


def _layout(c, x):
    if c["mode"] == "tensor":
        return 1, 1
    if c["mode"] == "axis0":
        return x.shape[0], x.shape[1]
    return x.numel() // 128, 128


def test_fp8_and_mxfp4_kernels_match_reference_run(golden):
    g = golden("qtensor")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        x = g.t(f"{k}_x", dt)
        want_q = torch.from_numpy(g.raw(f"{k}_q").copy())
        want_deq = g.t(f"{k}_deq", dt)
        if c["kind"] == "fp8":
            scales = g.t(f"{k}_scales", dt)
            got_q = ops.fp8_quantize(x.to(DEV), scales.to(DEV)).view(torch.uint8).cpu()
            assert torch.equal(got_q.reshape(-1), want_q.reshape(-1)), f"{k}: fp8 bytes differ from the reference"
            got = ops.fp8_dequantize(want_q.reshape(x.shape).to(DEV), scales.to(DEV), dt).cpu()
            assert_bits_equal(got, want_deq, f"{k}: fp8 dequant")
            # the QTensor mirror computes its own scales: same bytes, same scales
            kw = {"axis": 0} if c["mode"] == "axis0" else ({"block_sizes": {-1: 128}} if c["mode"] == "block128" else {})
            qt, sc = qtensor.FP8QTensor.quantize(x.to(DEV), **kw)
            assert list(sc.shape) == c["scale_shape"], f"{k}: scale shape {tuple(sc.shape)}"
            assert_bits_equal(sc.cpu(), scales, f"{k}: scales")
            assert torch.equal(qt._quantized_data.view(torch.uint8).cpu().reshape(-1), want_q.reshape(-1))
            deq = qt.dequantize(scale=sc, **({"block_sizes": {-1: 128}} if c["mode"] == "block128" else {}))
            assert_bits_equal(deq.cpu(), want_deq, f"{k}: FP8QTensor round trip")
        else:
            want_e = torch.from_numpy(g.raw(f"{k}_e8m0").copy())
            qt, e8 = qtensor.MXFP4QTensor.quantize(x.to(DEV), c["block"])
            assert torch.equal(e8.cpu(), want_e), f"{k}: e8m0 scales differ"
            assert torch.equal(qt._quantized_data.cpu(), want_q), f"{k}: mxfp4 bytes differ from the reference"
            deq = qt.dequantize(scale=e8, block_sizes={-1: c["block"]})
            assert_bits_equal(deq.cpu(), want_deq, f"{k}: MXFP4QTensor round trip")


@pytest.mark.parametrize("dn", ["f32", "f16", "bf16"])
def test_fp8_pack_unpack_vs_oracle(dn):
    dt = DT[dn]
    gen = torch.Generator().manual_seed(5)
    for shape, inner in [((64, 1024), None), ((96, 512), 512), ((40, 384), 128), ((17, 4096 + 64), 64)]:
        x = (torch.randn(*shape, generator=gen) * torch.exp(torch.randn(shape[0], 1, generator=gen))).to(dt)
        x.view(-1)[:6] = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), float("nan"), 1e-9], dtype=dt)
        if inner is None:
            scales, ax, inn = (x.float().abs().max() / 448.0).to(dt).reshape(1), 1, 1
            scales = torch.tensor([0.0123], dtype=dt)  # inf / nan in x: use a fixed finite scale
        else:
            ax, inn = x.numel() // inner, inner
            scales = (torch.rand(ax, generator=gen) * 0.01 + 1e-4).to(dt)
        want = oracle.fp8_pack(x, scales, ax, inn)
        got = ops.fp8_quantize(x.to(DEV), scales.to(DEV)).view(torch.uint8).cpu()
        assert torch.equal(got, want), f"fp8 pack {dn} {shape} inner={inner}"
        q = torch.randint(0, 256, shape, generator=gen, dtype=torch.uint8)
        want_d = oracle.fp8_unpack(q, scales, dt, ax, inn)
        got_d = ops.fp8_dequantize(q.to(DEV), scales.to(DEV), dt).cpu()
        assert_bits_equal(got_d, want_d, f"fp8 unpack {dn} {shape}")


def _every_16_bit_pattern(dt):
    return torch.arange(65536, dtype=torch.int32).to(torch.int16).view(dt)


@pytest.mark.parametrize("dn", ["f16", "bf16"])
def test_fp8_pack_every_16_bit_pattern(dn):
    """Every 16-bit input pattern under scales that put the packer's three per-packet levels next to each other (in range:
    |x| <= 441 * scale, no clamp / overflow patch; shared reciprocal with the patch; IEEE division): quotients at and around
    448 and 464 (where torch's cast turns into NaN), scales outside the shared division's window, negative and zero scales,
    fp32 scales (export) and a scale per row -- byte-exact against the oracle's literal `(x / scale).to(float8_e4m3fn)`."""
    dt = DT[dn]
    x = _every_16_bit_pattern(dt).repeat(2)  # 131 072 elements: 16 chunks
    big = 3.0e4 if dn == "f16" else 2.0**70
    for sc in (1.0, 0.0123, 2.0**-7, 3.3e-5, 117.0, 2.0**-14, big, -0.5, 0.0):
        s = torch.tensor([sc], dtype=dt)
        got = ops.fp8_quantize(x.to(DEV), s.to(DEV)).view(torch.uint8).cpu()
        assert torch.equal(got, oracle.fp8_pack(x, s)), f"fp8 pack {dn} scale {sc}"
        s32 = torch.tensor([sc * 1.0001], dtype=torch.float32)
        got = ops.fp8_quantize(x.to(DEV), s32.to(DEV), fp32_scales=True).view(torch.uint8).cpu()
        assert torch.equal(got, oracle.fp8_pack(x, s32, fp32_scales=True)), f"fp8 pack {dn} fp32 scale {sc}"
    # one scale per row of 512: each row's scale is its own abs-max / 448 (the export's choice: exactly one packet per row
    # leaves the in-range level) or a random one
    xr = x[torch.randperm(x.numel(), generator=torch.Generator().manual_seed(3))].reshape(256, 512)
    fin = torch.where(torch.isfinite(xr.float()), xr.float().abs(), torch.zeros(()))
    for rows_scale in ((fin.amax(dim=1) / 448.0).clamp_min(1e-6), torch.rand(256, generator=torch.Generator().manual_seed(4)) * 50 + 1e-3):
        s = rows_scale.to(dt)
        got = ops.fp8_quantize(xr.to(DEV), s.to(DEV)).view(torch.uint8).cpu()
        assert torch.equal(got, oracle.fp8_pack(xr, s, axis_size=256, inner=512)), f"fp8 pack {dn} per row"
        # the same through the 2-D tile packer: 1 x 512 tiles, scales of the tensor dtype and fp32 (promoted quotient)
        got = ops.fp8_quantize_tile(xr.to(DEV), s.to(DEV), 1, 512).view(torch.uint8).cpu()
        assert torch.equal(got, oracle.fp8_pack_tile(xr, s, 1, 512)), f"fp8 pack tile {dn}"
        got = ops.fp8_quantize_tile(xr.to(DEV), s.float().to(DEV), 1, 512).view(torch.uint8).cpu()
        assert torch.equal(got, oracle.fp8_pack_tile(xr, s.float(), 1, 512)), f"fp8 pack tile {dn} fp32 scales"


def test_fp8_pack_f32_boundaries():
    """fp32 tensors: quotients straddling 441 (the in-range test), 448 and 464, tiny numerators, signed zeros."""
    g = torch.Generator().manual_seed(9)
    for sc in (1.0, 0.37, 2.0**-20, 1234.5):
        base = torch.tensor([440.9, 441.0, 441.1, 447.9, 448.0, 448.1, 463.9, 464.0, 464.1, 1e-30, -1e-30, 0.0, -0.0, 1e-12,
                             2.0**-10, 2.0**-9, 3 * 2.0**-11, float("inf"), float("nan"), 65536.0, 65537.0])
        x = torch.cat([base * sc, -base * sc, torch.randn(8192 * 3 - 2 * base.numel(), generator=g) * 200 * sc]).float()
        x = x[torch.randperm(x.numel(), generator=g)]
        s = torch.tensor([sc], dtype=torch.float32)
        got = ops.fp8_quantize(x.to(DEV), s.to(DEV)).view(torch.uint8).cpu()
        assert torch.equal(got, oracle.fp8_pack(x, s)), f"fp8 pack f32 scale {sc}"


@pytest.mark.parametrize("dn", ["f32", "f16", "bf16"])
def test_int8_pack_rows_equals_the_reference_expression(dn):
    """export/quant_utils.py:868-869 -- (weight / wsf[:, None]).round().clamp(-128, 127).to(int8) with an fp32 factor per
    output channel, run by torch on the host: ties at .5 (half to even), the clamp edges, NaN -> 0, infinities, scales
    outside the shared division's window, negative scales, rows that are not a power of two long, ragged last chunk."""
    dt = DT[dn]
    g = torch.Generator().manual_seed(21)
    for rows, cols in ((64, 1024), (37, 1000), (5, 8192 + 8), (3, 24)):
        wsf = (torch.rand(rows, generator=g) * 0.02 + 1e-3).float()
        wsf[0] = 0.5  # exact halves below
        w = (torch.randn(rows, cols, generator=g) * wsf[:, None] * 60).to(dt)
        w[0, :16] = torch.tensor([0.25, 0.75, 1.25, -0.25, -0.75, 63.25, 63.75, -64.25, 64.0, -64.0, 63.5, -63.5, 1e4, -1e4, 0.0, -0.0]).to(dt)
        w[1, :4] = torch.tensor([float("nan"), float("inf"), -float("inf"), 1e-30]).to(dt)
        if rows > 2:
            wsf[2] = 1e-25 if dn != "f16" else 1e-9  # outside the window / tiny
        if rows > 4:
            wsf[3], wsf[4] = -0.01, 1e22
        want = (w / wsf[:, None]).round().clamp(-128, 127).to(torch.int8)
        got = ops.int8_pack_rows(w.to(DEV), wsf.to(DEV)).cpu()
        assert torch.equal(got, want), f"int8 pack {dn} {rows}x{cols}: {(got != want).sum().item()} differ"


@pytest.mark.parametrize("dn", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("block", [32, 16, 64, 8])
def test_mxfp4_pack_unpack_vs_oracle(dn, block):
    dt = DT[dn]
    gen = torch.Generator().manual_seed(block)
    for shape in [(64, 1024), (33, 320), (8200, 64)]:
        x = (torch.randn(*shape, generator=gen) * torch.exp(torch.randn(shape[0], 1, generator=gen) * 2)).to(dt)
        x[0, :block] = 0  # all-zero block: scale byte 0, every nibble 8
        x[1, :8] = torch.tensor([6.0, 3.0, 1.5, 0.75, 0.25, -0.25, 5.0, -2.5], dtype=dt)  # exact bounds: ties go down
        x[2, :block] = x[2, :block] * 1e-30  # tiny amax
        want_q, want_e = oracle.mxfp4_pack(x, block)
        got_q, got_e = ops.mxfp4_quantize(x.to(DEV), block)
        assert torch.equal(got_e.cpu(), want_e), f"mxfp4 e8m0 {dn} {shape} b={block}"
        assert torch.equal(got_q.cpu(), want_q), f"mxfp4 bytes {dn} {shape} b={block}"
        q = torch.randint(0, 256, want_q.shape, generator=gen, dtype=torch.uint8)
        e = torch.randint(0, 255, want_e.shape, generator=gen, dtype=torch.uint8)
        if dt == torch.float16:
            e = e.clamp(100, 140)  # keep 6 * 2^(e-127) inside the f16 range
        want_d = oracle.mxfp4_unpack(q, e, dt, block)
        got_d = ops.mxfp4_dequantize(q.to(DEV), e.to(DEV), dt, block).cpu()
        assert_bits_equal(got_d, want_d, f"mxfp4 unpack {dn} {shape} b={block}")


def test_mxfp4_power_of_two_edges():
    """amax / 6 exactly a power of two, and one / two fp32 ulps above it (where torch.log2 in fp32 still returns the
    integer): kernel and oracle agree on the exponent."""
    vals = []
    for k in (-20, -6, -1, 0, 3, 10):
        base = torch.tensor(6.0 * 2.0 ** k, dtype=torch.float32)
        bits = base.view(torch.int32)
        vals += [base, (bits + 1).view(torch.float32), (bits + 2).view(torch.float32), (bits + 40).view(torch.float32),
                 (bits - 1).view(torch.float32)]
    x = torch.zeros(len(vals), 32)
    x[:, 0] = torch.stack(vals)
    want_q, want_e = oracle.mxfp4_pack(x, 32)
    got_q, got_e = ops.mxfp4_quantize(x.to(DEV), 32)
    assert torch.equal(got_e.cpu(), want_e) and torch.equal(got_q.cpu(), want_q)


@pytest.mark.parametrize("dn", ["bf16", "f16"])
def test_mxfp4_pack_every_16_bit_pattern(dn):
    """Every FINITE 16-bit pattern of the dtype as an element, under block exponents that put it on every side of the seven
    E2M1 rounding bounds (exact ties included: the bounds are representable): nibbles and scale bytes equal the oracle's.
    Pins the key-table form of the rounding (moq_qtensor.hip, mxfp4_nibble).  Blocks with a non-finite abs-max are left
    out: the reference's ceil(log2(inf)) -> uint8 conversion is undefined there."""
    dt = DT[dn]
    pats = torch.arange(65536, dtype=torch.int32).to(torch.int16).view(dt)  # every pattern once
    pats = torch.cat([pats, torch.zeros(29, dtype=dt)])  # 65565 = 31 x 2115
    pats = torch.where(torch.isfinite(pats.float()), pats, torch.zeros((), dtype=dt))
    top = torch.finfo(dt).max / 64
    rows = []
    for shift in (0, 5, -9):  # the block's first element sets the exponent: 2^shift times the row's own magnitude
        x = pats.reshape(-1, 31).clone()
        lead = (x.float().abs().amax(dim=1, keepdim=True).clamp(2.0 ** -20, top) * 2.0 ** shift)
        rows.append(torch.cat([lead.to(dt), x], dim=1))
    x = torch.cat(rows)
    want_q, want_e = oracle.mxfp4_pack(x, 32)
    got_q, got_e = ops.mxfp4_quantize(x.to(DEV), 32)
    assert torch.equal(got_e.cpu(), want_e), f"{dn}: scale bytes differ"
    assert torch.equal(got_q.cpu(), want_q), f"{dn}: {int((got_q.cpu() != want_q).sum())} packed bytes differ"


def test_int4_qtensor_round_trip_matches_kernels():
    gen = torch.Generator().manual_seed(3)
    w = (torch.randn(64, 256, generator=gen) * 0.02).to(torch.bfloat16)
    qt, scales = qtensor.INT4QTensor.quantize(w.to(DEV), 128)
    assert qt._quantized_data.shape == (64, 128) and scales.shape == (128, 1)
    want = oracle.int4_pack(w.reshape(-1), scales.cpu().reshape(-1), 128, rounding=1)
    assert torch.equal(qt._quantized_data.cpu().reshape(-1), want)
    deq = qt.dequantize(scale=scales, block_sizes={-1: 128})
    assert deq.shape == w.shape and deq.dtype == w.dtype
    assert (deq.cpu().float() - w.float()).abs().max() <= (w.float().abs().max() / 7) * 0.51


# the reference's own literal vectors for real quantization: tests/gpu/torch/quantization/test_qtensor_cuda.py:110-250
# (test_qtensor_accuracy: quantize, dequantize, torch.allclose with the listed values; bf16 tensors)
_QT_LITERALS = [
    ("int4", {-1: 4}, None, [[0, 1, 2, 3, 4, 5, 6, 7]], [[0.0000, 0.8516, 2.1406, 2.9844, 4.0000, 5.0000, 6.0000, 7.0000]]),
    ("int4", {-1: 4}, None, [[0, 1, 2, 3, 4, 5, 6, 7, 3, 3]],
     [[0.0000, 0.8516, 2.1406, 2.9844, 4.0000, 5.0000, 6.0000, 7.0000, 2.9844, 2.9844]]),
    ("fp8", {-1: 2, -2: 2}, None, [[0, 1, 2, 3], [4, 5, 6, 7]], [[0.0000, 0.9844, 2.0000, 3.0000], [3.9375, 5.0000, 6.0000, 7.0000]]),
    ("fp8", {-1: 2, -2: 2}, None, [[0, 1, 3], [4, 5, 7]], [[0.0000, 0.9844, 3.0000], [3.9375, 5.0000, 7.0000]]),
    ("fp8", {-1: 2}, None, [[0, 1, 2, 3], [4, 5, 6, 7]], [[0.0000, 1.0000, 1.9219, 3.0000], [3.9375, 5.0000, 6.0000, 7.0000]]),
    ("fp8", None, 0, [[0, 1, 2, 3], [4, 5, 6, 7]], [[0.0000, 0.9609, 1.9219, 3.0000], [4.0000, 5.0000, 6.0000, 7.0000]]),
    ("fp8", None, None, [[0, 1, 2, 3], [4, 5, 6, 7]], [[0, 1, 2, 3], [4, 5, 6, 7]]),
]


@pytest.mark.parametrize("kind,block_sizes,axis,test_input,test_output", _QT_LITERALS)
def test_reference_literal_vectors_for_real_quantization(kind, block_sizes, axis, test_input, test_output):
    x = torch.tensor(test_input, dtype=torch.bfloat16, device=DEV)
    want = torch.tensor(test_output, dtype=torch.bfloat16)
    if kind == "int4":
        qt, scales = qtensor.INT4QTensor.quantize(x, block_sizes[-1])
        deq = qt.dequantize(torch.bfloat16, scale=scales, block_sizes=block_sizes)
    else:
        qt, scales = qtensor.FP8QTensor.quantize(x, None, axis=axis, block_sizes=block_sizes)
        deq = qt.dequantize(torch.bfloat16, scale=scales, block_sizes=block_sizes)
    assert deq.shape == want.shape
    assert torch.allclose(deq.cpu(), want), f"{deq.cpu()} vs {want}"
