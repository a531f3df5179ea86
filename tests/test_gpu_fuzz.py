"""Seeded fuzz of the main entries against the oracle: ragged shapes, every dtype, and tensors whose first element is
NOT 16-byte aligned (a contiguous slice of a larger buffer) -- the packet kernels must either take their unaligned
path or the host must route around them; a wrong answer or a refusal is a failure."""

import pytest
import torch

import _moa_import
from conftest import DT, assert_bits_equal

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import ops  # noqa: E402
from oracle import oracle  # noqa: E402

DEV = "cuda:0"


def _case(seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    dn = ["bf16", "f16", "f32"][r(0, 2)]
    rows = r(1, 70)
    cols = [r(1, 40) * 8, r(1, 300), 128 * r(1, 6), 4 * r(1, 64)][r(0, 3)]
    off = [0, 0, 1, 3, 8][r(0, 4)]
    base = (torch.randn(rows * cols + off, generator=g) * torch.exp(torch.randn(1, generator=g) * 2)).to(DT[dn])
    if r(0, 3) == 0:
        base[r(0, base.numel() - 1)] = 0.0
    x_cpu = base[off:].view(rows, cols)
    x_gpu = base.to(DEV)[off:].view(rows, cols)
    assert x_gpu.is_contiguous()
    return dn, rows, cols, off, x_cpu, x_gpu, g


@pytest.mark.parametrize("seed", range(60))
def test_fuzz_entries_vs_oracle(seed):
    dn, rows, cols, off, x, xg, g = _case(seed)
    tag = f"seed {seed}: {dn} [{rows}, {cols}] offset {off}"
    am = ops.reduce_amax(xg)
    assert_bits_equal(am.float().reshape(()), oracle.reduce_amax(x).float().reshape(()), f"{tag} amax")
    a32 = am.float().reshape(1)
    assert_bits_equal(ops.scaled_e4m3(xg, a32), oracle.fake_quant_e4m3(x, a32.cpu()), f"{tag} fp8")
    assert_bits_equal(ops.fake_tensor_quant(xg, a32, 8, False, True), oracle.fake_quant_int(x, a32.cpu(), 8, False, True),
                      f"{tag} int8")
    rows_am = ops.reduce_amax(xg, axis=[1]).float().reshape(-1)
    assert_bits_equal(rows_am.cpu(), oracle.reduce_amax_axis(x, 1, rows, cols).reshape(-1), f"{tag} row amax")
    assert_bits_equal(ops.fake_tensor_quant_with_axis(xg, rows_am, 0, 8, False, False),
                      oracle.fake_quant_int(x, rows_am.cpu(), 8, False, False, axis_size=rows, inner=cols, per_axis=True),
                      f"{tag} int8 per row")
    for block, fmt in ((32, "E2M1"), (16, "E4M3")):
        assert_bits_equal(ops.fused_amax_convert(xg, block, fmt), oracle.mx_fused_amax_convert(x, block, fmt),
                          f"{tag} mx {fmt}/{block}")
    if cols % 128 == 0:
        y, ga = ops.amax_qdq_int_group(xg, 128, num_bits=4, narrow_range=False)
        wy, wa = oracle.amax_qdq_int_group(x, 128, num_bits=4, narrow_range=False)
        assert_bits_equal(y, wy, f"{tag} int4 g128")
        assert_bits_equal(ga.cpu(), wa, f"{tag} group amax")
    if cols % 4 == 0:
        assert torch.equal(ops.mask_2to4(xg).cpu(), oracle.mask_2to4(x)), f"{tag} 2:4 mask"
    mx = float(a32)
    if mx > 0:
        h = ops.hist_abs(xg, 512, mx)
        assert torch.equal(h.cpu(), torch.from_numpy(oracle.hist_abs(x, 512, mx).astype("int64"))), f"{tag} hist"
    s = torch.exp(torch.randn(cols, generator=g) * 0.3)
    assert_bits_equal(ops.scale_cols(xg, s.to(DEV)), oracle.scale_cols(x, s), f"{tag} scale_cols")
    ssum, camax = ops.col_abs_stats(xg)
    s64, cam = oracle.col_abs_stats(x)
    assert_bits_equal(camax.cpu(), cam, f"{tag} col amax")
    assert ((ssum.cpu().double() - s64).abs() <= 1e-5 * s64.clamp_min(1e-30)).all(), f"{tag} col sum"


@pytest.mark.parametrize("seed", range(40))
def test_fuzz_packers_vs_oracle(seed):
    """Real-quant packers on ragged / unaligned inputs: INT4 pack + unpack, FP8 pack + unpack (per tensor and per row),
    MXFP4 pack + unpack, the checkpoint packer."""
    g = torch.Generator().manual_seed(1000 + seed)
    r = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    dn = ["bf16", "f16", "f32"][r(0, 2)]
    dt = DT[dn]
    rows, cols = 2 * r(1, 24), 128 * r(1, 4)
    off = [0, 0, 1, 3, 8][r(0, 4)]
    base = (torch.randn(rows * cols + off, generator=g) * 0.05).to(dt)
    x, xg = base[off:].view(rows, cols), base.to(DEV)[off:].view(rows, cols)
    tag = f"seed {seed}: {dn} [{rows}, {cols}] offset {off}"
    am = x.float().view(-1, 128).abs().amax(1).clamp_min(1e-6)
    scales = (7.0 / am).to(dt)
    q4 = ops.int4_quantize(xg.reshape(-1), scales.to(DEV), 128)
    assert torch.equal(q4.cpu(), oracle.int4_pack(x.reshape(-1), scales, 128)), f"{tag} int4 pack"
    inv = (am / 7.0).to(dt)
    assert_bits_equal(ops.int4_dequantize(q4, inv.to(DEV), 128), oracle.int4_unpack(q4.cpu(), inv, 128), f"{tag} int4 unpack")
    wsf = (am / 7.0).view(rows, cols // 128)
    assert torch.equal(ops.pack_int4_in_uint8(xg, wsf.to(DEV)).cpu(), oracle.int4_pack_export(x, wsf)), f"{tag} export pack"
    s1 = (x.float().abs().max() / 448.0).reshape(1).to(dt)
    q8 = ops.fp8_quantize(xg, s1.to(DEV)).view(torch.uint8)
    assert torch.equal(q8.cpu(), oracle.fp8_pack(x, s1)), f"{tag} fp8 pack"
    assert_bits_equal(ops.fp8_dequantize(q8, s1.to(DEV), dt), oracle.fp8_unpack(q8.cpu(), s1, dt), f"{tag} fp8 unpack")
    sr = (x.float().abs().amax(1, keepdim=True).clamp_min(1e-6) / 448.0).to(dt)
    q8r = ops.fp8_quantize(xg, sr.to(DEV)).view(torch.uint8)
    assert torch.equal(q8r.cpu(), oracle.fp8_pack(x, sr, axis_size=rows, inner=cols)), f"{tag} fp8 pack per row"
    qm, em = ops.mxfp4_quantize(xg, 32)
    om, oe = oracle.mxfp4_pack(x, 32)
    assert torch.equal(qm.cpu(), om) and torch.equal(em.cpu(), oe), f"{tag} mxfp4 pack"
    assert_bits_equal(ops.mxfp4_dequantize(qm, em, dt, 32), oracle.mxfp4_unpack(om, oe, dt, 32), f"{tag} mxfp4 unpack")
