"""Streaming kernels whose chunk bodies were split in round 3 (a branch-free copy for chunks inside the tensor, the
bounds-checked copy for the last one; row-group indexing of the FP8 tile packers; scale vectors requested up front;
two-level MX on the chunk skeleton; 2:4 mask without the table load) against the oracle on shapes that land on every
side of those splits: tensors smaller than a chunk (8192 elements), exactly one / two chunks, a full chunk plus a ragged
tail, row counts that are not multiples of the row group, column counts below / above / not dividing the chunk, tiles
narrower than a packet.  Bit-exact (bytes for packed results, bit patterns for floats)."""

import pytest
import torch

import _moa_import
from conftest import DT, assert_bits_equal

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import ops  # noqa: E402
from oracle import oracle  # noqa: E402

DEV = "cuda:0"
CHUNK = 8192


def _rand(shape, dt, seed, special=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g) * torch.exp(torch.randn(shape[0], 1, generator=g))
    if special and x.numel() >= 8:
        f = x.view(-1)
        f[0], f[1], f[2], f[3] = float("inf"), float("-inf"), 0.0, -0.0
    x = x.to(dt)
    if special and x.numel() >= 8:
        # a POSITIVE quiet NaN, written as a bit pattern after the conversion: torch's vectorised fp32 -> bf16 conversion on
        # this CPU turns float("nan") into 0xFFFF (negative), and the sign of a NaN byte of the FP8 packers follows the
        # sign of the NaN that reaches the cast -- which the kernels keep (hardware conversions) and the C oracle's bf16
        # rounding canonicalises (tools/exp/nan_sign_probe.py); only the sign-free case is a defined comparison
        f = x.view(-1)
        if dt == torch.float32:
            f.view(torch.int32)[-1] = 0x7FC00000
        else:
            f.view(torch.int16)[-1] = 0x7FC0 if dt == torch.bfloat16 else 0x7E00
    return x


# rows x cols: below a chunk, one chunk, two chunks, chunk + ragged tail, cols > chunk, cols not dividing the chunk
SHAPES = [(3, 8), (5, 24), (1, CHUNK), (2, CHUNK), (33, 264), (9, 1000), (3, 8200), (17, 4104), (2, 3 * CHUNK + 8)]


@pytest.mark.parametrize("dn", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("rows,cols", SHAPES)
def test_scale_and_rescale_cols_shapes(dn, rows, cols):
    dt = DT[dn]
    w = _rand((rows, cols), dt, rows * 131 + cols, special=False) * 0.05
    g = torch.Generator().manual_seed(cols)
    s = torch.exp(torch.randn(cols, generator=g) * 0.3)
    assert_bits_equal(ops.scale_cols(w.to(DEV), s.to(DEV)), oracle.scale_cols(w, s), "scale_cols")
    old, new = s.to(dt), torch.exp(torch.randn(cols, generator=g) * 0.3).to(dt)
    assert_bits_equal(ops.rescale_cols(w.to(DEV), old.to(DEV), new.to(DEV)), oracle.rescale_cols(w, old, new), "rescale_cols")


@pytest.mark.parametrize("dn", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("rows,cols,br,bc", [
    (6, 16, 2, 8), (5, 16, 1, 8), (7, 24, 7, 24), (130, 40, 65, 8), (129, 48, 3, 24), (4, CHUNK, 2, 128),
    (260, 264, 4, 8), (3, 2 * CHUNK + 64, 1, 64)])
@pytest.mark.parametrize("f32_scales", [False, True])
def test_fp8_tile_row_groups(dn, rows, cols, br, bc, f32_scales):
    """row counts that are not multiples of the kernel's four-row group, tile widths that are not powers of two"""
    dt = DT[dn]
    x = _rand((rows, cols), dt, rows * 17 + cols + br, special=False)
    x[0, 0], x[-1, -1] = float("inf"), 0.0
    amax = x.float().view(rows // br, br, cols // bc, bc).abs().amax(dim=(1, 3)).clamp(min=1e-3, max=1e4)
    scales = (amax / 448.0) if f32_scales else (amax.to(dt) / 448.0)
    got = ops.fp8_quantize_tile(x.to(DEV), scales.to(DEV), br, bc).view(torch.uint8).cpu()
    want = oracle.fp8_pack_tile(x, scales, br, bc)
    assert torch.equal(got, want), f"{(got != want).sum().item()} bytes differ"
    deq = ops.fp8_dequantize_tile(want.to(DEV), scales.to(dt).to(DEV), dt, br, bc).cpu()
    assert_bits_equal(deq, oracle.fp8_unpack_tile(want, scales.to(dt), dt, br, bc), "tile dequant")


@pytest.mark.parametrize("dn", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("n", [8, 4096, CHUNK, CHUNK + 8, 3 * CHUNK, 3 * CHUNK + 4096 + 8])
def test_fp8_pack_unpack_chunk_edges(dn, n):
    dt = DT[dn]
    x = _rand((1, n), dt, n).view(-1)
    for axis in (False, True):
        if axis:
            rows = 4 if n % 4 == 0 else 1
            xx = x.view(rows, n // rows)
            sc = (xx.float().abs().nan_to_num(0.0, 0.0, 0.0).amax(dim=1).clamp(min=1e-3) / 448.0).to(dt)
            got = ops.fp8_quantize(xx.to(DEV), sc.view(rows, 1).to(DEV)).view(torch.uint8).cpu()
            want = oracle.fp8_pack(xx, sc, axis_size=rows, inner=n // rows)
            deq = ops.fp8_dequantize(want.view(rows, -1).to(DEV), sc.view(rows, 1).to(DEV), dt).cpu()
            ref = oracle.fp8_unpack(want, sc, dt, axis_size=rows, inner=n // rows)
        else:
            sc = torch.tensor([0.37], dtype=dt)
            got = ops.fp8_quantize(x.to(DEV), sc.to(DEV)).view(torch.uint8).cpu()
            want = oracle.fp8_pack(x, sc)
            deq = ops.fp8_dequantize(want.to(DEV), sc.to(DEV), dt).cpu()
            ref = oracle.fp8_unpack(want, sc, dt)
        assert torch.equal(got.view(-1), want.view(-1)), f"axis={axis}: {(got.view(-1) != want.view(-1)).sum().item()} bytes differ"
        assert_bits_equal(deq.view(-1), ref.view(-1), f"axis={axis} dequant")


@pytest.mark.parametrize("dn", ["bf16", "f16"])
@pytest.mark.parametrize("n,g", [(128, 128), (CHUNK, 128), (CHUNK + 128, 128), (3 * CHUNK + 64, 64), (2 * CHUNK, 8)])
def test_int4_unpack_chunk_edges(dn, n, g):
    dt = DT[dn]
    x = _rand((1, n), dt, n + g, special=False).view(-1)
    scales = (7.0 / x.float().view(-1, g).abs().amax(dim=1).clamp(min=1e-3)).to(dt)
    q = oracle.int4_pack(x, scales, g)
    assert torch.equal(ops.int4_quantize(x.to(DEV), scales.to(DEV), g).cpu().view(torch.uint8).view(-1), q.view(-1))
    assert_bits_equal(ops.int4_dequantize(q.to(DEV), scales.to(DEV), g).cpu().view(-1), oracle.int4_unpack(q, scales, g).view(-1),
                      "int4 unpack")


@pytest.mark.parametrize("dn", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("rows,cols,block", [(2, 16, 16), (3, 48, 16), (1, CHUNK, 16), (5, CHUNK + 32, 32), (64, 264, 8),
                                             (4, 2 * CHUNK, 64)])
@pytest.mark.parametrize("with_global", [False, True])
def test_two_level_mx_on_the_chunk_skeleton(dn, rows, cols, block, with_global):
    """E2M1 elements with E4M3 block scales (and an optional tensor-wide abs-max): aligned shapes take the chunk kernel
    since round 3, the others the one-thread-per-block kernel -- both must equal the oracle"""
    dt = DT[dn]
    x = _rand((rows, cols), dt, rows + cols + block)
    ga = x.float().abs().nan_to_num(0.0, 0.0, 0.0).amax().reshape(1) if with_global else None
    want = oracle.mx_fused_amax_convert(x, block, "E2M1", "E4M3", ga)
    got = ops.fused_amax_convert(x.to(DEV), block, "E2M1", "E4M3", None if ga is None else ga.to(DEV))
    assert_bits_equal(got, want, "two-level MX")
    if cols % block == 0:  # an unaligned view of the same data: the generic kernel
        buf = torch.zeros(rows * cols + 1, dtype=dt, device=DEV)
        buf[1:] = x.to(DEV).view(-1)
        got2 = ops.fused_amax_convert(buf[1:].view(rows, cols), block, "E2M1", "E4M3", None if ga is None else ga.to(DEV))
        assert_bits_equal(got2, want, "two-level MX, unaligned")


@pytest.mark.parametrize("dn", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("fmt", ["E2M1", "E4M3", "E3M0", "E5M2", "INT8", "E2M3"])
def test_mx_rounding_specials_every_segment(dn, fmt):
    """the element rounding evaluates both segments and selects: values on and around the segment boundary, the format
    maximum, subnormals, +-0, +-inf and NaN, in blocks whose scale is 1 (abs-max = the format maximum)"""
    dt = DT[dn]
    fmax = {"E2M1": 6.0, "E4M3": 448.0, "E3M0": 16.0, "E5M2": 57344.0, "INT8": 127.0, "E2M3": 7.5}[fmt]
    vals = [0.0, -0.0, 1e-8, 0.24, 0.25, 0.26, 0.49, 0.5, 0.74, 0.75, 0.99, 1.0, 1.01, 1.24, 1.25, 1.26, 1.49, 1.5, 1.75, 2.0,
            2.49, 2.5, 2.51, 3.0, 3.49, 3.5, 5.0, 5.9, float("inf"), float("-inf"), float("nan")]
    rows = []
    for v in vals:
        rows.append([fmax, v, -v] + [0.015625 * k for k in range(29)])
    x = torch.tensor(rows, dtype=torch.float32).to(dt)
    want = oracle.mx_fused_amax_convert(x, 32, fmt)
    assert_bits_equal(ops.fused_amax_convert(x.to(DEV), 32, fmt), want, fmt)


@pytest.mark.parametrize("dn", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("rows,cols", [(1, 4), (3, 8), (1, CHUNK), (2, CHUNK + 4), (5, 3 * CHUNK + 12), (7, 1000)])
def test_mask_2to4_ties_and_specials(dn, rows, cols):
    """every group of four drawn from a small value set: ties between patterns in most groups, NaN / inf in some"""
    dt = DT[dn]
    g = torch.Generator().manual_seed(rows * 977 + cols)
    pool = torch.tensor([0.0, -0.0, 0.5, -0.5, 1.0, 1.0, 2.0, -2.0, float("inf"), float("nan")])
    w = pool[torch.randint(0, 8, (rows, cols), generator=g)]
    sp = torch.rand(rows, cols, generator=g) < 0.02
    w = torch.where(sp, pool[torch.randint(8, 10, (rows, cols), generator=g)], w).to(dt)
    want = oracle.mask_2to4(w)
    got = ops.mask_2to4(w.to(DEV)).cpu()
    assert torch.equal(got.view(torch.uint8), want.view(torch.uint8)), f"{(got != want).sum().item()} mask entries differ"
    # the multi-tensor launch over an unaligned second segment
    from model_optimizer_amd.multi_tensor import SegmentTable

    ws = [w.to(DEV), w.to(DEV).clone()]
    ms = [torch.empty(w.shape, dtype=torch.bool, device=DEV) for _ in ws]
    SegmentTable(ws, outputs=ms).mask_2to4()
    for m in ms:
        assert torch.equal(m.cpu().view(torch.uint8), want.view(torch.uint8))


@pytest.mark.parametrize("dn", ["bf16", "f16", "f32"])
@pytest.mark.parametrize("rows,cols", [(1, 4), (3, 8), (1, CHUNK), (2, CHUNK + 4), (5, 3 * CHUNK + 12), (7, 1000)])
def test_mask_2to4_apply_is_mask_then_multiply(dn, rows, cols):
    """moq_mt_mask_2to4_apply (sparsify's fused pass): the mask bytes equal the oracle's, the weight rewritten in place
    equals `w * mask` of the tensor multiply bit for bit -- a pruned negative weight is -0.0, inf / NaN times 0 is NaN --
    and, with calibrate=True, the per-tensor amax is the abs-max of that masked weight (NaN when it holds one).  Values
    from a small pool (ties between patterns in most groups), random magnitudes in a second tensor, an unaligned third."""
    from model_optimizer_amd.multi_tensor import SegmentTable

    dt = DT[dn]
    g = torch.Generator().manual_seed(rows * 131 + cols)
    pool = torch.tensor([0.0, -0.0, 0.5, -0.5, 1.0, 1.0, 2.0, -2.0, float("inf"), float("nan")])
    w0 = pool[torch.randint(0, 8, (rows, cols), generator=g)]
    sp = torch.rand(rows, cols, generator=g) < 0.02
    w0 = torch.where(sp, pool[torch.randint(8, 10, (rows, cols), generator=g)], w0).to(dt)
    w1 = (torch.randn(rows, cols, generator=g) * 0.02).to(dt)
    w2 = (torch.randn(rows * cols + 4, generator=g) * 3).to(dt)[4:].reshape(rows, cols)  # storage offset: not 16-byte aligned for 16-bit
    hosts = [w0, w1, w2.clone()]
    base = torch.empty(rows * cols + 4, dtype=dt, device=DEV)
    dev = [hosts[0].to(DEV), hosts[1].to(DEV), base[4:].reshape(rows, cols)]
    dev[2].copy_(hosts[2])
    masks = [torch.empty(h.shape, dtype=torch.bool, device=DEV) for h in hosts]
    tab = SegmentTable(dev, outputs=masks)
    tab.mask_2to4_apply(calibrate=True)
    for i, h in enumerate(hosts):
        want_mask = oracle.mask_2to4(h)
        want_w = h * want_mask.to(h.dtype)
        assert torch.equal(masks[i].cpu().view(torch.uint8), want_mask.view(torch.uint8)), f"tensor {i}: mask"
        assert_bits_equal(dev[i], want_w, f"tensor {i}: masked weight")
        want_amax = want_w.float().abs().max()
        got_amax = tab.amax_flat[i].cpu()
        if torch.isnan(want_w.float()).any():
            assert torch.isnan(got_amax), f"tensor {i}: NaN must reach the amax"
        else:
            assert got_amax.item() == want_amax.item(), f"tensor {i}: amax {got_amax.item()} vs {want_amax.item()}"
    # without the statistic: same tensors, nothing written to amax_flat
    dev2 = [h.to(DEV) for h in hosts]
    tab2 = SegmentTable(dev2, outputs=[torch.empty(h.shape, dtype=torch.bool, device=DEV) for h in hosts])
    tab2.amax_flat.fill_(-1.0)
    tab2.mask_2to4_apply()
    assert torch.equal(tab2.amax_flat.cpu(), torch.full((3,), -1.0))
    for a, b in zip(dev2, dev):
        assert_bits_equal(a, b.cpu(), "apply without calibrate")
