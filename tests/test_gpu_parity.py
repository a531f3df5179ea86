"""GPU parity: the HIP path (through the C-ABI) against the CPU oracle and the reference-generated golden
fixtures.  Integer / index / byte results must be bit-exact; floating-point QDQ outputs are compared
bit-exactly too (same fp32 arithmetic), with the few stated exceptions (column sums: fp32 summation order).
Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""

import json
import os

import numpy as np
import pytest
import torch

import _moa_import
from conftest import DT, GOLDEN, assert_bits_equal

pytestmark = pytest.mark.gpu

moa = _moa_import.load()
ops = moa.ops
from oracle import oracle  # noqa: E402  (the checker)

DEV = "cuda:0"
DTYPES = [torch.bfloat16, torch.float16, torch.float32]


def weight_like(shape, dtype, seed, outliers=True):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(*shape, generator=g) * 0.02
    if outliers:
        m = torch.rand(*shape, generator=g) < 0.001
        w = torch.where(m, w * 8, w)
    return w.to(dtype)


@pytest.fixture(scope="module", autouse=True)
def _needs_gpu():
    assert torch.cuda.is_available(), "these tests need a GPU (run with -m 'not gpu' elsewhere)"


# ------------------------------------------------------------------------------------------ golden fixtures
def test_golden_int_fake_quant(golden):
    g = golden("int_fq")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        x, y = g.t(f"{k}_x", dt), g.t(f"{k}_y", dt)
        amax = g.t(f"{k}_amax")
        xd = x.to(DEV)
        if c["mode"] == "group32":
            xv = xd.reshape(-1, 32)
            got = ops.fake_tensor_quant(xv, amax.to(DEV), c["bits"], c["unsigned"], c["narrow"]).reshape(x.shape)
        else:
            got = ops.fake_tensor_quant(xd, amax.to(DEV), c["bits"], c["unsigned"], c["narrow"])
        assert_bits_equal(got, y, f"int_fq {k} {c}")


def test_golden_fp8_fake_quant(golden):
    g = golden("fp8_fq")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        x, y = g.t(f"{k}_x", dt), g.t(f"{k}_y", dt)
        amax = None if c["mode"] == "none" else g.t(f"{k}_amax").to(DEV)
        got = ops.scaled_e4m3(x.to(DEV), amax)
        assert_bits_equal(got, y, f"fp8_fq {k} {c}")


def test_golden_reduce_amax(golden):
    g = golden("amax")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        x, want = g.t(f"{k}_x", dt), g.t(f"{k}_a")
        axis = c["axis"]
        nd = x.dim()
        if axis is None:
            red = None
        else:
            keep = [a % nd for a in (axis if isinstance(axis, list) else [axis])]
            red = [d for d in range(nd) if d not in keep]
        # (a list of kept axes may leave them apart -- round 5: one permuted copy in front of the per-row kernel; rounds
        # 1-4 refused these loudly and the S6 seam handed them back to the reference)
        got = ops.reduce_amax(x.to(DEV), axis=red)
        assert str(got.dtype) == c["out_dtype"], f"{k}: dtype {got.dtype} vs {c['out_dtype']}"
        assert list(got.shape) == c["out_shape"], f"{k}: shape {list(got.shape)} vs {c['out_shape']}"
        assert_bits_equal(got.float(), want, f"amax {k} {c}")


def test_golden_block_quantizer(golden):
    g = golden("tq_block")
    for k, c in g.cases.items():
        if c.get("kind") not in ("dynamic", "static"):
            continue
        dt = DT[c["dtype"]]
        x, y = g.t(f"{k}_x", dt), g.t(f"{k}_y", dt)
        gs = c["g"]
        cols = x.shape[-1]
        pad = (-cols) % gs
        xd = x.to(DEV)
        xp = torch.nn.functional.pad(xd, (0, pad)) if pad else xd
        yq, am = ops.amax_qdq_int_group(xp, gs, num_bits=4, narrow_range=False)
        assert_bits_equal(yq[..., :cols].contiguous(), y, f"tq_block {k} {c}")
        if c["kind"] == "static":
            assert_bits_equal(am.reshape(-1), g.t(f"{k}_amax").reshape(-1), f"tq_block amax {k}")
            # static path: amax from the axis kernel, then QDQ with that amax
            am2 = ops.reduce_amax(xp.reshape(-1, gs), axis=(1,))
            assert_bits_equal(am2.float().reshape(-1), g.t(f"{k}_amax").reshape(-1), f"static amax {k}")
            y2 = ops.fake_tensor_quant(xp.reshape(-1, gs), am2.float(), 4, False, False).reshape(xp.shape)
            assert_bits_equal(y2[..., :cols].contiguous(), y, f"tq_block static {k}")


def test_golden_max_calibrator_running(golden):
    g = golden("tq_block")
    for k, c in g.cases.items():
        if c.get("kind") != "maxcal":
            continue
        want = g.t(f"{k}_a")
        n = 1 if c["axis"] is None else want.numel()
        buf = torch.zeros(n, dtype=torch.float32, device=DEV)
        for b in range(3):
            x = g.t(f"{k}_b{b}", torch.bfloat16).to(DEV)
            red = None if c["axis"] is None else list(range(x.dim() - 1))
            ops.reduce_amax(x, axis=red, out=buf, accumulate=True)
        assert_bits_equal(buf.cpu().reshape(want.shape), want, f"maxcal {k}")


def test_golden_histogram(golden):
    g = golden("hist")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        hist, edges, nbins = None, None, c["num_bins"]
        for b in range(3):
            x = g.t(f"{k}_b{b}", dt).to(DEV)
            want_h = g.t(f"{k}_h{b}")
            xa = x.float().abs()
            if c["skip_zeros"]:
                xa = xa[xa != 0]
            x_max = xa.max().cpu()
            if hist is None:
                counts = ops.hist_abs(x, nbins, float(x_max), c["skip_zeros"])
                hist = counts.cpu().float()
                edges = torch.linspace(0, x_max, nbins + 1)
            else:
                if x_max > edges[-1]:
                    width = edges[1] - edges[0]
                    nbins = int((x_max / width).ceil().item())
                    edges = torch.arange(0, x_max + width, width)
                counts = ops.hist_abs(x, nbins, float(edges[-1]), c["skip_zeros"])
                new = counts.cpu().float()
                new[: hist.numel()] += hist
                hist = new
            assert torch.equal(hist, want_h), f"hist {k} batch {b}: {(hist != want_h).sum().item()} bins differ"


def test_golden_mask_2to4(golden):
    g = golden("mask24")
    for k, c in g.cases.items():
        w = g.t(f"{k}_w", DT[c["dtype"]])
        want = torch.from_numpy(g.raw(f"{k}_m")).bool()
        got = ops.mask_2to4(w.to(DEV)).cpu()
        assert torch.equal(got, want), f"mask {k} {c}: {(got != want).sum().item()} differ"


def test_golden_int4(golden):
    g = golden("int4")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        w = g.t(f"{k}_w", dt)
        if c["kind"] == "qtensor":
            s, q = g.t(f"{k}_s", dt), torch.from_numpy(g.raw(f"{k}_q"))
            got = ops.int4_quantize(w.reshape(-1).to(DEV), s.reshape(-1).to(DEV), c["g"]).cpu()
            assert torch.equal(got, q.reshape(-1)), f"int4 pack {k}"
            deq = ops.int4_dequantize(q.reshape(-1).to(DEV), s.reshape(-1).to(DEV), c["g"]).cpu()
            assert_bits_equal(deq.reshape(w.shape), g.t(f"{k}_d", dt), f"int4 unpack {k}")
        else:
            got = ops.pack_int4_in_uint8(w.to(DEV), g.t(f"{k}_wsf").to(DEV)).cpu()
            assert torch.equal(got, torch.from_numpy(g.raw(f"{k}_p"))), f"export pack {k}"


def test_golden_awq_blocks(golden):
    g = golden("awq")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        w, x = g.t(f"{k}_w", dt).to(DEV), g.t(f"{k}_x", dt).to(DEV)
        ssum, amax = ops.col_abs_stats(x)
        want = g.t(f"{k}_xscale")
        mean = (ssum / x.shape[0]).to(dt).float().cpu()
        # tolerance: one storage-dtype ulp (the reference's own mean differs CPU vs GPU by summation order)
        ulp = (want.abs() * (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10)).clamp_min(1e-30)
        assert ((mean - want).abs() <= ulp).all(), f"act scale {k}"
        _, am_ref = oracle.col_abs_stats(x.cpu())
        assert_bits_equal(amax.cpu(), am_ref, f"col amax {k}")
        for j, _ in enumerate(c["alphas"]):
            s = g.t(f"{k}_s{j}")
            got = ops.awq_scale_qdq(w, s.to(DEV), c["g"], 4)
            assert_bits_equal(got, g.t(f"{k}_wq{j}", dt), f"awq scale+qdq {k} alpha#{j}")
            assert_bits_equal(ops.scale_cols(w, s.to(DEV)), g.t(f"{k}_fold{j}", dt), f"fold {k} alpha#{j}")


def test_mx_golden_vectors():
    cases = json.load(open(os.path.join(GOLDEN, "mx_vectors.json")))
    for c in cases:
        blocks = [c["block_size"]] if c["block_size"] else [8, 16, 32]
        dtypes = [torch.float32] if c["dtype"] else DTYPES
        for bs in blocks:
            for dt in dtypes:
                rep = max(bs // c["in_size"], 1)
                tin = torch.tensor(c["test_in"], dtype=dt).repeat(1, rep)
                tout = torch.tensor(c["test_out"], dtype=dt).repeat(1, rep)
                for sign in (1.0, -1.0):
                    got = ops.fused_amax_convert((tin * sign).to(DEV), bs, c["fmt"]).cpu()
                    assert torch.allclose(got.float(), (tout * sign).float(), rtol=1e-5, atol=c["atol"]), \
                        f"{c['fn']} {c['fmt']} bs={bs} {dt}"


# ------------------------------------------------------------------------------------------ oracle, seeded
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n", [0, 1, 7, 8, 1000, 8192, 8193, 3 * 8192 + 5, 1 << 20])
def test_amax_per_tensor_vs_oracle(dtype, n):
    x = weight_like((n,), dtype, 11 + n) if n else torch.empty(0, dtype=dtype)
    got = ops.reduce_amax(x.to(DEV))
    want = oracle.reduce_amax(x) if n else torch.tensor(0.0)
    assert_bits_equal(got.float().cpu(), want, f"amax n={n}")
    if n > 16:
        # unaligned base pointer (slice of a bigger buffer)
        big = torch.zeros(n + 3, dtype=dtype, device=DEV)
        big[3:] = x.to(DEV)
        assert_bits_equal(ops.reduce_amax(big[3:]).float().cpu(), want, "unaligned amax")


@pytest.mark.parametrize("dtype", DTYPES)
def test_amax_nan_inf_propagation(dtype):
    x = weight_like((4, 4096), dtype, 5)
    x[2, 17] = float("nan")
    assert torch.isnan(ops.reduce_amax(x.to(DEV))).item()
    a = ops.reduce_amax(x.to(DEV), axis=(1,))
    assert torch.isnan(a[2]).item() and not torch.isnan(a[0]).item()
    x[2, 17] = float("-inf")
    assert torch.isinf(ops.reduce_amax(x.to(DEV))).item()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(1, 8, 64), (1, 300, 4096), (5, 64, 24), (1, 4, 1 << 18), (257, 4096, 1),
                                   (33, 520, 1), (7, 13, 5), (3, 100, 3)])
def test_amax_axis_vs_oracle(dtype, shape):
    outer, axis, inner = shape
    x = weight_like(shape, dtype, 21 + axis)
    got = ops.reduce_amax(x.to(DEV), axis=(0, 2))
    want = oracle.reduce_amax_axis(x, outer, axis, inner)
    assert_bits_equal(got.float().cpu().reshape(-1), want, f"amax_axis {shape}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,reduce", [((3, 4, 5, 16, 2), (3,)), ((2, 6, 3, 8), (1, 3)), ((4, 2, 8, 3, 2), (1, 3)),
                                          ((5, 7, 9), (0, 2)), ((2, 3, 4, 5), (0, 2))])
def test_amax_over_dims_that_leave_the_kept_dims_apart(dtype, shape, reduce):
    """reduce_amax with kept dims that are not one adjacent block (the reference's N-D block views, core_utils.py:146-183;
    the S6 seam used to hand these back to the reference): one permuted copy, then the per-row kernel.  A maximum is exact:
    bit-equal to torch's amax, keepdims on and off; (0, 2) of a 3-D tensor is the adjacent case for comparison."""
    x = weight_like(shape, dtype, 77 + len(shape))
    want = x.float().abs().amax(dim=reduce, keepdim=True).to(dtype)
    got = ops.reduce_amax(x.to(DEV), axis=reduce)
    assert got.shape == want.shape and got.dtype == dtype
    assert_bits_equal(got.float().cpu().reshape(-1), want.float().reshape(-1), f"amax {shape} over {reduce}")
    flat = ops.reduce_amax(x.to(DEV), axis=reduce, keepdims=False)
    assert list(flat.shape) == [s for d, s in enumerate(shape) if d not in reduce]
    assert torch.equal(flat.float().cpu().reshape(-1), want.float().reshape(-1))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("bits,unsigned,narrow", [(8, False, True), (4, False, False), (3, False, True),
                                                  (8, True, False), (11, False, False)])
def test_fake_quant_int_vs_oracle(dtype, bits, unsigned, narrow):
    x = weight_like((129, 520), dtype, 31 + bits)
    x[0, :6] = torch.tensor([0.0, -0.0, float("inf"), float("-inf"), float("nan"), 1e-30]).to(dtype)
    if unsigned:
        x = x.abs()
    # scalar
    amax = x[torch.isfinite(x)].abs().max().float().reshape(1)
    got = ops.fake_tensor_quant(x.to(DEV), amax.to(DEV), bits, unsigned, narrow)
    assert_bits_equal(got, oracle.fake_quant_int(x, amax, bits, unsigned, narrow), "scalar")
    # per row (axis 0) and per column (axis 1)
    xf = torch.nan_to_num(x.float(), nan=0.0, posinf=0.0, neginf=0.0)
    am0 = xf.abs().amax(dim=1, keepdim=True)
    got = ops.fake_tensor_quant(x.to(DEV), am0.to(DEV), bits, unsigned, narrow)
    want = oracle.fake_quant_int(x, am0, bits, unsigned, narrow, axis_size=x.shape[0], inner=x.shape[1], per_axis=True)
    assert_bits_equal(got, want, "axis0")
    am1 = xf.abs().amax(dim=0, keepdim=True)
    got = ops.fake_tensor_quant(x.to(DEV), am1.to(DEV), bits, unsigned, narrow)
    want = oracle.fake_quant_int(x, am1, bits, unsigned, narrow, axis_size=x.shape[1], inner=1, per_axis=True)
    assert_bits_equal(got, want, "axis1")
    # zero / tiny amax
    for a in (0.0, 2.0 ** -24, 2.0 ** -23):
        am = torch.tensor([a])
        assert_bits_equal(ops.fake_tensor_quant(x.to(DEV), am.to(DEV), bits, unsigned, narrow),
                          oracle.fake_quant_int(x, am, bits, unsigned, narrow), f"tiny amax {a}")
    # in place
    xd = x.to(DEV).clone()
    ops.fake_tensor_quant(xd, amax.to(DEV), bits, unsigned, narrow, inplace=True)
    assert_bits_equal(xd, oracle.fake_quant_int(x, amax, bits, unsigned, narrow), "inplace")


@pytest.mark.parametrize("dtype", DTYPES)
def test_fp8_vs_oracle_dense_sweep(dtype):
    """Every bf16 / f16 bit pattern (and 1M random f32) through the E4M3 QDQ, several amax values."""
    if dtype == torch.float32:
        x = torch.randn(1 << 20, generator=torch.Generator().manual_seed(3)) * torch.logspace(-6, 4, 1 << 20)
    else:
        x = torch.arange(0, 1 << 16, dtype=torch.int32).to(torch.int16).view(dtype)
    for a in (None, 1.0, 3.0, 448.0, 0.017, 1e-8, 57344.0):
        amax = None if a is None else torch.tensor([a])
        got = ops.scaled_e4m3(x.to(DEV), None if amax is None else amax.to(DEV))
        want = oracle.fake_quant_e4m3(x, amax)
        assert_bits_equal(got, want, f"fp8 {dtype} amax={a}")
    xw = weight_like((64, 264), dtype, 77)
    am = xw.float().abs().amax(dim=1, keepdim=True)
    got = ops.scaled_e4m3(xw.to(DEV), am.to(DEV))
    assert_bits_equal(got, oracle.fake_quant_e4m3(xw, am, axis_size=64, inner=264, per_axis=True), "fp8 axis0")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("g", [16, 32, 128, 256])
@pytest.mark.parametrize("bits", [4, 8])
def test_fused_group_vs_oracle(dtype, g, bits):
    for rows, cols in [(3, g), (64, 4 * g), (96, 8192 // g * g + g)]:
        x = weight_like((rows, cols), dtype, 41 + g + rows)
        x[0, :2] = torch.tensor([float("nan"), float("inf")]).to(dtype)
        if rows > 3:
            x[1, :g] = 0  # an all-zero group
        y, am = ops.amax_qdq_int_group(x.to(DEV), g, num_bits=bits, narrow_range=False)
        wy, wam = oracle.amax_qdq_int_group(x, g, num_bits=bits, narrow_range=False)
        assert_bits_equal(am.cpu(), wam, f"group amax g={g} {rows}x{cols}")
        assert_bits_equal(y, wy, f"group qdq g={g} {rows}x{cols}")


@pytest.mark.parametrize("bits", [4, 8, 16])
def test_shared_division_exact(bits):
    """The group / per-tensor INT QDQ kernels divide by a group-shared scale with a hand-rolled
    reciprocal-refinement sequence; it must equal IEEE division bit for bit.  32768 groups with log-uniform
    magnitudes over 14 decades (plus scales outside the fast window) against the oracle's `/`."""
    g_ = torch.Generator().manual_seed(bits)
    n_groups, g = 32768, 128
    mag = torch.exp(torch.empty(n_groups, 1).uniform_(-16.0, 16.0, generator=g_) * 2.302585)  # 1e-16..1e16
    x = (torch.randn(n_groups, g, generator=g_) * mag).float()
    x[:64] *= 1e-20
    x[64:128] *= 1e18
    y, am = ops.amax_qdq_int_group(x.to(DEV), g, num_bits=bits, narrow_range=False)
    wy, wam = oracle.amax_qdq_int_group(x, g, num_bits=bits, narrow_range=False)
    assert_bits_equal(am.cpu(), wam, "amax")
    assert_bits_equal(y, wy, f"shared division, bits={bits}")
    for a in (1e-12, 3e-5, 0.0131, 1.0, 7.77, 1234.5, 3.3e9, 1e25):
        xs = (torch.randn(1 << 16, generator=g_) * a).float()
        am1 = xs.abs().max().reshape(1)
        assert_bits_equal(ops.fake_tensor_quant(xs.to(DEV), am1.to(DEV), bits, False, True),
                          oracle.fake_quant_int(xs, am1, bits, False, True), f"scalar amax {a}")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("fmt", ["E2M1", "E3M2", "E2M3", "E4M3", "E5M2", "INT8", "E1M2", "E0M3", "E3M0"])
def test_mx_vs_oracle(dtype, fmt):
    for shape, bs in [((16, 64), 32), ((8, 32), 8), ((5, 40), 16), ((3, 7, 50), 32), ((130, 4096), 32)]:
        g_ = torch.Generator().manual_seed(hash((fmt, bs)) % 1000)
        x = (torch.randn(*shape, generator=g_) * torch.exp(2 * torch.randn(*shape, generator=g_))).to(dtype)
        x.view(-1)[:5] = torch.tensor([0.0, -0.0, float("inf"), float("nan"), 6.0]).to(dtype)
        got = ops.fused_amax_convert(x.to(DEV), bs, fmt)
        want = oracle.mx_fused_amax_convert(x, bs, fmt)
        assert_bits_equal(got, want, f"mx {fmt} {shape} bs={bs}")


@pytest.mark.parametrize("dtype", DTYPES)
def test_mask_vs_oracle(dtype):
    for shape in [(8, 16), (64, 4096), (33, 20), (1, 4)]:
        w = weight_like(shape, dtype, 51 + shape[1])
        w2 = torch.randint(-2, 3, shape, generator=torch.Generator().manual_seed(1)).to(dtype)
        for t in (w, w2):
            got = ops.mask_2to4(t.to(DEV)).cpu()
            assert torch.equal(got, oracle.mask_2to4(t)), f"mask {shape}"
            assert (got.view(-1, 4).sum(1) == 2).all()


@pytest.mark.parametrize("dtype", DTYPES)
def test_int4_pack_vs_oracle(dtype):
    for shape, g in [((64, 256), 128), ((7, 64), 32), ((16, 4096), 128)]:
        w = weight_like(shape, dtype, 61 + g)
        scales = (7 / w.reshape(-1, g).abs().amax(dim=1, keepdim=True)).to(dtype)
        for rounding in (0, 1):
            got = ops.int4_quantize(w.reshape(-1).to(DEV), scales.reshape(-1).to(DEV), g, rounding).cpu()
            assert torch.equal(got, oracle.int4_pack(w.reshape(-1), scales.reshape(-1), g, rounding)), \
                f"int4 pack {shape} rounding={rounding}"
        q = oracle.int4_pack(w.reshape(-1), scales.reshape(-1), g, 0)
        deq = ops.int4_dequantize(q.to(DEV), scales.reshape(-1).to(DEV), g).cpu()
        assert_bits_equal(deq, oracle.int4_unpack(q, scales.reshape(-1), g), f"int4 unpack {shape}")
        if shape[0] % 2 == 0:
            wsf = w.reshape(shape[0], -1, g).abs().amax(-1).float() / 7.0
            got = ops.pack_int4_in_uint8(w.to(DEV), wsf.to(DEV)).cpu()
            assert torch.equal(got, oracle.int4_pack_export(w, wsf)), f"export pack {shape}"


@pytest.mark.parametrize("dtype", DTYPES)
def test_int4_packers_ties_edges_and_odd_scales(dtype):
    """The INT4 packers round through the low mantissa bits of `clamp(t) + 1.5 * 2^23` and divide by a group-shared
    reciprocal where that is provably exact.  Products and quotients on every .5 tie from -9.5 to 8.5, values far outside
    [-8, 7], infinities, numerators above 2^16 (IEEE fallback of the export packer), scales that are tiny, huge (outside the
    shared division's window), negative -- against the oracle's literal restatement, both rounding modes."""
    g = 32
    ties = torch.arange(-19, 18, dtype=torch.float32) / 2.0                      # -9.5 ... 8.5
    extra = torch.tensor([float("inf"), -float("inf"), 1e5, -1e5, 3e4, 7.49, -8.51, 0.0, -0.0, 1e-30])
    base = torch.cat([ties, extra, torch.zeros(64 - ties.numel() - extra.numel())])  # two groups of 32
    rows = []
    scs = [1.0, 0.5, 0.25, 3.0, 1e-3, 2.0 ** -70 if dtype != torch.float16 else 2.0 ** -12, -1.0, 1e3]
    for sc in scs:
        rows.append(base / sc)                                                    # x * scale hits the ties exactly (powers of two)
    x = torch.stack(rows).to(dtype)                                               # [8, 64]
    scales = torch.tensor(scs).repeat_interleave(64 // g).to(dtype)               # one per group of 32
    for rounding in (0, 1):
        got = ops.int4_quantize(x.reshape(-1).to(DEV), scales.to(DEV), g, rounding).cpu()
        assert torch.equal(got, oracle.int4_pack(x.reshape(-1), scales, g, rounding)), f"int4 pack rounding={rounding}"
    # export packer: fp32 scaling factors, quotient = w / wsf
    wsf = (1.0 / torch.tensor(scs)).reshape(8, 1).repeat(1, 64 // g).float().contiguous()
    got = ops.pack_int4_in_uint8(x.to(DEV), wsf.to(DEV)).cpu()
    assert torch.equal(got, oracle.int4_pack_export(x, wsf)), "export pack on ties"
    # unpack: every nibble under the same scales (shared reciprocal inside the window, IEEE outside)
    q = torch.arange(256, dtype=torch.uint8).repeat(2)                           # 1024 elements = 32 groups of 32
    sc_u = torch.tensor(scs + [7.0 / 0.0131, 2.0 ** 70 if dtype != torch.float16 else 6e4]).repeat(4)[:32].to(dtype)
    deq = ops.int4_dequantize(q.to(DEV), sc_u.to(DEV), g).cpu()
    assert_bits_equal(deq, oracle.int4_unpack(q, sc_u, g), "int4 unpack, every nibble")


@pytest.mark.parametrize("dtype", DTYPES)
def test_col_stats_and_scale_vs_oracle(dtype):
    g_ = torch.Generator().manual_seed(71)
    x = (torch.randn(300, 4096, generator=g_) * torch.exp(torch.randn(4096, generator=g_))).to(dtype)
    ssum, amax = ops.col_abs_stats(x.to(DEV))
    s64, am = oracle.col_abs_stats(x)
    assert_bits_equal(amax.cpu(), am, "col amax")
    rel = ((ssum.cpu().double() - s64).abs() / s64.clamp_min(1e-30)).max().item()
    assert rel < 1e-5, f"col sum rel err {rel}"  # fp32 two-stage sum vs fp64: tolerance 1e-5 relative
    # accumulate over two batches == one pass over the concatenation
    s2, a2 = ops.col_abs_stats(x[:100].to(DEV))
    ops.col_abs_stats(x[100:].to(DEV), sum_out=s2, amax_out=a2, accumulate=True)
    assert_bits_equal(a2.cpu(), am, "col amax accumulate")
    assert ((s2.cpu().double() - s64).abs() / s64.clamp_min(1e-30)).max().item() < 1e-5
    w = weight_like((64, 4096), dtype, 72)
    s = torch.exp(torch.randn(4096, generator=g_) * 0.5)
    assert_bits_equal(ops.scale_cols(w.to(DEV), s.to(DEV)), oracle.scale_cols(w, s), "scale_cols")
    assert_bits_equal(ops.awq_scale_qdq(w.to(DEV), s.to(DEV), 128, 4), oracle.awq_scale_qdq(w, s.to(dtype), 128, 4),
                      "awq_scale_qdq")


# ------------------------------------------------------------------------------------------ multi-tensor
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_multi_tensor_matches_single(dtype):
    shapes = [(64, 256), (3, 8192), (17, 128), (1, 128), (100, 4096), (1024, 1024)]
    ws = [weight_like(s, dtype, 81 + i).to(DEV) for i, s in enumerate(shapes)]
    tab = moa.multi_tensor.SegmentTable(ws)
    am = tab.calibrate_amax().cpu()
    for i, w in enumerate(ws):
        assert_bits_equal(am[i].reshape(()), oracle.reduce_amax(w.cpu()).reshape(()), f"mt amax {i}")
    tab.amax_flat.fill_(7.0)  # the atomic single-launch form gives the same answer (and resets the table itself)
    assert_bits_equal(tab.calibrate_amax(atomic=True).cpu(), am, "mt amax atomic form")
    ws[2].view(-1)[5] = float("nan")  # NaN poisons that tensor's amax in both forms (torch.max semantics)
    a_ws, a_at = tab.calibrate_amax().cpu().clone(), tab.calibrate_amax(atomic=True).cpu().clone()
    assert torch.isnan(a_ws[2]) and torch.isnan(a_at[2]) and not torch.isnan(a_ws[[0, 1, 3, 4, 5]]).any()
    ws[2].view(-1)[5] = 0.0
    am = tab.calibrate_amax().cpu()
    outs = tab.fake_quant_e4m3()
    for i, w in enumerate(ws):
        assert_bits_equal(outs[i], oracle.fake_quant_e4m3(w.cpu(), am[i:i + 1]), f"mt fp8 {i}")
    outs = tab.fake_quant_int(8, False, True)
    for i, w in enumerate(ws):
        assert_bits_equal(outs[i], oracle.fake_quant_int(w.cpu(), am[i:i + 1], 8, False, True), f"mt int8 {i}")
    tabg = moa.multi_tensor.SegmentTable(ws, group_size=128)
    outs = tabg.amax_qdq_int_group(4, False, False)
    for i, w in enumerate(ws):
        wy, wam = oracle.amax_qdq_int_group(w.cpu(), 128, num_bits=4, narrow_range=False)
        assert_bits_equal(outs[i], wy, f"mt group qdq {i}")
        assert_bits_equal(tabg.amax[i].cpu(), wam, f"mt group amax {i}")


# ------------------------------------------------------------------------------------------ full-size properties
def test_full_size_properties_llama70b_tensor():
    """BASELINE-size tensor (28672 x 8192 bf16, Llama-3-70B gate/up): size-independent properties."""
    torch.manual_seed(1234)
    w = (torch.randn(28672, 8192, device=DEV) * 0.02).to(torch.bfloat16)
    y, am = ops.amax_qdq_int_group(w, 128, num_bits=4, narrow_range=False)
    # (1) group amax == torch's own reduction
    assert torch.equal(am, w.view(-1, 128).abs().amax(dim=1).float())
    # (2) idempotence: QDQ of the QDQ output with the same amax is a fixed point
    y2 = ops.fake_tensor_quant(y.view(-1, 128), am.view(-1, 1), 4, False, False).view_as(y)
    assert torch.equal(y2, y)
    # (3) at most 16 distinct values per group, error bounded by half a step
    step = (am / 7).view(-1, 1)
    err = (y.view(-1, 128).float() - w.view(-1, 128).float()).abs()
    assert (err <= step * 0.5 + am.view(-1, 1) * 2.0 ** -8).all()  # half a step + bf16 output rounding
    # (4) spot-check a slice bit-exactly against the oracle
    sl = w[1000:1016].cpu()
    wy, _ = oracle.amax_qdq_int_group(sl, 128, num_bits=4, narrow_range=False)
    assert_bits_equal(y[1000:1016], wy, "slice vs oracle")
    # (5) per-tensor amax == max of group amax; FP8 QDQ idempotent
    a = ops.reduce_amax(w)
    assert a.float().item() == am.max().item()
    f1 = ops.scaled_e4m3(w, a.float())
    assert torch.equal(ops.scaled_e4m3(f1, a.float()), f1)
    # (6) 2:4 mask keeps exactly 2 of 4 and the kept magnitude sum is the best achievable (top-2 sum)
    m = ops.mask_2to4(w)
    assert (m.view(-1, 4).sum(1) == 2).all()
    wa = w.view(-1, 4).abs().float()
    kept = (wa * m.view(-1, 4)).sum(1)
    assert torch.equal(kept, wa.topk(2, dim=1).values.sum(1))


def test_tensor_beyond_int32_elements():
    """A single tensor of 2^31 + 2^19 elements (4.3 GB bf16): every index the kernels form must be 64-bit.  Slices at
    the start, around element 2^31 and at the very end are compared bit-exactly with the oracle."""
    rows, cols = 32768 + 8, 65536
    assert rows * cols > 2 ** 31
    torch.manual_seed(77)
    w = torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV)
    for r0 in range(0, rows, 4096):  # generate in slabs: randn of 2^31 elements at once needs 8.6 GB of fp32
        r1 = min(rows, r0 + 4096)
        w[r0:r1] = (torch.randn(r1 - r0, cols, device=DEV) * 0.02).to(torch.bfloat16)
    w[rows - 1, cols - 3] = 3.0  # the per-tensor abs-max sits in the last packet of the tensor
    w[5, 7] = -2.5
    probes = [(0, 2), (32766, 32770), (rows - 2, rows)]  # row 32768 starts at element 2^31

    a = ops.reduce_amax(w)
    assert a.float().item() == 3.0
    am_rows = ops.reduce_amax(w, axis=[1])  # per-channel (axis=0 kept)
    assert am_rows.shape == (rows, 1) and am_rows[rows - 1].float().item() == 3.0 and am_rows[5].float().item() == 2.5

    y = ops.scaled_e4m3(w, a.float())
    yi = ops.fake_tensor_quant(w, a.float(), 8, False, True)
    yg, amg = ops.amax_qdq_int_group(w, 128, num_bits=4, narrow_range=False)
    yc = ops.fake_tensor_quant_with_axis(w, am_rows.reshape(-1).float(), 0, 8, False, True)
    ymx = ops.fused_amax_convert(w, 32, "E2M1")
    m = ops.mask_2to4(w)
    assert amg.numel() == rows * cols // 128
    for r0, r1 in probes:
        sl = w[r0:r1].cpu()
        assert_bits_equal(y[r0:r1], oracle.fake_quant_e4m3(sl, a.float().cpu().reshape(1)), f"fp8 rows {r0}")
        assert_bits_equal(yi[r0:r1], oracle.fake_quant_int(sl, a.float().cpu().reshape(1), 8, False, True), f"int8 rows {r0}")
        gy, gam = oracle.amax_qdq_int_group(sl, 128, num_bits=4, narrow_range=False)
        assert_bits_equal(yg[r0:r1], gy, f"int4 g128 rows {r0}")
        assert_bits_equal(amg.view(rows, -1)[r0:r1].reshape(-1).cpu(), gam.reshape(-1), f"group amax rows {r0}")
        cy = oracle.fake_quant_int(sl, am_rows[r0:r1].reshape(-1).float().cpu(), 8, False, True, axis_size=r1 - r0,
                                   inner=cols, per_axis=True)
        assert_bits_equal(yc[r0:r1], cy, f"per-channel rows {r0}")
        assert_bits_equal(ymx[r0:r1], oracle.mx_fused_amax_convert(sl, 32, "E2M1"), f"mxfp4 rows {r0}")
        assert torch.equal(m[r0:r1].cpu().bool(), oracle.mask_2to4(sl).bool()), f"mask rows {r0}"
    del y, yi, yg, yc, ymx, m
    # multi-tensor table holding the big tensor behind a small one: chunk prefix sums past 2^31 elements
    small = (torch.randn(3, 1000, device=DEV) * 0.02).to(torch.bfloat16)
    tab = moa.multi_tensor.SegmentTable([small, w])
    am = tab.calibrate_amax().cpu()
    assert am[1].item() == 3.0 and am[0].item() == small.abs().max().float().item()
    assert torch.equal(tab.calibrate_amax(atomic=True).cpu(), am)
    outs = tab.fake_quant_e4m3()
    for r0, r1 in probes:
        assert_bits_equal(outs[1][r0:r1], oracle.fake_quant_e4m3(w[r0:r1].cpu(), am[1:2]), f"mt fp8 rows {r0}")


def test_packers_beyond_int32_elements():
    """The real-quant / export / statistics entries on the same 2^31 + 2^19 element tensor."""
    rows, cols, g = 32768 + 8, 65536, 128
    torch.manual_seed(78)
    w = torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV)
    for r0 in range(0, rows, 4096):
        r1 = min(rows, r0 + 4096)
        w[r0:r1] = (torch.randn(r1 - r0, cols, device=DEV) * 0.02).to(torch.bfloat16)
    w[rows - 1, cols - 3] = 1.5
    probes = [(0, 2), (32766, 32770), (rows - 2, rows)]
    am = ops.reduce_amax(w.view(-1, g), axis=[1]).float()          # [n/g, 1] block amax
    scales = (am / 7.0).to(torch.bfloat16)                          # INT4QTensor: scales = amax / 7
    q4 = ops.int4_quantize(w.view(-1), scales, g)
    d4 = ops.int4_dequantize(q4, scales, g)
    wsf = (am / 7.0).view(rows, cols // g)
    qe = ops.pack_int4_in_uint8(w, wsf)
    a_t = ops.reduce_amax(w).float()
    q8 = ops.fp8_quantize(w, (a_t / 448.0).reshape(1))
    d8 = ops.fp8_dequantize(q8, (a_t / 448.0).reshape(1), torch.bfloat16)
    qm, em = ops.mxfp4_quantize(w, 32)
    dm = ops.mxfp4_dequantize(qm, em, torch.bfloat16, 32)
    s_col = torch.exp(torch.randn(cols, device=DEV) * 0.3)
    ws = ops.scale_cols(w, s_col)
    for r0, r1 in probes:
        sl, n_sl = w[r0:r1].cpu(), (r1 - r0) * cols
        e0 = r0 * cols
        sc = scales.view(-1)[e0 // g:(e0 + n_sl) // g].cpu()
        assert torch.equal(q4[e0 // 2:(e0 + n_sl) // 2].cpu(), oracle.int4_pack(sl.reshape(-1), sc, g)), f"int4 pack {r0}"
        assert_bits_equal(d4[e0:e0 + n_sl], oracle.int4_unpack(q4[e0 // 2:(e0 + n_sl) // 2].cpu(), sc, g), f"int4 unpack {r0}")
        if r0 % 2 == 0:  # export packer pairs rows (2i, 2i + 1)
            assert torch.equal(qe[r0 // 2:r1 // 2].cpu(), oracle.int4_pack_export(sl, wsf[r0:r1].cpu())), f"export {r0}"
        s8 = (a_t / 448.0).reshape(1).cpu()
        assert torch.equal(q8[r0:r1].cpu().view(torch.uint8), oracle.fp8_pack(sl, s8)), f"fp8 pack {r0}"
        assert_bits_equal(d8[r0:r1], oracle.fp8_unpack(q8[r0:r1].cpu(), s8, torch.bfloat16), f"fp8 unpack {r0}")
        op, oe = oracle.mxfp4_pack(sl, 32)
        assert torch.equal(qm[r0:r1].cpu(), op) and torch.equal(em.view(rows, -1)[r0:r1].reshape(-1, 1).cpu(), oe), f"mxfp4 {r0}"
        assert_bits_equal(dm[r0:r1], oracle.mxfp4_unpack(op, oe, torch.bfloat16, 32), f"mxfp4 unpack {r0}")
        assert_bits_equal(ws[r0:r1], oracle.scale_cols(sl, s_col.cpu()), f"scale_cols {r0}")
    del q4, d4, qe, q8, d8, qm, em, dm, ws
    # histogram and column statistics: additive over a split at a row boundary (each half < 2^31 elements)
    half = 20000
    mx = float(ops.reduce_amax(w).float())
    h_all = ops.hist_abs(w, 2048, mx)
    h_sum = ops.hist_abs(w[:half], 2048, mx)
    ops.hist_abs(w[half:], 2048, mx, counts=h_sum)
    assert int(h_all.sum()) == rows * cols and torch.equal(h_all, h_sum)
    csum, camax = ops.col_abs_stats(w)
    c2, a2 = ops.col_abs_stats(w[:half])
    ops.col_abs_stats(w[half:], sum_out=c2, amax_out=a2, accumulate=True)
    assert torch.equal(camax, a2) and torch.equal(camax, w.abs().amax(0).float())
    assert ((csum - c2).abs() / c2.clamp_min(1e-30)).max().item() < 1e-5
    assert torch.equal(ops.mask_2to4(w[rows - 4:]).cpu(), oracle.mask_2to4(w[rows - 4:].cpu()))


@pytest.mark.parametrize("dn", ["bf16", "f16", "f32"])
def test_multi_tensor_mx_equals_per_tensor(dn):
    """moq_mt_mx_fused_amax_convert over a segment table == moq_mx_fused_amax_convert tensor by tensor == oracle."""
    dt = DT[dn]
    gen = torch.Generator().manual_seed(9)
    shapes = [(64, 256), (8192 + 32, 32), (3, 96), (128, 1024)]
    ws = [(torch.randn(*s, generator=gen) * torch.exp(torch.randn(s[0], 1, generator=gen))).to(dt).to(DEV) for s in shapes]
    for fmt, block in [("E2M1", 32), ("E4M3", 32), ("E2M1", 16), ("E3M2", 32)]:
        tab = moa.multi_tensor.SegmentTable(ws)
        outs = tab.mx_fused_amax_convert(block, fmt)
        for w, y in zip(ws, outs):
            assert_bits_equal(y.cpu(), oracle.mx_fused_amax_convert(w.cpu(), block, fmt), f"mt mx {dn} {fmt} b{block} {tuple(w.shape)}")
            assert_bits_equal(y, ops.fused_amax_convert(w, block, fmt), f"mt vs single {dn} {fmt}")
    # in place (outputs = inputs) and the `out=` form of the single-tensor op
    w0 = ws[0].clone()
    want = ops.fused_amax_convert(w0, 32, "E2M1")
    moa.multi_tensor.SegmentTable([w0], outputs=[w0]).mx_fused_amax_convert(32, "E2M1")
    assert_bits_equal(w0, want, "in-place mt mx")
    w1 = ws[3].clone()
    ops.fused_amax_convert(w1, 32, "E2M1", out=w1)
    assert_bits_equal(w1, ops.fused_amax_convert(ws[3], 32, "E2M1"), "in-place single mx")


@pytest.mark.parametrize("dn", ["bf16", "f16", "f32"])
def test_fold_composed_with_mx_in_one_launch_equals_fold_then_qdq(dn):
    """moq_mt_fold_mx_fused / moq_mt_fold_mxfp4_pack (BASELINE configs[4]: SmoothQuant's column fold composed with MXFP4 g32):
    one launch over a segment table == the oracle's separate steps -- scale_cols (model_calib.py:1208-1216: fp32 product, one
    rounding to the weight dtype), then the MX QDQ / MXFP4QTensor.quantize of what the fold wrote -- bit for bit, and == this
    library's own two-pass form.  Rows shorter and longer than a chunk (the column index wraps by remainder / by
    compare-and-subtract), a tensor without a fold, a chunk that starts mid-row, scales that over- and underflow the dtype."""
    dt = DT[dn]
    gen = torch.Generator().manual_seed(31)
    # (8, 4096), (16, 6144), (24, 2048): shapes that TILE (rows % kPackets == 0, cols % (256 * kVec) == 0: one scale read per
    # thread and tile; (24, 2048) tiles for the 16-bit types only) -- several bands and several tiles per band; the others walk
    # linearly
    shapes = [(64, 256), (5, 8192 + 64), (3, 96), (130, 1024), (2, 28672), (40, 32), (8, 4096), (16, 6144), (24, 2048)]
    ws = [(torch.randn(*s, generator=gen) * torch.exp(torch.randn(s[0], 1, generator=gen))).to(dt) for s in shapes]
    scales = [torch.exp(torch.randn(s[1], generator=gen) * 1.5) for s in shapes]
    scales[2] = None  # no fold for this tensor
    ws[6][5, 4095], ws[6][2, 17] = float("inf"), float("nan")
    scales[3][7], scales[3][8] = 1e30, 1e-30
    ws[0][3, 5], ws[0][9, 64] = float("nan"), float("inf")
    dws = [w.to(DEV) for w in ws]
    dsc = [None if s is None else s.to(DEV) for s in scales]
    folded = [w if s is None else oracle.scale_cols(w, s) for w, s in zip(ws, scales)]
    for fmt, block in [("E2M1", 32), ("E4M3", 32), ("E2M1", 16)]:
        outs = moa.multi_tensor.SegmentTable(dws).fold_mx_fused(dsc, block, fmt)
        for w, f, y, s in zip(dws, folded, outs, dsc):
            assert_bits_equal(y.cpu(), oracle.mx_fused_amax_convert(f, block, fmt), f"fold+mx {dn} {fmt} b{block} {tuple(w.shape)}")
            two_pass = ops.fused_amax_convert(w if s is None else ops.scale_cols(w, s), block, fmt)
            assert_bits_equal(y, two_pass, f"fused vs two passes {dn} {fmt} {tuple(w.shape)}")
    # in place, with the side table kept across launches (bench.py's step)
    mine = [w.clone() for w in dws]
    tab = moa.multi_tensor.SegmentTable(mine, outputs=mine)
    side = tab.fold_side(dsc, 32)
    tab.fold_mx_fused(side=side)
    for y, f in zip(mine, folded):
        assert_bits_equal(y.cpu(), oracle.mx_fused_amax_convert(f, 32, "E2M1"), "in place")
    # the checkpoint form: nibbles + E8M0 of the folded weight
    packed = [torch.empty(*w.shape[:-1], w.shape[-1] // 2, dtype=torch.uint8, device=DEV) for w in dws]
    outs, e8 = moa.multi_tensor.SegmentTable(dws, outputs=packed).fold_mxfp4_pack(dsc, 32)
    for w, f, q, e in zip(dws, folded, outs, e8):
        q2, e2 = ops.mxfp4_quantize(f.to(DEV), 32)  # this library's packer on what the separate fold wrote
        assert torch.equal(q, q2) and torch.equal(e, e2), f"fold+pack vs fold, then pack {dn} {tuple(w.shape)}"
        if torch.isfinite(f.float()).all():  # (a block with NaN / inf has no defined MXFP4 exponent in the reference either:
            wq, we = oracle.mxfp4_pack(f, 32)  # uint8(NaN), qtensor/mxfp4_tensor.py:72-81)
            assert torch.equal(q.cpu(), wq) and torch.equal(e.cpu(), we), f"fold+pack {dn} {tuple(w.shape)}"


def test_multi_tensor_mask_equals_per_tensor():
    gen = torch.Generator().manual_seed(2)
    ws = [torch.randn(*s, generator=gen).to(torch.bfloat16).to(DEV) for s in [(64, 256), (8192 + 8, 16), (3, 96), (100, 1024)]]
    masks = [torch.empty(w.shape, dtype=torch.bool, device=DEV) for w in ws]
    moa.multi_tensor.SegmentTable(ws, outputs=masks).mask_2to4()
    for w, m in zip(ws, masks):
        assert torch.equal(m.cpu(), oracle.mask_2to4(w.cpu()))


@pytest.mark.parametrize("fmt", ["E2M1", "E1M2", "E0M3", "E3M0", "E3M2", "E2M3", "E4M3", "E5M2", "INT8"])
def test_convert_to_exmy_vs_oracle(fmt):
    """cuda_ext_mx.convert_to_exmy: a dense sweep (every tie of the small formats included), negatives, zeros,
    infinities, NaN, values beyond the format maximum."""
    grid = torch.arange(-40000, 40001, dtype=torch.float32) / 64.0            # multiples of 1/64: all ties of the tables
    gen = torch.Generator().manual_seed(4)
    rnd = torch.randn(20000, generator=gen) * torch.exp(torch.randn(20000, generator=gen) * 4)
    special = torch.tensor([0.0, -0.0, float("inf"), -float("inf"), float("nan"), 1e30, -1e30, 1e-30, 448.0, 464.0, 465.0,
                            57344.0, 61440.0, 127.5, -127.5, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -16])
    x = torch.cat([grid, rnd, special])
    got = ops.convert_to_exmy(x.to(DEV), fmt).cpu()
    want = oracle.mx_convert(x, fmt)
    assert_bits_equal(got, want, f"convert_to_exmy {fmt}")
    assert ops.convert_to_exmy(2.4, "E2M1") == 2.0 and ops.convert_to_exmy(-2.5, "E2M1") == -2.0  # tie -> even code
    assert isinstance(ops.convert_to_exmy(0.3, fmt), float)


_FP4_LITERALS = [([0, 0.5, 1, 1.5, 2, 3, 4, 6], [0, 0.5, 1, 1.5, 2, 3, 4, 6]),
                 ([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5, 6], [0.0, 1, 1, 2, 2, 4, 4, 6]),
                 ([0.15, 0.65, 1.15, 1.65, 2.4, 3.4, 4.9, 6], [0.0, 0.5, 1, 1.5, 2, 3, 4, 6]),
                 ([0.35, 0.85, 1.35, 1.85, 2.6, 3.6, 5.1, 6], [0.5, 1, 1.5, 2, 3, 4, 6, 6])]


@pytest.mark.parametrize("dn", ["f32", "bf16", "f16"])
def test_two_level_scales_from_the_table_equal_the_fp64_steps(dn):
    """With a tensor-wide amax the aligned kernel takes the block scale from a per-workgroup table of the eight
    (4-bit mantissa) quotients instead of two fp64 divisions per block.  Block magnitudes from 1e-30 to 1e30 times the
    tensor-wide value: the rounded scale-format value hits zero, its subnormals, every mantissa, the saturation at the
    format maximum, and results whose fp32 exponent leaves the normal range (the kernel's fallback) -- E4M3, E5M2 and
    E3M2 scales (table), INT8 scales (no table), bit for bit against the oracle's literal fp64 restatement."""
    dt = DT[dn]
    gen = torch.Generator().manual_seed(23)
    spread = 30.0 if dn != "f16" else 3.5
    for glob in (1.0, 3.7e-3, 2.5e4, 1e-30 if dn != "f16" else 1e-4, 1e30 if dn != "f16" else 6e4):
        rows, cols = 96, 1024
        mag = torch.exp(torch.empty(rows, cols // 16, 1).uniform_(-spread, spread, generator=gen) * 2.302585) * glob
        x = (torch.randn(rows, cols // 16, 16, generator=gen) * mag).reshape(rows, cols).to(dt)
        x[0, :16] = 0
        g = torch.tensor(glob, dtype=torch.float32)
        for fmt, sfmt in (("E2M1", "E4M3"), ("E2M1", "E5M2"), ("E4M3", "E4M3"), ("E2M3", "E3M2"), ("E2M1", "INT8")):
            got = ops.fused_amax_convert(x.to(DEV), 16, fmt, sfmt, g.to(DEV))
            want = oracle.mx_fused_amax_convert(x, 16, fmt, sfmt, g)
            assert_bits_equal(got, want, f"{fmt}/{sfmt} tensor-wide amax {glob}")


@pytest.mark.parametrize("dn", ["f32", "f16", "bf16"])
def test_block_scale_formats_other_than_e8m0(dn):
    """fused_amax_convert with element-format block scales (NVFP4-style E2M1 + E4M3 scales, with and without the
    tensor-wide amax; E5M2 scales; MXFP8-style E4M3 elements with E4M3 scales): the reference test's literal rows
    (test_tensor_quant_cuda.py:205-261) and random tensors against the oracle, ragged widths included."""
    dt = DT[dn]
    for block_size in (8, 16, 32):
        for test_in, test_out in _FP4_LITERALS:
            for sign in (1.0, -1.0):
                x = torch.cat([torch.tensor([test_in]) * sign] * (block_size // 8), dim=-1).to(dt).to(DEV)
                want = torch.cat([torch.tensor([test_out]) * sign] * (block_size // 8), dim=-1).to(dt)
                got = ops.fused_amax_convert(x, 16, "E2M1", "E4M3", x.abs().amax())
                assert torch.allclose(got.cpu().float(), want.float())
    gen = torch.Generator().manual_seed(17)
    for shape, block in [((64, 256), 16), ((7, 100), 16), ((33, 48), 32), ((5, 8), 8)]:
        x = (torch.randn(*shape, generator=gen) * torch.exp(torch.randn(shape[0], 1, generator=gen) * 2)).to(dt)
        x[0, :block] = 0
        g = x.abs().amax().float()
        for fmt, sfmt, glob in [("E2M1", "E4M3", g), ("E2M1", "E4M3", None), ("E2M1", "E5M2", g), ("E4M3", "E4M3", g),
                                ("E2M3", "E4M3", None), ("E2M1", "E4M3", torch.tensor(0.0))]:
            got = ops.fused_amax_convert(x.to(DEV), block, fmt, sfmt, None if glob is None else glob.to(DEV))
            want = oracle.mx_fused_amax_convert(x, block, fmt, sfmt, glob)
            assert_bits_equal(got, want, f"{fmt}/{sfmt} global={glob is not None} {shape} block {block}")
    # the front door: dynamic_block_quant with scale_bits (4, 3) takes the quantizer's amax as the tensor-wide amax
    x = (torch.randn(16, 64, generator=gen) * 3).to(dt)
    got = ops.dynamic_block_quant(x.to(DEV), 16, x.abs().amax().to(DEV), (2, 1), (4, 3))
    assert_bits_equal(got, oracle.mx_fused_amax_convert(x, 16, "E2M1", "E4M3", x.abs().amax().float()), "nvfp4 front door")
    with pytest.raises(moa.MoquantUnsupported):
        ops.fused_amax_convert(x.to(DEV), 16, "E2M1", "E4M3", torch.ones(16, device=DEV))
