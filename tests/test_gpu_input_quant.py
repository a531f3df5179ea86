"""The fused input-quantizer pass (moq_input_quant): pre_quant_scale * x -> running abs-max / |.| histogram
-> INT-k / FP8 quantize-dequantize in one read, against the chain of the oracle's single stages (bit-exact) and against
the unfused kernels; TensorQuantizer.forward takes it for per-tensor quantizers with a pre_quant_scale; the histogram
calibrator's later batches get abs-max + counts from one pass."""

import numpy as np
import pytest
import torch

import _moa_import
from conftest import assert_bits_equal

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
ops = moa.ops
from oracle import oracle  # noqa: E402  (the checker)

DEV = "cuda:0"


def _acts(rows, cols, dtype, seed, outliers=True):
    g = torch.Generator().manual_seed(seed)
    ch = torch.exp(torch.randn(cols, generator=g))
    if outliers:
        ch[:3] *= 50.0
    x = torch.randn(rows, cols, generator=g) * ch
    x[0, 0] = 0.0
    return x.to(dtype)


def _want(x, pqs, fmt, amax_q, bins, edge, skip):
    v = oracle.scale_cols(x, pqs.to(x.dtype).float()) if pqs is not None else x
    amax = oracle.reduce_amax(v).float()
    hist = oracle.hist_abs(v.reshape(-1), bins, edge, skip).astype(np.int64) if bins else None
    if fmt == "int8":
        y = oracle.fake_quant_int(v, amax_q.reshape(1), 8, False, False)
    elif fmt == "int4u":
        y = oracle.fake_quant_int(v, amax_q.reshape(1), 4, True, True)
    elif fmt == "fp8":
        y = oracle.fake_quant_e4m3(v, amax_q.reshape(1))
    else:
        y = v
    return y, amax, hist


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("fmt", [None, "int8", "int4u", "fp8"])
@pytest.mark.parametrize("with_pqs", [False, True])
def test_input_quant_equals_stage_chain(dtype, fmt, with_pqs):
    for rows, cols in [(257, 1024), (64, 4096), (3, 136)]:  # whole chunks + ragged tail; rows shorter than a chunk
        x = _acts(rows, cols, dtype, rows)
        if fmt == "int4u":
            x = x.abs()  # the reference refuses negative values in unsigned quantization (tensor_quant.py:610-611)
            x[1, 1] = -0.0  # ... but not a negative zero, whose sign survives the clamp at +0
        g = torch.Generator().manual_seed(cols)
        pqs = torch.exp(torch.randn(cols, generator=g) * 0.5).to(dtype) if with_pqs else None
        bins, skip = 2048, False
        edge = float((x.float() * (pqs.float() if with_pqs else 1.0)).abs().max()) * 0.5  # part of the data is outside
        amax_q = torch.tensor(float(x.float().abs().max()) * 0.7)
        y_want, amax_want, hist_want = _want(x, pqs, fmt, amax_q, bins, edge, skip)
        xd = x.to(DEV)
        running = torch.tensor([0.25], dtype=torch.float32, device=DEV)
        counts = torch.zeros(bins, dtype=torch.int64, device=DEV)
        nb = {"int8": 8, "int4u": 4, "fp8": (4, 3), None: None}[fmt]
        y = ops.input_quant(xd, pqs.to(DEV) if with_pqs else None, amax_running=running,
                            qdq_amax=amax_q.to(DEV) if fmt else None, num_bits=nb, unsigned=fmt == "int4u",
                            narrow_range=fmt == "int4u", hist_counts=counts, hist_max_edge=edge, hist_skip_zeros=skip)
        assert running.item() == max(0.25, amax_want.item())
        assert np.array_equal(counts.cpu().numpy(), hist_want), f"{rows}x{cols}: histogram differs"
        if fmt or with_pqs:
            assert_bits_equal(y.cpu(), y_want, f"{dtype} {fmt} pqs={with_pqs} {rows}x{cols}")
        else:
            assert y is None
        # the same stages one by one on the GPU
        v = ops.scale_cols(xd, pqs.to(DEV)) if with_pqs else xd
        assert np.array_equal(ops.hist_abs(v, bins, edge, skip).cpu().numpy(), hist_want)
        if fmt == "fp8":
            assert_bits_equal(ops.scaled_e4m3(v, amax_q.to(DEV)), y, "fused vs unfused FP8")


def test_histogram_hot_bins_and_tails():
    """Counts are exact whatever the distribution: everything in the lowest bins, flat over the range, one constant,
    skip_zeros, NaN / inf / out-of-range values dropped, bin counts at the edges of what the kernel takes; 16-bit inputs
    go through the per-workgroup pattern table, fp32 through the arithmetic rule."""
    g = torch.Generator().manual_seed(0)
    n = 8192 * 40 + 13
    cases = {
        "hot": torch.randn(n, generator=g) * 1e-3,
        "flat": torch.rand(n, generator=g) * 2 - 1,
        "const": torch.full((n,), 0.0004),
        "mixed": torch.cat([torch.randn(n // 2, generator=g) * 0.01, torch.rand(n - n // 2, generator=g)]),
    }
    cases["mixed"][::1001] = float("nan")
    cases["mixed"][5::1003] = float("inf")
    cases["flat"][7::97] = 0.0
    for name, x in cases.items():
        for dtype in (torch.bfloat16, torch.float16, torch.float32):
            xv = x.to(dtype)
            for bins, edge, skip in [(2048, 1.0, False), (2048, 0.75, True), (8, 1.0, False), (1, 0.5, False),
                                     (16383, 1.0, False), (4096, 3.0, True)]:
                want = oracle.hist_abs(xv, bins, edge, skip).astype(np.int64)
                got = ops.hist_abs(xv.to(DEV), bins, edge, skip).cpu().numpy()
                assert np.array_equal(got, want), f"{name} {dtype} bins={bins} edge={edge} skip={skip}"
                # and accumulation into existing counts
                acc = torch.ones(bins, dtype=torch.int64, device=DEV)
                ops.hist_abs(xv.to(DEV), bins, edge, skip, counts=acc)
                assert np.array_equal(acc.cpu().numpy(), want + 1)
                if dtype != torch.float32:
                    # moq_hist_abs sends 16-bit inputs below 12 M elements to the arithmetic kernel (round 3); the pattern-table
                    # kernel is reached at any size through the fused input-quantizer entry
                    tab = torch.zeros(bins, dtype=torch.int64, device=DEV)
                    ops.input_quant(xv.to(DEV), None, hist_counts=tab, hist_max_edge=edge, hist_skip_zeros=skip)
                    assert np.array_equal(tab.cpu().numpy(), want), f"table kernel: {name} {dtype} bins={bins}"


def test_histogram_pattern_counters_and_table_give_the_same_counts():
    """Round 6: 16-bit inputs whose bins fit beside 128 KiB of LDS pattern counters are counted per |x| PATTERN and binned
    at the flush (bins <= 7039); more bins take the round-2 pattern TABLE.  Both sides of the switch, zero-heavy data (zeros
    have a counter per lane), a view that starts off a 16-byte boundary, a ragged end, and the pre_quant_scale stage in
    front (the patterns counted are those of dtype(x * s))."""
    g = torch.Generator().manual_seed(11)
    n = 8192 * 24 + 8 * 5 + 3
    base = torch.randn(n + 8, generator=g) * torch.exp(torch.randn(n + 8, generator=g))
    relu = torch.relu(base)
    relu[::7] = -0.0
    for dtype in (torch.bfloat16, torch.float16):
        for name, x in (("gauss", base), ("relu", relu)):
            xv = x.to(dtype)
            edge = float(xv.float().abs().max())
            for off in (0, 1, 3):
                xs = xv[off:off + n]
                for bins, skip in [(7039, False), (7040, False), (2048, True), (4095, False), (3, True)]:
                    want = oracle.hist_abs(xs, bins, edge * 0.9, skip).astype(np.int64)
                    tab = torch.zeros(bins, dtype=torch.int64, device=DEV)
                    ops.input_quant(xs.to(DEV) if off == 0 else xv.to(DEV)[off:off + n], None, hist_counts=tab,
                                    hist_max_edge=edge * 0.9, hist_skip_zeros=skip)
                    assert np.array_equal(tab.cpu().numpy(), want), f"{name} {dtype} off={off} bins={bins} skip={skip}"
    # pre_quant_scale in front of the histogram and the running abs-max (2-D: one scale per column)
    rows, cols = 96 * 3 + 1, 1024
    x = (torch.randn(rows, cols, generator=g) * 0.5).to(torch.bfloat16)
    s = (torch.rand(cols, generator=g) * 4).to(torch.bfloat16)
    s[::17] = 0.0
    v = (x.float() * s.float()).to(torch.bfloat16)
    edge = float(v.float().abs().max())
    for bins in (2048, 7040):
        want = oracle.hist_abs(v.reshape(-1), bins, edge, False).astype(np.int64)
        tab = torch.zeros(bins, dtype=torch.int64, device=DEV)
        amax = torch.zeros(1, dtype=torch.float32, device=DEV)
        out = ops.input_quant(x.to(DEV), s.to(DEV), amax_running=amax, hist_counts=tab, hist_max_edge=edge)
        assert np.array_equal(tab.cpu().numpy(), want)
        assert float(amax) == edge
        assert_bits_equal(out, v, "scaled activation")


def test_histogram_table_dispatch_at_flow_size():
    """13 M elements (26 MB, above moq_hist_abs' 12 M threshold): the pattern-table kernel through moq_hist_abs itself."""
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(13 * (1 << 20) + 8, generator=g) * torch.exp(torch.randn(1, generator=g))).to(torch.bfloat16)
    x[::100003] = 0.0
    for bins, edge, skip in [(2048, 2.5, False), (2048, 1.0, True)]:
        want = oracle.hist_abs(x, bins, edge, skip).astype(np.int64)
        assert np.array_equal(ops.hist_abs(x.to(DEV), bins, edge, skip).cpu().numpy(), want)


def test_tensor_quantizer_takes_the_fused_pass():
    """A per-tensor input quantizer with a pre_quant_scale: calibration (x * s handed on, running amax), quantization and
    both at once equal the unfused chain bit for bit."""
    TQ, Cfg = moa.TensorQuantizer, moa.QuantizerAttributeConfig
    x1, x2 = _acts(96, 512, torch.bfloat16, 1).to(DEV), _acts(40, 512, torch.bfloat16, 2).to(DEV)
    pqs = torch.exp(torch.randn(512, generator=torch.Generator().manual_seed(3)) * 0.3).to(torch.bfloat16).to(DEV)
    for nb in (8, (4, 3)):
        q = TQ(Cfg(num_bits=nb, axis=None))
        q.pre_quant_scale = pqs
        q.disable_quant()
        q.enable_calib()
        calls = []
        real = ops.input_quant
        ops.input_quant = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        try:
            o1, o2 = q(x1), q(x2)
            assert len(calls) == 2
            assert_bits_equal(o1, x1 * pqs, "calibration pass hands x * s on")
            q.load_calib_amax()
            want_amax = torch.maximum((x1 * pqs).abs().max(), (x2 * pqs).abs().max())
            assert q.amax.dtype == torch.bfloat16 and q.amax.item() == want_amax.item()
            q.enable_quant()
            q.disable_calib()
            y = q(x1)
            assert len(calls) == 3
        finally:
            ops.input_quant = real
        v = ops.scale_cols(x1, pqs)
        want = ops.scaled_e4m3(v, q.amax) if nb == (4, 3) else ops.fake_tensor_quant(v, q.amax, 8, False, False)
        assert_bits_equal(y, want, f"fused forward, num_bits={nb}")


def test_histogram_calibrator_later_batches_are_one_pass():
    from model_optimizer_amd.calib import HistogramCalibrator

    batches = [_acts(64, 1024, torch.bfloat16, s).to(DEV) for s in range(4)]
    batches[2][5, 5] = 9000.0  # exceeds the range: the optimistic counts are dropped, the batch is binned again
    cal = HistogramCalibrator(8, None, False)
    ref_hist, ref_edges = None, None
    for b in batches:
        cal.collect(b)
    # the reference's sequence restated with the unfused ops
    x_max = ops.reduce_amax(batches[0]).float().cpu()
    edges = torch.linspace(0, x_max, 2049)
    hist = ops.hist_abs(batches[0], 2048, float(x_max))
    nbins = 2048
    for b in batches[1:]:
        xm = ops.reduce_amax(b).float().cpu()
        if xm > edges[-1]:
            width = edges[1] - edges[0]
            nbins = int((xm / width).ceil().item())
            edges = torch.arange(0, xm + width, width)
            grown = torch.zeros(nbins, dtype=torch.int64, device=DEV)
            grown[: hist.numel()] = hist
            hist = grown
        ops.hist_abs(b, nbins, float(edges[-1]), counts=hist)
    assert torch.equal(cal._calib_hist, hist) and torch.equal(cal._calib_bin_edges, edges)
    assert int(hist.sum()) == sum(b.numel() for b in batches)
