"""Pin the CPU oracle (oracle/moq_oracle.c) to the reference.

Every case compares the C restatement with outputs the reference itself produced on CPU in the build
container (tests/golden/*.npz, generator: tests/golden/gen_golden.py) or with the literal golden vectors of
the reference's own MX tests (tests/golden/mx_vectors.json).  Integer / index results must be bit-exact.
"""

import json
import os

import numpy as np
import pytest
import torch

from conftest import DT, GOLDEN, assert_bits_equal, from_bits
from oracle import oracle


def test_int_fake_quant_matches_reference(golden):
    g = golden("int_fq")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        x, y = g.t(f"{k}_x", dt), g.t(f"{k}_y", dt)
        amax = g.t(f"{k}_amax")
        if c["mode"] == "scalar":
            got = oracle.fake_quant_int(x, amax, c["bits"], c["unsigned"], c["narrow"])
        elif c["mode"] == "axis0":
            got = oracle.fake_quant_int(x, amax, c["bits"], c["unsigned"], c["narrow"],
                                        axis_size=x.shape[0], inner=x.shape[1], per_axis=True)
        else:
            got = oracle.fake_quant_int(x, amax, c["bits"], c["unsigned"], c["narrow"],
                                        axis_size=x.numel() // 32, inner=32, per_axis=True)
        assert_bits_equal(got, y, f"int_fq {k} {c}")


def test_fp8_fake_quant_matches_reference(golden):
    g = golden("fp8_fq")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        x, y = g.t(f"{k}_x", dt), g.t(f"{k}_y", dt)
        if c["mode"] == "none":
            got = oracle.fake_quant_e4m3(x, None)
        elif c["mode"] == "scalar":
            got = oracle.fake_quant_e4m3(x, g.t(f"{k}_amax"))
        else:
            got = oracle.fake_quant_e4m3(x, g.t(f"{k}_amax"), axis_size=x.shape[0], inner=x.shape[1],
                                         per_axis=True)
        assert_bits_equal(got, y, f"fp8_fq {k} {c}")


def _view3(shape, axis):
    """[outer, axis, inner] factorisation of keeping one axis of a contiguous tensor."""
    axis = axis % len(shape)
    outer = int(np.prod(shape[:axis])) if axis > 0 else 1
    inner = int(np.prod(shape[axis + 1:])) if axis + 1 < len(shape) else 1
    return outer, shape[axis], inner


def test_reduce_amax_matches_reference(golden):
    g = golden("amax")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        x, want = g.t(f"{k}_x", dt), g.t(f"{k}_a")
        axis = c["axis"]
        if axis is None:
            got = oracle.reduce_amax(x).reshape(want.shape)
        elif isinstance(axis, int):
            o, a, i = _view3(list(x.shape), axis)
            got = oracle.reduce_amax_axis(x, o, a, i).reshape(want.shape)
        else:
            continue  # multi-axis keep: composed by the adapter, covered in the adapter tests
        assert_bits_equal(got, want, f"amax {k} {c}")
        # the reference returns the input dtype; fp32 -> dtype must be lossless
        assert torch.equal(got.to(dt).float()[~torch.isnan(got)], got[~torch.isnan(got)])


def test_block_quantizer_matches_reference(golden):
    """TensorQuantizer INT4 block {-1: g}: right-pad last dim, (-1, g) view, amax, QDQ, slice back."""
    g = golden("tq_block")
    for k, c in g.cases.items():
        if c.get("kind") not in ("dynamic", "static"):
            continue
        dt = DT[c["dtype"]]
        x, y = g.t(f"{k}_x", dt), g.t(f"{k}_y", dt)
        gs = c["g"]
        cols = x.shape[-1]
        pad = (-cols) % gs
        xp = torch.nn.functional.pad(x, (0, pad)) if pad else x
        yq, am = oracle.amax_qdq_int_group(xp, gs, num_bits=4, narrow_range=False)
        got = yq[..., :cols]
        assert_bits_equal(got.contiguous(), y, f"tq_block {k} {c}")
        if c["kind"] == "static":
            assert_bits_equal(am.reshape(-1), g.t(f"{k}_amax").reshape(-1), f"tq_block amax {k}")


def test_max_calibrator_running_max(golden):
    g = golden("tq_block")
    for k, c in g.cases.items():
        if c.get("kind") != "maxcal":
            continue
        want = g.t(f"{k}_a")
        acc = None
        for b in range(3):
            x = g.t(f"{k}_b{b}", torch.bfloat16)
            if c["axis"] is None:
                a = oracle.reduce_amax(x)
            else:
                a = oracle.reduce_amax_axis(x, x.numel() // x.shape[-1], x.shape[-1], 1)
            acc = a if acc is None else torch.maximum(acc, a)
        assert_bits_equal(acc.reshape(want.shape), want, f"maxcal {k}")


def test_histogram_collect_matches_reference(golden):
    g = golden("hist")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        hist = None
        width = None
        nbins = c["num_bins"]
        for b in range(3):
            x = g.t(f"{k}_b{b}", dt)
            want_h, want_e = g.t(f"{k}_h{b}"), g.t(f"{k}_e{b}")
            xf = x.float().abs()
            if c["skip_zeros"]:
                xf = xf[xf != 0]
            x_max = xf.max()
            if hist is None:
                edge = float(x_max)
                counts = oracle.hist_abs(x, nbins, edge, c["skip_zeros"])
                hist = counts.astype(np.float32)
                edges = torch.linspace(0, x_max, nbins + 1)
            else:
                # growth rule of calib/histogram.py:121-127, host side (tiny tensors)
                if x_max > edges[-1]:
                    width = edges[1] - edges[0]
                    nbins = int((x_max / width).ceil().item())
                    edges = torch.arange(0, x_max + width, width)
                counts = oracle.hist_abs(x, nbins, float(edges[-1]), c["skip_zeros"])
                new = counts.astype(np.float32)
                new[: hist.size] += hist
                hist = new
            assert torch.equal(torch.from_numpy(hist), want_h), f"hist {k} batch {b}"
            assert torch.equal(edges, want_e), f"edges {k} batch {b}"


def test_mask_2to4_matches_reference(golden):
    g = golden("mask24")
    pats = g.t("patterns")
    assert pats.tolist() == [[0, 1, 0, 1], [1, 1, 0, 0], [0, 1, 1, 0], [1, 0, 1, 0], [1, 0, 0, 1],
                             [0, 0, 1, 1]], "reference pattern order changed"
    for k, c in g.cases.items():
        w = g.t(f"{k}_w", DT[c["dtype"]])
        want = torch.from_numpy(g.raw(f"{k}_m")).bool()
        got = oracle.mask_2to4(w)
        assert torch.equal(got, want), f"mask {k} {c}: {(got != want).sum().item()} differ"


def test_int4_qtensor_and_export_pack(golden):
    g = golden("int4")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        w = g.t(f"{k}_w", dt)
        if c["kind"] == "qtensor":
            s = g.t(f"{k}_s", dt)
            q = torch.from_numpy(g.raw(f"{k}_q"))
            got = oracle.int4_pack(w.reshape(-1), s.reshape(-1), c["g"], rounding=0)
            assert torch.equal(got, q.reshape(-1)), f"int4 pack {k}"
            deq = oracle.int4_unpack(q.reshape(-1), s.reshape(-1), c["g"])
            assert_bits_equal(deq.reshape(w.shape), g.t(f"{k}_d", dt), f"int4 unpack {k}")
        else:
            wsf = g.t(f"{k}_wsf")
            got = oracle.int4_pack_export(w, wsf)
            assert torch.equal(got, torch.from_numpy(g.raw(f"{k}_p"))), f"export pack {k}"


def test_awq_building_blocks(golden):
    g = golden("awq")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        w, x = g.t(f"{k}_w", dt), g.t(f"{k}_x", dt)
        # act scale: the reference averages |x| in the storage dtype (fp32 accumulate, one rounding)
        s64, _ = oracle.col_abs_stats(x)
        mean = (s64 / x.shape[0]).to(torch.float32).to(dt).float()
        want = g.t(f"{k}_xscale")
        ulp = (want.abs() * 2.0 ** -7).clamp_min(1e-30)
        assert ((mean - want).abs() <= ulp).all(), f"act scale {k}"
        for j, _ in enumerate(c["alphas"]):
            s = g.t(f"{k}_s{j}")
            got = oracle.awq_scale_qdq(w, s.to(dt), c["g"], 4)
            assert_bits_equal(got, g.t(f"{k}_wq{j}", dt), f"awq scale+qdq {k} alpha#{j}")
            fold = oracle.scale_cols(w, s)
            assert_bits_equal(fold, g.t(f"{k}_fold{j}", dt), f"weight fold {k} alpha#{j}")


def test_mx_golden_vectors():
    """Literal vectors of the reference's tests/gpu/torch/quantization/test_quantize_mxformats_cuda.py."""
    cases = json.load(open(os.path.join(GOLDEN, "mx_vectors.json")))
    assert len(cases) == 6
    for c in cases:
        blocks = [c["block_size"]] if c["block_size"] else [8, 16, 32]
        dtypes = [torch.float32] if c["dtype"] else [torch.float32, torch.float16, torch.bfloat16]
        for bs in blocks:
            for dt in dtypes:
                rep = max(bs // c["in_size"], 1)
                tin = torch.tensor(c["test_in"], dtype=dt).repeat(1, rep)
                tout = torch.tensor(c["test_out"], dtype=dt).repeat(1, rep)
                for sign in (1.0, -1.0):
                    got = oracle.mx_fused_amax_convert(tin * sign, bs, c["fmt"])
                    assert torch.allclose(got.float(), (tout * sign).float(), rtol=1e-5, atol=c["atol"]), \
                        f"{c['fn']} {c['fmt']} bs={bs} {dt}"


def _mse_candidates(g, name, c):
    """Candidate amax values exactly as MseCalibrator builds them (calib/mse.py:75-81, :99-101): a 0-dim fp32
    multiplier times the initial amax tensor in ITS dtype (torch type promotion keeps the dimensioned dtype)."""
    dt = {"torch.float32": torch.float32, "torch.bfloat16": torch.bfloat16, "torch.float16": torch.float16}[c["init_dtype"]]
    init = g.t(f"{name}_init_amax").to(dt).reshape(c["init_shape"])
    mult = torch.linspace(0.25, 4.0, steps=39)
    return torch.stack([(init * m).float().reshape(-1) for m in mult]), init, mult


def test_oracle_mse_sweep_matches_reference_losses(golden):
    g = golden("mse")
    for name, c in g.cases.items():
        if name.startswith("flow_"):
            continue
        w = g.t(f"{name}_w", DT[c["dtype"]])
        cand, init, mult = _mse_candidates(g, name, c)
        cfg = c["cfg"]
        nb = cfg["num_bits"]
        fp8 = isinstance(nb, list)
        if "block_sizes" in cfg:
            gsz = cfg["block_sizes"]["-1"]
            pad = (-w.shape[-1]) % gsz
            wv = torch.nn.functional.pad(w, (0, pad)).reshape(-1, gsz)
            outer, axis_size, inner = 1, wv.shape[0], gsz
        elif cfg.get("axis", None) == 0:
            wv, outer, axis_size, inner = w, 1, w.shape[0], w.shape[1]
        else:
            wv, outer, axis_size, inner = w, 1, 1, w.numel()
        got = oracle.mse_sweep(wv.contiguous(), cand, outer, axis_size, inner, fp8=fp8, num_bits=8 if fp8 else nb,
                               unsigned=False, narrow_range=False)
        want = g.t(f"{name}_losses").double()
        assert got.shape == want.shape, name
        rel = ((got - want).abs() / want.abs().clamp_min(1e-20)).max().item()
        assert rel < 2e-5, f"{name}: max rel loss diff {rel:.2e}"  # reference sums in fp32, the oracle in fp64
        best = got.argmin(0)
        amax = (init.reshape(-1) * mult[best]).float() if init.dim() else (init * mult[best]).float().reshape(-1)
        assert torch.equal(amax.reshape(-1), g.t(f"{name}_amax").reshape(-1)), f"{name}: chosen amax differs"


@pytest.mark.parametrize("name", ["clip_f32", "clip_bf16", "clip_f16_200"])
def test_awq_clip_loss_matches_reference_run(golden, name):
    """orc_awq_clip_loss vs the block losses the reference's awq_clip accumulated over three calibration batches
    (fc1 of the tiny MLP: its inputs are the stored batches).  Differences: fp32 summation order of torch's sum
    only (fp32 <= 1e-4; 16-bit: an occasional flipped final rounding, <= 2e-3); chosen clip ratio identical."""
    g = golden("awq_clip")
    c = g.cases[name]
    dt = getattr(torch, c["dtype"])
    w = g.t(f"{name}_w1", dt)
    amax = g.t(f"{name}_fc1_w_amax").to(getattr(torch, c["fc1_w_amax_dtype"])).reshape(w.shape[0], -1)
    loss = None
    for i in range(c["n_batches"]):
        x = g.t(f"{name}_x{i}", dt)
        xs = x[0::max(1, x.shape[0] // 64)].contiguous()
        loss = oracle.awq_clip_loss(xs, w, amax, c["fc1_shrinks"], 128, 4, loss)
    want = g.t(f"{name}_fc1_loss").reshape(loss.shape)
    rel = ((loss - want).abs() / want.abs().clamp_min(1e-30)).max().item()
    assert rel <= (1e-4 if dt == torch.float32 else 2e-3), f"{name}: max rel err {rel:.3e}"
    assert torch.equal(loss.argmin(0), want.argmin(0))


def test_real_quant_fp8_mxfp4_match_reference(golden):
    """orc_fp8_pack/unpack and orc_mxfp4_pack/unpack vs FP8QTensor / MXFP4QTensor run on CPU: bytes identical."""
    g = golden("qtensor")
    for k, c in g.cases.items():
        dt = DT[c["dtype"]]
        x = g.t(f"{k}_x", dt)
        want_q = torch.from_numpy(g.raw(f"{k}_q").copy())
        want_deq = g.t(f"{k}_deq", dt)
        if c["kind"] == "fp8":
            scales = g.t(f"{k}_scales", dt)
            if c["mode"] == "tensor":
                ax, inner = 1, 1
            elif c["mode"] == "axis0":
                ax, inner = x.shape[0], x.shape[1]
            else:
                ax, inner = x.numel() // 128, 128
            got_q = oracle.fp8_pack(x, scales.reshape(-1), ax, inner)
            assert torch.equal(got_q.reshape(-1), want_q.reshape(-1)), f"{k}: fp8 bytes differ"
            got = oracle.fp8_unpack(want_q.reshape(x.shape), scales.reshape(-1), dt, ax, inner)
            assert_bits_equal(got, want_deq, f"{k}: fp8 dequant")
        else:
            got_q, got_e = oracle.mxfp4_pack(x, c["block"])
            assert torch.equal(got_e.reshape(-1), torch.from_numpy(g.raw(f"{k}_e8m0").copy()).reshape(-1)), f"{k}: e8m0"
            assert torch.equal(got_q.reshape(-1), want_q.reshape(-1)), f"{k}: mxfp4 bytes differ"
            got = oracle.mxfp4_unpack(want_q, torch.from_numpy(g.raw(f"{k}_e8m0").copy()), dt, c["block"])
            assert_bits_equal(got, want_deq, f"{k}: mxfp4 dequant")


def test_block2d_matches_reference(golden):
    """orc_block2d vs TensorQuantizer with blocks on both axes run by the reference on CPU: amax and QDQ bit-exact."""
    g = golden("block2d")
    for k, c in g.cases.items():
        if "grid" in c:
            continue  # grids on other axis sets are a host-side permutation onto the per-row entries (test_gpu_host.py)
        dt = DT[c["dtype"]]
        x, want = g.t(f"{k}_x", dt), g.t(f"{k}_y", dt)
        fp8 = isinstance(c["num_bits"], list)
        if "lead" in c:
            # tiles on the last two axes of a rank-3 / rank-4 tensor: padded to whole tiles, the leading dims folded
            # into the tile rows -- the SAME entry point
            x, want = x.reshape(c["shape"]), want.reshape(c["shape"])
            rows, cols = c["shape"][-2:]
            padded = torch.nn.functional.pad(x, (0, (-cols) % c["bc"], 0, (-rows) % c["br"]))
            y, am = oracle.block2d(padded.reshape(-1, padded.shape[-1]).contiguous(), c["br"], c["bc"], 2, fp8=fp8,
                                   num_bits=8 if fp8 else c["num_bits"])
            y = y.reshape(padded.shape)[..., :rows, :cols]
        else:
            y, am = oracle.block2d(x, c["br"], c["bc"], 2, fp8=fp8, num_bits=8 if fp8 else c["num_bits"])
        assert torch.equal(am.reshape(-1), g.t(f"{k}_amax").reshape(-1)), f"{k}: block amax"
        assert_bits_equal(y, want, f"block2d {k} {c}")


def _np_range(w):
    mx = w.abs().amax(1)
    zero = mx == 0
    return torch.where(zero, torch.full_like(mx, -0.5), torch.zeros_like(mx)), torch.where(zero, torch.full_like(mx, 0.5), mx)


def test_row_histograms_match_numpy_run_by_the_reference(golden):
    """calibrate_weights' per-channel np.histogram(|w_row|, 2048, range=(0, max)) counts, as produced next to the
    reference run (values on bin edges, an all-zero channel and heavy tails included)."""
    g = golden("calibrate_weights")
    for k, c in g.cases.items():
        w = g.t(f"{k}_w", torch.float32)
        first, last = _np_range(w)
        got = oracle.row_hist_np(w, 2048, first, last)
        want = torch.from_numpy(g.raw(f"{k}_hist").copy())
        assert torch.equal(got, want), f"{k} ({c['kind']}): {(got != want).sum().item()} bins differ"
        assert int(got.sum()) == w.numel()


def test_fp8_tile_pack_matches_reference(golden):
    """FP8QTensor.quantize / dequantize with blocks on both axes (fp8_tensor.py:60-151) run by the reference: the
    oracle's tile composition of the pinned per-tensor pack gives the same bytes and dequantised values."""
    import torch.nn.functional as F

    g = golden("export_llama_fp8_2d")
    for k, c in g.cases["qt"].items():
        dt = DT[c["dtype"]]
        x = g.t(f"{k}_x", dt)
        br, bc = c["blocks"]["-2"], c["blocks"]["-1"]
        sdt = torch.float32 if c["scale_dtype"] == "torch.float32" else dt
        scales = g.t(f"{k}_scales", sdt)
        xp = F.pad(x, (0, (-x.shape[1]) % bc, 0, (-x.shape[0]) % br))
        got = oracle.fp8_pack_tile(xp, scales, br, bc)[: x.shape[0], : x.shape[1]]
        want = torch.from_numpy(g.raw(f"{k}_q").copy())
        assert torch.equal(got, want), f"{k}: {(got != want).sum().item()} bytes differ"
        qp = F.pad(want, (0, (-x.shape[1]) % bc, 0, (-x.shape[0]) % br))
        deq = oracle.fp8_unpack_tile(qp, scales.to(dt), dt, br, bc)[: x.shape[0], : x.shape[1]]
        assert_bits_equal(deq, g.t(f"{k}_deq", dt), f"{k} dequant")


_FP4_LITERALS = [  # tests/gpu/torch/quantization/test_tensor_quant_cuda.py:236-258 (test_cuda_ext_fp4)
    ([0, 0.5, 1, 1.5, 2, 3, 4, 6], [0, 0.5, 1, 1.5, 2, 3, 4, 6]),                      # table values
    ([0.25, 0.75, 1.25, 1.75, 2.5, 3.5, 5, 6], [0.0, 1, 1, 2, 2, 4, 4, 6]),            # exact ties: even code
    ([0.15, 0.65, 1.15, 1.65, 2.4, 3.4, 4.9, 6], [0.0, 0.5, 1, 1.5, 2, 3, 4, 6]),      # just below the ties
    ([0.35, 0.85, 1.35, 1.85, 2.6, 3.6, 5.1, 6], [0.5, 1, 1.5, 2, 3, 4, 6, 6]),        # just above
]


@pytest.mark.parametrize("dn", ["f32", "f16", "bf16"])
@pytest.mark.parametrize("block_size", [8, 16, 32])
def test_two_level_fp4_scaling_literal_vectors(dn, block_size):
    """fused_amax_convert(inputs, 16, E2M1, E4M3 block scales, global amax) on the reference test's literal rows:
    pins compute_scale_with_global (tensor_quant_mx.cu:154-183) in the oracle."""
    dt = DT[dn]
    for test_in, test_out in _FP4_LITERALS:
        for sign in (1.0, -1.0):
            x = torch.cat([torch.tensor([test_in]) * sign] * (block_size // 8), dim=-1).to(dt)
            want = torch.cat([torch.tensor([test_out]) * sign] * (block_size // 8), dim=-1).to(dt)
            got = oracle.mx_fused_amax_convert(x, 16, "E2M1", "E4M3", x.abs().amax())
            assert torch.allclose(got.float(), want.float()), f"{dn} bs={block_size}: {got} vs {want}"
