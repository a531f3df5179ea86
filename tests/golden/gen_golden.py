"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE on CPU.

Run in the build container only (needs /root/reference):  python tests/golden/gen_golden.py
The GPU box has no reference checkout; tests there read the committed .npz files.

What is produced (every array is the reference's own output on seeded inputs that are stored next to it):
  int_fq.npz    -- _tensor_quant / fake_tensor_quant (quantization/tensor_quant.py:607-645)
  fp8_fq.npz    -- _fp8_eager (quantization/tensor_quant.py:46-59)
  amax.npz      -- reduce_amax (quantization/utils/core_utils.py:146-183)
  tq_block.npz  -- TensorQuantizer static-block INT4 forward incl. padding
                   (nn/modules/tensor_quantizer.py:975-1061, :1119-1221) and MaxCalibrator
  hist.npz      -- HistogramCalibrator.collect with bin growth (calib/histogram.py:77-130)
  mask24.npz    -- create_asp_mask (sparsity/weight_sparsity/magnitude.py:91-128) + pattern order
  int4.npz      -- INT4QTensor.quantize/dequantize eager twin (qtensor/int4_tensor.py:39-130) and
                   pack_int4_in_uint8 (export/quant_utils.py:792-833)
  awq.npz       -- AWQ-lite building blocks on one linear (quantization/model_calib.py:1453-1495)
  model_flows.npz -- mtq.quantize() end to end on a tiny MLP: max (INT8, FP8), smoothquant, awq_lite
  export_llama.npz -- INT4-AWQ export_hf_checkpoint of a tiny Llama: pre-export state and exported tensors
  export_llama_replay.npz -- the same run: every linear's input per calibration batch + the search's statistics
  sq_mxfp4.npz  -- configs[4]: MXFP4QTensor.quantize of the INT8-SmoothQuant run's smoothed weights (packed bytes, E8M0)
  awq_clip.npz  -- mtq.quantize() with awq_clip / awq_full: w_amax, per-shrink block losses, best_clip_val
  block2d.npz   -- TensorQuantizer with blocks on both axes (FP8 128x128, INT8 64x32): amax + fake-quant output
  sgpt.npz      -- SparseGPT: hook-accumulated Hessian, prepared inverse factor, create_sgpt_mask result
  gptq.npz      -- GPTQ: Hessian, inverse factor and updated weights of one linear per format; mtq.quantize(gptq) on the tiny MLP
  local_hessian.npz -- the local-Hessian amax search (INT4 blocks of 16) on the tiny MLP: amax after max / mse / local_hessian
  gptq_llama.npz -- mtq.quantize(algorithm = gptq) on a tiny Llama in one pass (updated weights, logits)
  w4a8.npz      -- SequentialQuantizer (INT4 blocks -> FP8) weights + FP8 inputs, max calibration
  qtensor.npz   -- FP8QTensor / MXFP4QTensor quantize + dequantize (bytes, scales, dequantised values)
  export_llama_mxfp4.npz -- MXFP4 export_hf_checkpoint of the tiny Llama (packed nibbles + E8M0 scales)
  export_llama_w4a8_mxfp4_fp8.npz, export_llama_mxfp4_mlp.npz -- the two other MXFP4 presets' exports (tensors + hf_quant_config)
  export_configs.npz -- hf_quant_config.json, config.json's quantization_config and the file list of export_hf_checkpoint for twelve presets
  export_llama_fp8.npz -- FP8 export_hf_checkpoint of the tiny Llama (amax state, exported tensors)
  export_llama_fp8_kv.npz -- FP8 + FP8 KV-cache quantizers on the tiny Llama: k / v amax, logits, k_scale / v_scale
  moe_fp8.npz   -- FP8 on a tiny Mixtral with fused 3-D expert weights: per-expert amax, logits, exported tensors
  calibrate_weights.npz -- calib.calibrate_weights per-channel / per-tensor percentile amax + numpy's channel histograms
  export_llama_fp8_2d.npz -- FP8 2-D blockwise weight-only export of a tiny Llama + FP8QTensor with blocks on both axes
  export_llama_int8_sq.npz -- INT8 SmoothQuant export of the tiny Llama (pre-export state + exported tensors)
  mse.npz       -- MseCalibrator losses / chosen amax (calib/mse.py:83-172) and mtq.quantize(algorithm="mse")
  affine_bias.npz -- TensorQuantizer with an affine offset (`bias`: static / dynamic, mean / max_min, per tensor / per head and
                   channel; FP8 and INT8): calibrated offset and amax, outputs
  mx_vectors.json -- the literal MX golden vectors of tests/gpu/torch/quantization/
                   test_quantize_mxformats_cuda.py (extracted from the test source with ast, not run:
                   the MX kernels have no CPU implementation in the reference)
"""

import ast
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

ref_shim.install()

from modelopt.torch.quantization import tensor_quant as tq  # noqa: E402
from modelopt.torch.quantization import utils as quant_utils  # noqa: E402
from modelopt.torch.quantization.calib import HistogramCalibrator, MaxCalibrator  # noqa: E402
from modelopt.torch.quantization.config import QuantizerAttributeConfig  # noqa: E402
from modelopt.torch.quantization.nn import TensorQuantizer  # noqa: E402
from modelopt.torch.quantization.qtensor import INT4QTensor  # noqa: E402
from modelopt.torch.sparsity.weight_sparsity import magnitude  # noqa: E402
from modelopt.torch.export.quant_utils import pack_int4_in_uint8  # noqa: E402

DT = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}


def bits(t: torch.Tensor) -> np.ndarray:
    """Store tensors losslessly: 16-bit floats as uint16 patterns, the rest as-is."""
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    return t.numpy().copy()


def weight_like(shape, dtype, seed, outliers=True):
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(*shape, generator=g) * 0.02
    if outliers:
        m = torch.rand(*shape, generator=g) < 0.001
        w = torch.where(m, w * 8, w)
    return w.to(dtype)


def gen_int_fq(out):
    cases = {}
    idx = 0
    for dn, dt in DT.items():
        for bits_, unsigned, narrow in [(8, False, True), (8, False, False), (4, False, False),
                                        (3, False, True), (8, True, False), (5, False, False)]:
            x = weight_like((24, 128), dt, 100 + idx)
            if unsigned:
                x = x.abs()
            # scalar amax
            amax = x.abs().amax().float()
            y = tq._tensor_quant(x, amax, bits_, unsigned, narrow)
            cases[f"c{idx}"] = dict(dtype=dn, bits=bits_, unsigned=unsigned, narrow=narrow, mode="scalar")
            out[f"c{idx}_x"], out[f"c{idx}_amax"], out[f"c{idx}_y"] = bits(x), bits(amax.reshape(1)), bits(y)
            idx += 1
            # per-channel amax (axis 0)
            amax = x.abs().amax(dim=1, keepdim=True).float()
            y = tq._tensor_quant(x, amax, bits_, unsigned, narrow)
            cases[f"c{idx}"] = dict(dtype=dn, bits=bits_, unsigned=unsigned, narrow=narrow, mode="axis0")
            out[f"c{idx}_x"], out[f"c{idx}_amax"], out[f"c{idx}_y"] = bits(x), bits(amax), bits(y)
            idx += 1
            # per-group (g=32) amax via the (-1, g) view
            xg = x.reshape(-1, 32)
            amax = xg.abs().amax(dim=1, keepdim=True).float()
            y = tq._tensor_quant(xg, amax, bits_, unsigned, narrow).reshape(x.shape)
            cases[f"c{idx}"] = dict(dtype=dn, bits=bits_, unsigned=unsigned, narrow=narrow, mode="group32")
            out[f"c{idx}_x"], out[f"c{idx}_amax"], out[f"c{idx}_y"] = bits(x), bits(amax), bits(y)
            idx += 1
    # tiny / zero amax rows (tests/gpu/torch/quantization/test_tensor_quant_cuda.py:115-119)
    x = weight_like((4, 64), torch.float32, 7)
    amax = torch.tensor([[0.0], [2.0 ** -24], [2.0 ** -23], [1.0]])
    y = tq._tensor_quant(x, amax, 8, False, True)
    cases[f"c{idx}"] = dict(dtype="f32", bits=8, unsigned=False, narrow=True, mode="axis0")
    out[f"c{idx}_x"], out[f"c{idx}_amax"], out[f"c{idx}_y"] = bits(x), bits(amax), bits(y)
    out["cases"] = np.array(json.dumps(cases))


def gen_fp8(out):
    cases = {}
    idx = 0
    special = torch.tensor([0.0, -0.0, 1e-4, 0.0019, 0.00195, 2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10, 0.0156,
                            1.0, 1.0625, 1.125, 1.1875, 240.0, 447.0, 448.0, 449.0, 463.9, 464.0, 464.1,
                            479.0, 480.0, 1e6, -1e6, float("inf"), float("-inf"), float("nan"), -449.0,
                            -464.0, -465.0, 17.0, 18.0, 19.0, 20.0, 21.0, 22.0, 23.0, 25.0, 27.0, 0.3])
    for dn, dt in DT.items():
        x = torch.cat([special.to(dt).float(), (weight_like((32, 128), dt, 200 + idx).float() * 40).reshape(-1)])
        x = x[: x.numel() // 8 * 8].reshape(-1, 8).to(dt)
        # amax=None: plain cast
        y = tq._fp8_eager(x, None)
        cases[f"c{idx}"] = dict(dtype=dn, mode="none")
        out[f"c{idx}_x"], out[f"c{idx}_y"] = bits(x), bits(y)
        idx += 1
        for amax_v in [x[torch.isfinite(x)].abs().max().float(), torch.tensor(3.0), torch.tensor(0.0),
                       torch.tensor(2.0 ** -24), torch.tensor(1e-3)]:
            y = tq._fp8_eager(x, amax_v)
            cases[f"c{idx}"] = dict(dtype=dn, mode="scalar")
            out[f"c{idx}_x"], out[f"c{idx}_amax"], out[f"c{idx}_y"] = bits(x), bits(amax_v.reshape(1)), bits(y)
            idx += 1
        xw = weight_like((48, 128), dt, 300 + idx)
        amax = xw.abs().amax(dim=1, keepdim=True).float()
        y = tq._fp8_eager(xw, amax)
        cases[f"c{idx}"] = dict(dtype=dn, mode="axis0")
        out[f"c{idx}_x"], out[f"c{idx}_amax"], out[f"c{idx}_y"] = bits(xw), bits(amax), bits(y)
        idx += 1
    out["cases"] = np.array(json.dumps(cases))


def gen_amax(out):
    cases = {}
    idx = 0
    for dn, dt in DT.items():
        x = weight_like((6, 40, 24), dt, 400 + idx)
        for axis in [None, 0, 1, 2, -1, (0, 2)]:
            red = quant_utils.convert_quantization_axis_to_reduce_axis(x, axis)
            a = quant_utils.reduce_amax(x, axis=red)
            cases[f"c{idx}"] = dict(dtype=dn, axis=axis if not isinstance(axis, tuple) else list(axis),
                                    out_dtype=str(a.dtype), out_shape=list(a.shape))
            out[f"c{idx}_x"], out[f"c{idx}_a"] = bits(x), bits(a.float())
            idx += 1
    # NaN / inf propagation
    x = torch.tensor([[1.0, float("nan"), -3.0], [2.0, float("-inf"), 0.5]])
    for axis in [None, 0]:
        red = quant_utils.convert_quantization_axis_to_reduce_axis(x, axis)
        a = quant_utils.reduce_amax(x, axis=red)
        cases[f"c{idx}"] = dict(dtype="f32", axis=axis, out_dtype=str(a.dtype), out_shape=list(a.shape))
        out[f"c{idx}_x"], out[f"c{idx}_a"] = bits(x), bits(a.float())
        idx += 1
    out["cases"] = np.array(json.dumps(cases))


def gen_tq_block(out):
    """TensorQuantizer(INT4, block {-1: g}) standalone: dynamic amax + QDQ, with and without padding,
    then MaxCalibrator static path."""
    cases = {}
    idx = 0
    for dn, dt in [("bf16", torch.bfloat16), ("f16", torch.float16), ("f32", torch.float32)]:
        for shape, g in [((32, 256), 128), ((16, 200), 128), ((8, 96), 32), ((4, 3, 64), 16)]:
            w = weight_like(shape, dt, 500 + idx)
            q = TensorQuantizer(QuantizerAttributeConfig(num_bits=4, block_sizes={-1: g}))
            y = q(w)  # no _amax buffer -> dynamic per-block amax
            cases[f"c{idx}"] = dict(dtype=dn, g=g, kind="dynamic")
            out[f"c{idx}_x"], out[f"c{idx}_y"] = bits(w), bits(y)
            idx += 1
            # static: calibrate then quantize
            q = TensorQuantizer(QuantizerAttributeConfig(num_bits=4, block_sizes={-1: g}))
            q.disable_quant(); q.enable_calib()
            q(w)
            q.load_calib_amax(); q.enable_quant(); q.disable_calib()
            y = q(w)
            cases[f"c{idx}"] = dict(dtype=dn, g=g, kind="static", amax_dtype=str(q._amax.dtype),
                                    amax_shape=list(q._amax.shape))
            out[f"c{idx}_x"], out[f"c{idx}_y"], out[f"c{idx}_amax"] = bits(w), bits(y), bits(q._amax.float())
            idx += 1
    # MaxCalibrator running max over batches, per-tensor and per-channel(axis=-1 -> reduce others)
    g_ = torch.Generator().manual_seed(9)
    batches = [torch.randn(4, 16, 64, generator=g_).to(torch.bfloat16) * (i + 1) for i in range(3)]
    for axis in [None, -1]:
        cal = MaxCalibrator(8, axis, False)
        for b in batches:
            cal.collect(b)
        a = cal.compute_amax()
        cases[f"c{idx}"] = dict(kind="maxcal", axis=axis, out_dtype=str(a.dtype), out_shape=list(a.shape))
        for k, b in enumerate(batches):
            out[f"c{idx}_b{k}"] = bits(b)
        out[f"c{idx}_a"] = bits(a.float())
        idx += 1
    out["cases"] = np.array(json.dumps(cases))


def gen_hist(out):
    cases = {}
    idx = 0
    for dn, dt in [("bf16", torch.bfloat16), ("f32", torch.float32), ("f16", torch.float16)]:
        for skip_zeros in [False, True]:
            g_ = torch.Generator().manual_seed(600 + idx)
            b0 = (torch.randn(64, 257, generator=g_)).to(dt)
            b1 = (torch.randn(64, 257, generator=g_) * 1.7).to(dt)   # larger max -> bins grow
            b2 = (torch.randn(64, 257, generator=g_) * 0.5).to(dt)   # smaller max -> same edges
            b0.view(-1)[::37] = 0
            cal = HistogramCalibrator(8, None, False, num_bins=256, skip_zeros=skip_zeros)
            snaps = []
            for b in (b0, b1, b2):
                cal.collect(b)
                snaps.append((cal._calib_hist.clone(), cal._calib_bin_edges.clone()))
            cases[f"c{idx}"] = dict(dtype=dn, skip_zeros=skip_zeros, num_bins=256)
            out[f"c{idx}_pct"] = bits(cal.compute_amax("percentile", percentile=99.9).float().reshape(1))
            out[f"c{idx}_ent"] = bits(cal.compute_amax("entropy", start_bin=64).float().reshape(1))
            # "mse" as the reference COMPUTES it (bit width in the bias slot: see calib._compute_amax_mse here): through the
            # calibrator, and on the same counts with unsigned / other widths / strides and edges scaled so that the
            # count-weighted mean of the centres lies above num_bits (the non-degenerate branch)
            from modelopt.torch.quantization.calib import histogram as ref_hist

            out[f"c{idx}_mse"] = bits(cal.compute_amax("mse", start_bin=64).float().reshape(1))
            h_np, e_np = cal._calib_hist.cpu().numpy(), cal._calib_bin_edges.cpu().numpy()
            out[f"c{idx}_mseu"] = bits(ref_hist._compute_amax_mse(h_np, e_np, 8, True, 1, 64).float().reshape(1))
            out[f"c{idx}_mse100"] = bits(ref_hist._compute_amax_mse(h_np, e_np * 100, 8, False, 1, 16).float().reshape(1))
            out[f"c{idx}_mse4s3"] = bits(ref_hist._compute_amax_mse(h_np, e_np * 37, 4, False, 3, 16).float().reshape(1))
            for k, b in enumerate((b0, b1, b2)):
                out[f"c{idx}_b{k}"] = bits(b)
                out[f"c{idx}_h{k}"] = bits(snaps[k][0])
                out[f"c{idx}_e{k}"] = bits(snaps[k][1])
            idx += 1
    out["cases"] = np.array(json.dumps(cases))


def gen_mask(out):
    cases = {}
    idx = 0
    pats = magnitude.compute_valid_1d_patterns(4, 2)
    out["patterns"] = bits(pats)
    for dn, dt in DT.items():
        w = weight_like((64, 128), dt, 700 + idx)
        # force ties: coarse values, zeros, equal groups
        w2 = (torch.randint(-3, 4, (32, 64), generator=torch.Generator().manual_seed(701 + idx)).float()).to(dt)
        w2[0, :8] = 0
        w2[1, :8] = 1.5
        for name, t in (("rand", w), ("ties", w2)):
            m = magnitude.create_asp_mask(t, "2:4 sparsity")
            cases[f"c{idx}"] = dict(dtype=dn, kind=name)
            out[f"c{idx}_w"], out[f"c{idx}_m"] = bits(t), m.numpy().astype(np.uint8)
            idx += 1
    # special values
    t = torch.tensor([[3.0, -3.0, 1.0, 3.0], [float("inf"), 1.0, 2.0, 3.0], [float("nan"), 5.0, 1.0, 2.0],
                      [0.0, 0.0, 0.0, 0.0]] * 2).reshape(8, 4).repeat(1, 4)
    m = magnitude.create_asp_mask(t, "2:4 sparsity")
    cases[f"c{idx}"] = dict(dtype="f32", kind="special")
    out[f"c{idx}_w"], out[f"c{idx}_m"] = bits(t), m.numpy().astype(np.uint8)
    out["cases"] = np.array(json.dumps(cases))


def gen_int4(out):
    cases = {}
    idx = 0
    for dn, dt in DT.items():
        for shape, g in [((16, 256), 128), ((8, 64), 32)]:
            w = weight_like(shape, dt, 800 + idx)
            qt, scales = INT4QTensor.quantize(w, g)
            deq = qt.dequantize(scale=scales, block_sizes={-1: g})
            cases[f"c{idx}"] = dict(dtype=dn, g=g, kind="qtensor")
            out[f"c{idx}_w"], out[f"c{idx}_q"] = bits(w), qt._quantized_data.numpy()
            out[f"c{idx}_s"], out[f"c{idx}_d"] = bits(scales), bits(deq)
            idx += 1
            # export packer: weights_scaling_factor = amax/7 fp32 [rows, cols/g]
            wsf = (w.reshape(shape[0], -1, g).abs().amax(-1).float() / 7.0)
            packed = pack_int4_in_uint8(w, wsf)
            cases[f"c{idx}"] = dict(dtype=dn, g=g, kind="export")
            out[f"c{idx}_w"], out[f"c{idx}_wsf"], out[f"c{idx}_p"] = bits(w), bits(wsf), packed.numpy()
            idx += 1
    out["cases"] = np.array(json.dumps(cases))


def gen_awq(out):
    """AWQ-lite building blocks, restated inline from model_calib.py:1453-1495 by CALLING the same torch
    ops the reference calls (the helpers are closures inside awq_lite and cannot be imported)."""
    import torch.nn.functional as F  # noqa: F401

    cases = {}
    idx = 0
    for dn, dt in [("bf16", torch.bfloat16), ("f16", torch.float16)]:
        w = weight_like((64, 256), dt, 900 + idx)
        g_ = torch.Generator().manual_seed(901 + idx)
        x = (torch.randn(96, 256, generator=g_) * torch.exp(torch.randn(256, generator=g_))).to(dt)
        g = 128
        # get_weight_scale (model_calib.py:1453-1469)
        wv = w.contiguous().view(-1, g)
        wa = wv.abs()
        scale = wa / (wa.amax(dim=1, keepdim=True) + torch.finfo(w.dtype).tiny)
        w_scale = scale.view(w.shape).mean(0).to(torch.float32)
        # get_act_scale (model_calib.py:1471-1472)
        x_scale = x.abs().contiguous().view(-1, x.shape[-1]).mean(0).to(torch.float32)
        out[f"c{idx}_w"], out[f"c{idx}_x"] = bits(w), bits(x)
        out[f"c{idx}_wscale"], out[f"c{idx}_xscale"] = bits(w_scale), bits(x_scale)
        alphas = [0.0, 0.3, 0.5, 1.0]
        for k, alpha in enumerate(alphas):
            # get_scale (model_calib.py:1474-1487)
            s = (x_scale.pow(alpha) / (w_scale.pow(1 - alpha) + torch.finfo(torch.float32).tiny)).clamp(
                min=1e-4, max=1e4).view(-1)
            s = (s / (s.max() * s.min()).sqrt()).view(-1)
            # search forward weight side (model_calib.py:1552-1554): weight_quantizer with pre_quant_scale
            q = TensorQuantizer(QuantizerAttributeConfig(num_bits=4, block_sizes={-1: g}))
            q.pre_quant_scale = s.to(w.dtype)
            wq = q(w)
            out[f"c{idx}_s{k}"], out[f"c{idx}_wq{k}"] = bits(s), bits(wq)
            # postprocess weight fold (model_calib.py:1208-1216): fp32 multiply then cast
            folded = (w * s.squeeze()[None, :]).to(w.dtype)
            out[f"c{idx}_fold{k}"] = bits(folded)
        cases[f"c{idx}"] = dict(dtype=dn, g=g, alphas=alphas)
        idx += 1
    out["cases"] = np.array(json.dumps(cases))


class _TinyMLP(torch.nn.Module):
    def __init__(self, d=128, h=128, dtype=torch.float32, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.fc1 = torch.nn.Linear(d, h, bias=False)
        self.fc2 = torch.nn.Linear(h, d, bias=True)
        with torch.no_grad():
            self.fc1.weight.copy_(torch.randn(h, d, generator=g) * 0.05)
            self.fc2.weight.copy_(torch.randn(d, h, generator=g) * 0.05)
            self.fc2.bias.copy_(torch.randn(d, generator=g) * 0.01)
        self.to(dtype)

    def forward(self, x):
        return self.fc2(torch.nn.functional.gelu(self.fc1(x)))


def _calib_batches(d, dtype, seed, n=4):
    g = torch.Generator().manual_seed(seed)
    ch = torch.exp(torch.randn(d, generator=g))
    ch[:4] *= 30  # a few massive channels so that the AWQ / SmoothQuant search is not degenerate
    return [(torch.randn(24, d, generator=g) * ch).to(dtype) for _ in range(n)]


def gen_model_flows(out):
    """mtq.quantize() on a tiny MLP: max (INT8 / FP8), smoothquant (INT8) and awq_lite (INT4 g128).
    Stores inputs, the reference's resulting buffers and one forward output."""
    import copy

    import modelopt.torch.quantization as mtq

    cases = {}
    flows = [("int8_max", mtq.INT8_DEFAULT_CFG, torch.float32), ("fp8_max", mtq.FP8_DEFAULT_CFG, torch.bfloat16),
             ("int8_sq", mtq.INT8_SMOOTHQUANT_CFG, torch.float32), ("int4_awq", mtq.INT4_AWQ_CFG, torch.float32),
             ("int4_awq_bf16", mtq.INT4_AWQ_CFG, torch.bfloat16)]
    for name, cfg, dt in flows:
        dn = {torch.float32: "f32", torch.bfloat16: "bf16"}[dt]
        model = _TinyMLP(dtype=dt, seed=3)
        batches = _calib_batches(128, dt, 5)
        # inputs are identical for every flow of one dtype: stored once under the dtype name
        out[f"{dn}_w1"], out[f"{dn}_w2"], out[f"{dn}_b2"] = bits(model.fc1.weight), bits(model.fc2.weight), bits(model.fc2.bias)
        for i, b in enumerate(batches):
            out[f"{dn}_x{i}"] = bits(b)

        def loop(m):
            for b in batches:
                m(b)

        cfg = copy.deepcopy(cfg)
        if isinstance(cfg.get("algorithm"), dict) and cfg["algorithm"].get("method") == "awq_lite":
            cfg["algorithm"]["debug"] = True  # keeps module.awq_lite (best_alpha, losses) after calibration
        q = mtq.quantize(copy.deepcopy(model), cfg, loop)
        info = dict(dtype=dn, n_batches=len(batches), tensors=[])
        for lname in ("fc1", "fc2"):
            lin = getattr(q, lname)
            out[f"{name}_{lname}_wfinal"] = bits(lin.weight)
            for qn in ("input_quantizer", "weight_quantizer"):
                tq_ = getattr(lin, qn)
                for attr in ("_amax", "_pre_quant_scale"):
                    if hasattr(tq_, attr):
                        t = getattr(tq_, attr)
                        out[f"{name}_{lname}_{qn}{attr}"] = bits(t.float())
                        info["tensors"].append([f"{lname}_{qn}{attr}", str(t.dtype), list(t.shape)])
            if hasattr(lin, "awq_lite"):
                h = lin.awq_lite
                info[f"{lname}_best_alpha"] = float(h.best_alpha)
                info[f"{lname}_loss"] = {str(k): float(v) for k, v in h.loss.items()}
                out[f"{name}_{lname}_act_scale"] = bits(h.act_scale.float())
                out[f"{name}_{lname}_weight_scale"] = bits(h.weight_scale.float())
                out[f"{name}_{lname}_best_scale"] = bits(h.best_scale.float())
        out[f"{name}_y"] = bits(q(batches[0]))
        cases[name] = info
    out["cases"] = np.array(json.dumps(cases))


def gen_awq_clip(out):
    """mtq.quantize() with algorithm awq_clip / awq_full (model_calib.py:1724-1940) on the tiny MLP, debug=True so
    that module.awq_clip (w_amax, per-shrink block losses, best_clip_val) survives.  fc1's inputs are the stored
    batches themselves, so its losses pin the block-search kernel directly."""
    import copy

    import modelopt.torch.quantization as mtq

    cases = {}
    for name, method, dt, d, ntok in [("clip_f32", "awq_clip", torch.float32, 128, 24),
                                      ("clip_bf16", "awq_clip", torch.bfloat16, 128, 24),
                                      ("clip_f16_200", "awq_clip", torch.float16, 256, 200),
                                      ("full_f32", "awq_full", torch.float32, 128, 24),
                                      ("full_bf16", "awq_full", torch.bfloat16, 128, 24)]:
        model = _TinyMLP(d=d, h=128, dtype=dt, seed=11)
        g = torch.Generator().manual_seed(17)
        ch = torch.exp(torch.randn(d, generator=g))
        ch[:4] *= 30
        batches = [(torch.randn(ntok, d, generator=g) * ch).to(dt) for _ in range(3)]
        out[f"{name}_w1"], out[f"{name}_w2"], out[f"{name}_b2"] = bits(model.fc1.weight), bits(model.fc2.weight), bits(model.fc2.bias)
        for i, b in enumerate(batches):
            out[f"{name}_x{i}"] = bits(b)

        def loop(m):
            for b in batches:
                m(b)

        cfg = copy.deepcopy(mtq.INT4_AWQ_CFG)
        cfg["algorithm"] = {"method": method, "debug": True}
        q = mtq.quantize(copy.deepcopy(model), cfg, loop)
        info = dict(dtype=str(dt).split(".")[-1], n_batches=len(batches), d=d, method=method)
        for lname in ("fc1", "fc2"):
            lin = getattr(q, lname)
            h = lin.awq_clip
            info[f"{lname}_shrinks"] = [float(k) for k in h.loss]
            info[f"{lname}_w_amax_dtype"] = str(h.w_amax.dtype).split(".")[-1]
            info[f"{lname}_num_tokens"] = int(h.num_tokens)
            out[f"{name}_{lname}_w_amax"] = bits(h.w_amax.float())
            out[f"{name}_{lname}_loss"] = bits(torch.stack([v.float() for v in h.loss.values()]))
            out[f"{name}_{lname}_best_clip_val"] = bits(h.best_clip_val.float())
            out[f"{name}_{lname}_wfinal"] = bits(lin.weight)
            wq = lin.weight_quantizer
            out[f"{name}_{lname}_amax_final"] = bits(wq._amax.float())
            info[f"{lname}_amax_final_dtype"] = str(wq._amax.dtype).split(".")[-1]
            info[f"{lname}_amax_final_shape"] = list(wq._amax.shape)
            if hasattr(lin.input_quantizer, "_pre_quant_scale"):
                out[f"{name}_{lname}_pre_quant_scale"] = bits(lin.input_quantizer._pre_quant_scale.float())
        out[f"{name}_y"] = bits(q(batches[0]))
        cases[name] = info
    out["cases"] = np.array(json.dumps(cases))


def gen_qtensor(out):
    """FP8QTensor / MXFP4QTensor real quantisation (qtensor/fp8_tensor.py:40-151, qtensor/mxfp4_tensor.py:37-144):
    quantized bytes, scales and the dequantised tensors the reference produces on CPU."""
    from modelopt.torch.quantization.qtensor import FP8QTensor, MXFP4QTensor

    cases = {}
    idx = 0
    for dn, dt in DT.items():
        w = weight_like((48, 256), dt, 1300 + idx)
        w[0, :8] = torch.tensor([0.0, -0.0, 1e-6, -1e-6, 0.5, -0.5, 3.0, -3.0], dtype=dt)
        # FP8: per-tensor, per-channel (axis 0), 1-D blocks of 128 along the last dim
        for mode, kw in [("tensor", {}), ("axis0", {"axis": 0}), ("block128", {"block_sizes": {-1: 128}})]:
            qt, scales = FP8QTensor.quantize(w, **kw)
            deq = qt.dequantize(scale=scales, **({"block_sizes": {-1: 128}} if mode == "block128" else {}))
            k = f"fp8_{dn}_{mode}"
            out[f"{k}_x"], out[f"{k}_q"] = bits(w), qt._quantized_data.view(torch.uint8).numpy().copy()
            out[f"{k}_scales"], out[f"{k}_deq"] = bits(scales), bits(deq)
            cases[k] = dict(kind="fp8", dtype=dn, mode=mode, scale_shape=list(scales.shape))
        for block in (32, 16):
            qt, e8 = MXFP4QTensor.quantize(w, block)
            deq = qt.dequantize(scale=e8, block_sizes={-1: block})
            k = f"mxfp4_{dn}_b{block}"
            out[f"{k}_x"], out[f"{k}_q"], out[f"{k}_e8m0"] = bits(w), qt._quantized_data.numpy().copy(), e8.numpy().copy()
            out[f"{k}_deq"] = bits(deq)
            cases[k] = dict(kind="mxfp4", dtype=dn, block=block)
        idx += 1
    out["cases"] = np.array(json.dumps(cases))


def gen_w4a8(out):
    """SequentialQuantizer flow: the W4A8 quantizer layout (INT4 g128 blocks then FP8 per tensor on the weights, FP8
    inputs; presets/model/w4a8_awq_beta.yaml) calibrated with algorithm "max" on the tiny MLP."""
    import copy

    import modelopt.torch.quantization as mtq

    cases = {}
    for name, dt in [("w4a8_f32", torch.float32), ("w4a8_bf16", torch.bfloat16)]:
        model = _TinyMLP(dtype=dt, seed=21)
        batches = _calib_batches(128, dt, 23)
        out[f"{name}_w1"], out[f"{name}_w2"], out[f"{name}_b2"] = bits(model.fc1.weight), bits(model.fc2.weight), bits(model.fc2.bias)
        for i, b in enumerate(batches):
            out[f"{name}_x{i}"] = bits(b)
        cfg = copy.deepcopy(mtq.W4A8_AWQ_BETA_CFG)
        cfg["algorithm"] = "max"
        q = mtq.quantize(copy.deepcopy(model), cfg, lambda m: [m(b) for b in batches])
        info = dict(dtype=str(dt).split(".")[-1], n_batches=len(batches))
        for lname in ("fc1", "fc2"):
            lin = getattr(q, lname)
            wq = lin.weight_quantizer
            info[f"{lname}_n_stages"] = len(wq)
            for i, st in enumerate(wq):
                out[f"{name}_{lname}_w{i}_amax"] = bits(st._amax.float())
                info[f"{lname}_w{i}_amax_shape"] = list(st._amax.shape)
            out[f"{name}_{lname}_in_amax"] = bits(lin.input_quantizer._amax.float())
            out[f"{name}_{lname}_wq"] = bits(wq(lin.weight))
        out[f"{name}_y"] = bits(q(batches[0]))
        cases[name] = info
    out["cases"] = np.array(json.dumps(cases))


def gen_sgpt(out):
    """SparseGPT (sparsity/weight_sparsity/sparsegpt.py): Hessian accumulated by the reference's forward hook over
    seeded activation batches, prepare() (damping + Cholesky inverse) and create_sgpt_mask, all run on CPU."""
    from modelopt.torch.sparsity.weight_sparsity import sparsegpt

    cases = {}
    for name, dt, co, ci, ntok in [("sgpt_f32", torch.float32, 64, 256, 96), ("sgpt_bf16", torch.bfloat16, 96, 384, 128)]:
        w = weight_like((co, ci), dt, 1700 + co)
        g = torch.Generator().manual_seed(1701 + ci)
        ch = torch.exp(torch.randn(ci, generator=g) * 0.7)
        batches = [(torch.randn(ntok, ci, generator=g) * ch).to(dt) for _ in range(3)]
        mod = type("Linear", (), {})()  # the hook looks at type(mod).__name__
        mod.hessian, mod.samples = torch.zeros(ci, ci, dtype=torch.float32), 0
        for b in batches:
            sparsegpt.SparseGPTSearcher._hook_compute_hessian(mod, (b.clone().unsqueeze(0),), None)
        cfg = {"pattern": "2:4 sparsity", "col_block_size": 128, "row_block_size": -1, "hessian_damp": 0.1}
        hessian = mod.hessian.clone()
        _, hinv = sparsegpt.prepare(w, hessian.clone(), cfg["hessian_damp"])
        mask = sparsegpt.create_sgpt_mask(w, hessian.clone(), cfg)
        out[f"{name}_w"] = bits(w)
        for i, b in enumerate(batches):
            out[f"{name}_x{i}"] = bits(b)
        out[f"{name}_hessian"], out[f"{name}_hinv"] = bits(hessian), bits(hinv)
        out[f"{name}_mask"] = mask.numpy().astype(np.uint8)
        cases[name] = dict(dtype=str(dt).split(".")[-1], n_batches=len(batches), samples=int(mod.samples),
                           kept=float(mask.float().mean()))
    out["cases"] = np.array(json.dumps(cases))


def gen_gptq(out):
    """GPTQ (quantization/model_calib.py:2192-2271, utils/calib_utils.py:50-276), run on CPU.
    Building blocks on one linear per weight format -- the steps of gptq() made one by one so that the intermediates can
    be stored: mtq.quantize(max) for the amax, GPTQHelper.setup + a forward loop with the weight quantizers off (the
    Hessian), update_weights (inverse factor, blockwise update) -- and mtq.quantize(algorithm = gptq) end to end on the
    tiny MLP."""
    import copy

    import modelopt.torch.quantization as mtq
    from modelopt.torch.quantization.conversion import set_quantizer_by_cfg_context
    from modelopt.torch.quantization.utils.calib_utils import GPTQHelper

    int4_g32 = copy.deepcopy(mtq.INT4_BLOCKWISE_WEIGHT_ONLY_CFG)
    for entry in (int4_g32["quant_cfg"] if isinstance(int4_g32["quant_cfg"], list) else []):
        if isinstance(entry, dict) and isinstance(entry.get("cfg"), dict) and "block_sizes" in entry["cfg"]:
            entry["cfg"]["block_sizes"] = {-1: 32}
    if isinstance(int4_g32["quant_cfg"], dict):
        for k, v in int4_g32["quant_cfg"].items():
            if isinstance(v, dict) and "block_sizes" in v:
                v["block_sizes"] = {-1: 32}
    cases = {}
    blocks = [("int4_g128_bf16", mtq.INT4_BLOCKWISE_WEIGHT_ONLY_CFG, torch.bfloat16, 48, 256, 128, 0.01),
              ("int4_g32_f32", int4_g32, torch.float32, 40, 192, 64, 0.01),
              ("fp8_bf16", mtq.FP8_DEFAULT_CFG, torch.bfloat16, 64, 256, 128, 0.01),
              ("int8_pc_f32", mtq.INT8_DEFAULT_CFG, torch.float32, 32, 160, 128, 0.1)]
    for name, cfg, dt, co, ci, block_size, perc_damp in blocks:
        lin = torch.nn.Linear(ci, co, bias=False)
        with torch.no_grad():
            lin.weight.copy_(weight_like((co, ci), torch.float32, 2300 + co + ci).float() * 4)
            if name == "int8_pc_f32":
                lin.weight[:, 5] = 0  # a dead input column (compute_hessian_inverse zeroes it out of the Hessian)
        lin = lin.to(dt)
        g = torch.Generator().manual_seed(2301 + ci)
        ch = torch.exp(torch.randn(ci, generator=g) * 0.7)
        batches = [(torch.randn(2, 40, ci, generator=g) * ch).to(dt) for _ in range(3)]

        def loop(m):
            for b in batches:
                m(b)

        w0 = lin.weight.detach().clone()
        q = mtq.quantize(lin, copy.deepcopy(cfg), loop)  # (converts in place; algorithm max)
        q.weight.data = w0.clone()
        helper = GPTQHelper(q, name, offload_to_cpu=False)
        helper.setup()
        with set_quantizer_by_cfg_context(q, [{"quantizer_name": "*weight_quantizer", "enable": False}]):
            loop(q)
        helper.cleanup()
        hessian = helper.hessian.clone()
        n_samples = int(helper.n_samples)
        helper.update_weights(block_size, perc_damp)
        out[f"{name}_w"] = bits(w0)
        for i, b in enumerate(batches):
            out[f"{name}_x{i}"] = bits(b)
        out[f"{name}_hessian"], out[f"{name}_hinv"] = bits(hessian), bits(helper.h_inv)
        out[f"{name}_wfinal"] = bits(q.weight.data)
        out[f"{name}_w_amax"] = bits(q.weight_quantizer._amax.float())
        if q.input_quantizer.is_enabled and hasattr(q.input_quantizer, "_amax"):
            out[f"{name}_in_amax"] = bits(q.input_quantizer._amax.float())
        cases[name] = dict(dtype=str(dt).split(".")[-1], n_batches=len(batches), n_samples=n_samples, block_size=block_size,
                           perc_damp=perc_damp, w_amax_shape=list(q.weight_quantizer._amax.shape),
                           input_quantizer=bool(q.input_quantizer.is_enabled))
    # end to end: mtq.quantize with the gptq algorithm (not layerwise) on the tiny MLP
    for name, cfg, dt in [("flow_int4_f32", mtq.INT4_BLOCKWISE_WEIGHT_ONLY_CFG, torch.float32),
                          ("flow_fp8_bf16", mtq.FP8_DEFAULT_CFG, torch.bfloat16)]:
        model = _TinyMLP(dtype=dt, seed=31)
        batches = _calib_batches(128, dt, 32)
        out[f"{name}_w1"], out[f"{name}_w2"], out[f"{name}_b2"] = bits(model.fc1.weight), bits(model.fc2.weight), bits(model.fc2.bias)
        for i, b in enumerate(batches):
            out[f"{name}_x{i}"] = bits(b)

        def loop2(m):
            for b in batches:
                m(b)

        cfg = copy.deepcopy(cfg)
        cfg["algorithm"] = {"method": "gptq", "perc_damp": 0.01, "block_size": 128}
        qm = mtq.quantize(copy.deepcopy(model), cfg, loop2)
        for lname in ("fc1", "fc2"):
            out[f"{name}_{lname}_wfinal"] = bits(getattr(qm, lname).weight)
            out[f"{name}_{lname}_w_amax"] = bits(getattr(qm, lname).weight_quantizer._amax.float())
        out[f"{name}_y"] = bits(qm(batches[0]))
        cases[name] = dict(dtype=str(dt).split(".")[-1], n_batches=len(batches))
    out["cases"] = np.array(json.dumps(cases))


def gen_gptq_llama(out):
    """mtq.quantize(tiny Llama, INT4 blockwise weight-only, algorithm = gptq) in one pass over the whole model: original
    weights, tokens, every updated weight, the logits.
    (Not stored: the layer-by-layer mode, layerwise.enable.  With the transformers version of this container (5.15,
    "not tested" by the reference's own warning) the reference's replay of a decoder layer returns an all-zero attention
    output -- o_proj's Hessian is the zero matrix, "not positive definite, using identity", and the MLP's Hessian is that
    of the layer input -- so that run is not an oracle for anything.)"""
    import copy

    import modelopt.torch.quantization as mtq
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfgd = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, vocab_size=128, max_position_embeddings=64)
    cfg = LlamaConfig(architectures=["LlamaForCausalLM"], **cfgd)
    base = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    batches = [torch.randint(0, 128, (4, 32), generator=torch.Generator().manual_seed(40 + i)) for i in range(3)]
    for k, v in base.state_dict().items():
        out[f"orig/{k}"] = bits(v)
    for i, b in enumerate(batches):
        out[f"tokens{i}"] = b.numpy()
    cases = {"config": cfgd, "n_batches": len(batches), "runs": {}}
    for run, layerwise in (("whole", {"enable": False}),):
        qcfg = copy.deepcopy(mtq.INT4_BLOCKWISE_WEIGHT_ONLY_CFG)
        qcfg["algorithm"] = {"method": "gptq", "perc_damp": 0.01, "block_size": 128, "layerwise": layerwise}
        q = mtq.quantize(copy.deepcopy(base), qcfg, lambda m: [m(b) for b in batches])
        names = []
        for n, m in q.named_modules():
            if hasattr(m, "weight_quantizer") and m.weight_quantizer.is_enabled:
                names.append(n)
                out[f"{run}/{n}.weight"] = bits(m.weight)
                out[f"{run}/{n}.amax"] = bits(m.weight_quantizer._amax.float())
        with torch.no_grad():
            out[f"{run}/logits"] = bits(q(batches[0]).logits)
        cases["runs"][run] = names
    out["cases"] = np.array(json.dumps(cases))


def gen_local_hessian(out):
    """mtq.quantize(tiny MLP, INT4 blocks of 16, algorithm = local_hessian with the multiplier search) -- the refined amax of
    both linears, next to the max-calibrated one and to what plain `mse` picks (model_calib.py:1005-1127)."""
    import copy

    import modelopt.torch.quantization as mtq

    cases = {}
    for name, dt in (("lh_f32", torch.float32), ("lh_bf16", torch.bfloat16)):
        model = _TinyMLP(dtype=dt, seed=41)
        batches = _calib_batches(128, dt, 42)
        out[f"{name}_w1"], out[f"{name}_w2"], out[f"{name}_b2"] = bits(model.fc1.weight), bits(model.fc2.weight), bits(model.fc2.bias)
        for i, b in enumerate(batches):
            out[f"{name}_x{i}"] = bits(b)
        for alg_name, alg in (("max", "max"), ("mse", {"method": "mse"}),
                              ("local_hessian", {"method": "local_hessian", "fp8_scale_sweep": False, "block_size": 16})):
            cfg = copy.deepcopy(mtq.INT4_BLOCKWISE_WEIGHT_ONLY_CFG)
            for entry in (cfg["quant_cfg"] if isinstance(cfg["quant_cfg"], list) else []):
                if isinstance(entry, dict) and isinstance(entry.get("cfg"), dict) and "block_sizes" in entry["cfg"]:
                    entry["cfg"]["block_sizes"] = {-1: 16}
            if isinstance(cfg["quant_cfg"], dict):
                for v in cfg["quant_cfg"].values():
                    if isinstance(v, dict) and "block_sizes" in v:
                        v["block_sizes"] = {-1: 16}
            cfg["algorithm"] = alg
            q = mtq.quantize(copy.deepcopy(model), cfg, lambda m: [m(b) for b in batches])
            for lname in ("fc1", "fc2"):
                out[f"{name}_{alg_name}_{lname}_amax"] = bits(getattr(q, lname).weight_quantizer._amax.float())
        cases[name] = dict(dtype=str(dt).split(".")[-1], n_batches=len(batches))
    out["cases"] = np.array(json.dumps(cases))


def gen_affine_bias(out):
    """TensorQuantizer(bias=...) on key / value-shaped states [batch, heads, tokens, head_dim] (tensor_quantizer.py:389-503,
    :1397-1407; calib/bias.py): three calibration batches with an off-centre distribution per head and channel, then the
    fake-quantized first batch.  Static offsets are calibrated before the abs-max (which sees the centred tensor); dynamic
    ones are recomputed per call and the amax is that of the tensor as it came."""
    cases = {}
    specs = [("fp8_static_mean_head", (4, 3), {-2: None, -4: None, "type": "static"}),
             ("fp8_static_maxmin_head", (4, 3), {-2: None, -4: None, "type": "static", "method": "max_min"}),
             ("int8_static_mean_tensor", 8, {-1: None, -2: None, -3: None, -4: None, "type": "static"}),
             ("int8_static_mean_channel", 8, {-2: None, -3: None, -4: None}),
             ("fp8_dynamic_mean_head", (4, 3), {-2: None, -4: None, "type": "dynamic"}),
             ("int8_dynamic_maxmin_token", 8, {-1: None, "type": "dynamic", "method": "max_min"})]
    for dt_name in ("f32", "bf16"):
        dt = DT[dt_name]
        g = torch.Generator().manual_seed(77)
        centre = torch.randn(1, 3, 1, 16, generator=g) * 2.0
        xs = [((torch.randn(2, 3, 10, 16, generator=g) * (0.5 + i) + centre)).to(dt) for i in range(3)]
        for i, x in enumerate(xs):
            out[f"{dt_name}_x{i}"] = bits(x)
        for name, nb, bias in specs:
            q = TensorQuantizer(QuantizerAttributeConfig(num_bits=nb, axis=None, bias=dict(bias)))
            static = bias.get("type", "static") == "static"
            q.disable_quant()
            q.enable_calib()
            for x in xs:
                q(x)
            q.load_calib_amax()
            if static:
                q.load_calib_bias()
            q.enable_quant()
            q.disable_calib()
            key = f"{dt_name}_{name}"
            out[f"{key}_amax"] = bits(q._amax.float())
            if static:
                out[f"{key}_bias"] = bits(q._bias_value)
            out[f"{key}_y0"] = bits(q(xs[0]))
            cases[key] = dict(dtype=dt_name, num_bits=list(nb) if isinstance(nb, tuple) else nb,
                              bias={str(k): v for k, v in bias.items()}, static=static,
                              bias_shape=list(q._bias_value.shape) if static else None)
    out["cases"] = np.array(json.dumps(cases))


def gen_block2d(out):
    """TensorQuantizer with block_sizes on BOTH axes (FP8 128 x 128 tiles, INT8 64 x 32 tiles): calibrated amax
    (reduce over the (R/br, br, C/bc, bc) view, tensor_quantizer.py:1018-1043) and the eager fake-quant output."""
    cases = {}
    idx = 0
    for dn, dt in DT.items():
        for nb, br, bc, shape in [((4, 3), 128, 128, (256, 384)), (8, 64, 32, (128, 96))]:
            w = weight_like(shape, dt, 1900 + idx)
            q = TensorQuantizer(QuantizerAttributeConfig(num_bits=nb, block_sizes={-1: bc, -2: br}))
            q.disable_quant(); q.enable_calib()
            q(w)
            q.load_calib_amax()
            q.enable_quant(); q.disable_calib()
            y = q(w)
            k = f"c{idx}"
            out[f"{k}_x"], out[f"{k}_y"], out[f"{k}_amax"] = bits(w), bits(y), bits(q._amax.float())
            cases[k] = dict(dtype=dn, num_bits=list(nb) if isinstance(nb, tuple) else nb, br=br, bc=bc,
                            amax_shape=list(q._amax.shape), amax_dtype=str(q._amax.dtype).split(".")[-1])
            idx += 1
    # the same tiles on the last two axes of tensors of rank 3 / 4 (stacked experts, leading batch dims): whole tiles
    # and ragged matrices, calibrated and dynamic amax (tensor_quantizer.py:1018-1043)
    nidx = 0
    for dn, dt in DT.items():
        for nb, br, bc, shape in [((4, 3), 16, 32, (3, 32, 64)), (8, 8, 16, (2, 3, 20, 40)), ((4, 3), 16, 16, (4, 30, 50))]:
            w = weight_like(shape, dt, 2100 + nidx)
            q = TensorQuantizer(QuantizerAttributeConfig(num_bits=nb, block_sizes={-1: bc, -2: br}))
            q.disable_quant(); q.enable_calib()
            q(w)
            q.load_calib_amax()
            q.enable_quant(); q.disable_calib()
            k = f"n{nidx}"
            out[f"{k}_x"], out[f"{k}_y"], out[f"{k}_amax"] = bits(w), bits(q(w)), bits(q._amax.float())
            out[f"{k}_ydyn"] = bits(TensorQuantizer(QuantizerAttributeConfig(num_bits=nb, block_sizes={-1: bc, -2: br}))(w))
            cases[k] = dict(dtype=dn, num_bits=list(nb) if isinstance(nb, tuple) else nb, br=br, bc=bc, lead=list(shape[:-2]),
                            shape=list(shape), amax_shape=list(q._amax.shape), amax_dtype=str(q._amax.dtype).split(".")[-1])
            nidx += 1
    # grids on other axis sets (rows only, a conv weight's input channels, three axes at once; ragged sizes): two
    # calibration calls (running maximum), then the output; and the dynamic-amax output
    gidx = 0
    for dn, dt in DT.items():
        for nb, grid, shape in [((4, 3), {0: 8}, (32, 24)), (8, {0: 4, -1: 8}, (6, 5, 20)), ((4, 3), {1: 4}, (6, 10, 3, 3)),
                                (8, {0: 2, 1: 3, 2: 4}, (4, 7, 8)), (4, {-2: 16}, (40, 16))]:
            w = weight_like(shape, dt, 2300 + gidx)
            q = TensorQuantizer(QuantizerAttributeConfig(num_bits=nb, block_sizes=dict(grid)))
            q.disable_quant(); q.enable_calib()
            q(w); q(w * 0.5)
            q.load_calib_amax()
            q.enable_quant(); q.disable_calib()
            k = f"g{gidx}"
            out[f"{k}_x"], out[f"{k}_y"], out[f"{k}_amax"] = bits(w), bits(q(w)), bits(q._amax.float())
            out[f"{k}_ydyn"] = bits(TensorQuantizer(QuantizerAttributeConfig(num_bits=nb, block_sizes=dict(grid)))(w))
            cases[k] = dict(dtype=dn, num_bits=list(nb) if isinstance(nb, tuple) else nb, grid={str(a): b for a, b in grid.items()},
                            lead=[], shape=list(shape), amax_shape=list(q._amax.shape), amax_dtype=str(q._amax.dtype).split(".")[-1])
            gidx += 1
    out["cases"] = np.array(json.dumps(cases))


def extract_mx_vectors():
    """Pull the literal test_in / test_out tables out of the reference's MX test (no execution)."""
    path = os.path.join(ref_shim.REFERENCE_ROOT, "tests/gpu/torch/quantization/test_quantize_mxformats_cuda.py")
    tree = ast.parse(open(path).read())
    fmt_names = {"8": "INT8", "(2, 1)": "E2M1", "(3, 2)": "E3M2", "(2, 3)": "E2M3", "(4, 3)": "E4M3",
                 "(5, 2)": "E5M2"}
    cases = []
    for fn in tree.body:
        if not isinstance(fn, ast.FunctionDef) or fn.name not in ("test_mxfp4", "test_mxfp6", "test_mxfp8",
                                                                 "test_mxint8"):
            continue
        state = {"block_size": None, "in_size": None, "test_in": None, "test_out": None, "dtype": None}
        pending = None
        for node in fn.body:
            if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name):
                name = node.targets[0].id
                if name in ("test_in", "test_out"):
                    state[name] = ast.literal_eval(node.value.args[0])
                elif name == "block_size":
                    state["block_size"] = ast.literal_eval(node.value)
                elif name == "dtype":
                    state["dtype"] = ast.unparse(node.value)
                elif name == "outputs":
                    key = None
                    for sub in ast.walk(node.value):
                        if isinstance(sub, ast.Subscript) and getattr(sub.value, "id", "") == "mx_format_map":
                            key = ast.unparse(sub.slice)
                            break
                    pending = dict(fn=fn.name, fmt=fmt_names[key], block_size=state["block_size"],
                                   in_size=state["in_size"], test_in=state["test_in"],
                                   test_out=state["test_out"], dtype=state["dtype"])
                elif isinstance(node.targets[0], ast.Name) and name == "sign":
                    pass
            if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Tuple):
                # inputs, expected_outputs = _get_test_inputs_outputs(..., block_size, IN_SIZE)
                state["in_size"] = ast.literal_eval(node.value.args[3])
            if isinstance(node, ast.Assert) and pending is not None:
                atol = 1e-8
                for kw in node.test.keywords:
                    if kw.arg == "atol":
                        atol = ast.literal_eval(kw.value)
                pending["atol"] = atol
                cases.append(pending)
                pending = None
    return cases


def gen_mse(out):
    """MseCalibrator (calib/mse.py) driven the way mse_calibrate drives it (model_calib.py:639-826):
    per-candidate losses and the chosen amax for INT8 per-tensor / per-channel, INT4 static blocks (with
    padding) and FP8 per-tensor quantizers, plus mtq.quantize(..., algorithm='mse') on the tiny MLP."""
    import copy
    from functools import partial

    import modelopt.torch.quantization as mtq
    from modelopt.torch.quantization.calib import MseCalibrator
    from modelopt.torch.quantization.model_calib import _mse_quant_func, max_calibrate

    cases = {}
    specs = [
        ("int8_tensor_f32", dict(num_bits=8, axis=None), (48, 200), "f32"),
        ("int8_chan_bf16", dict(num_bits=8, axis=0), (40, 264), "bf16"),
        ("int4_chan_f16", dict(num_bits=4, axis=0), (16, 4100), "f16"),
        ("int4_block128_bf16", dict(num_bits=4, block_sizes={-1: 128, "type": "static"}), (24, 512), "bf16"),
        ("int4_block32_pad_f32", dict(num_bits=4, block_sizes={-1: 32, "type": "static"}), (10, 72), "f32"),
        ("fp8_tensor_bf16", dict(num_bits=(4, 3), axis=None), (64, 320), "bf16"),
        ("int8_tensor_big_bf16", dict(num_bits=8, axis=None), (96, 1100), "bf16"),
    ]
    for i, (name, cfg, shape, dn) in enumerate(specs):
        w = weight_like(shape, DT[dn], 500 + i)
        q = TensorQuantizer(QuantizerAttributeConfig(**cfg))
        max_calibrate(q, lambda qq: qq(w), distributed_sync=False)
        init = q._amax.clone().detach()
        cal = MseCalibrator(amax=init, axis=q._calibrator._axis, step_size=0.1, start_multiplier=0.25,
                            stop_multiplier=4.0, quant_func=partial(_mse_quant_func, quantizer=q))
        q._calibrator = cal
        q.disable_quant(); q.enable_calib()
        q(w)
        losses = torch.stack([l.reshape(-1) for l in cal._losses_sum]).float()
        amax = cal.compute_amax()
        out[f"{name}_w"] = bits(w)
        out[f"{name}_init_amax"] = bits(init.float())
        out[f"{name}_losses"] = bits(losses)
        out[f"{name}_amax"] = bits(amax.float())
        cases[name] = dict(cfg={k: (list(v) if isinstance(v, tuple) else ({str(a): b for a, b in v.items()} if isinstance(v, dict) else v))
                                for k, v in cfg.items()}, dtype=dn, shape=list(shape),
                           init_dtype=str(init.dtype), init_shape=list(init.shape), amax_dtype=str(amax.dtype),
                           amax_shape=list(amax.shape))
    # model flow
    for name, base, dt in [("flow_int8_mse", mtq.INT8_DEFAULT_CFG, torch.float32),
                           ("flow_int4blk_mse_bf16", mtq.INT4_BLOCKWISE_WEIGHT_ONLY_CFG, torch.bfloat16)]:
        dn = {torch.float32: "f32", torch.bfloat16: "bf16"}[dt]
        model = _TinyMLP(dtype=dt, seed=3)
        batches = _calib_batches(128, dt, 5)
        cfg = copy.deepcopy(base)
        cfg["algorithm"] = {"method": "mse"}
        q = mtq.quantize(copy.deepcopy(model), cfg, lambda m: [m(b) for b in batches])
        info = dict(dtype=dn, n_batches=len(batches))
        for lname in ("fc1", "fc2"):
            a = getattr(q, lname).weight_quantizer._amax
            out[f"{name}_{lname}_weight_amax"] = bits(a.float())
            info[f"{lname}_amax_shape"] = list(a.shape)
        cases[name] = info
    out["cases"] = np.array(json.dumps(cases))


def gen_export(out, capture=None, preset="INT4_AWQ_CFG"):
    """INT4-AWQ checkpoint export of a tiny bf16 Llama by the reference (mtq.quantize(INT4_AWQ_CFG) +
    export_hf_checkpoint, export/unified_export_hf.py:1491): the original weights + calibration tokens (for the
    end-to-end test), the calibrated state right before export (folded weights, pre_quant_scales, per-block amax,
    norm weights) and every tensor of the exported model.safetensors.  `capture` (a dict) additionally receives what
    gen_export_replay stores: every linear's input per forward call and the search's intermediate statistics."""
    import tempfile

    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint
    from safetensors import safe_open
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfgd = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, vocab_size=128, max_position_embeddings=64)
    cfg = LlamaConfig(architectures=["LlamaForCausalLM"], **cfgd)
    model = LlamaForCausalLM(cfg).to(torch.bfloat16)
    batches = [torch.randint(0, 128, (4, 32), generator=torch.Generator().manual_seed(10 + i)) for i in range(3)]
    for k, v in model.state_dict().items():
        out[f"orig/{k}"] = bits(v)
    for i, b in enumerate(batches):
        out[f"tokens{i}"] = b.numpy()
    calls = {}
    if capture is not None:
        for n, m in model.named_modules():
            if isinstance(m, torch.nn.Linear):
                m.register_forward_pre_hook(lambda mod, args, n=n: calls.setdefault(n, []).append(args[0].detach().clone()))
    import copy as _copy
    awq_cfg = _copy.deepcopy(getattr(mtq, preset))
    if isinstance(awq_cfg["algorithm"], str):
        awq_cfg["algorithm"] = {"method": awq_cfg["algorithm"]}
    awq_cfg["algorithm"]["debug"] = True  # keeps module.awq_lite (best_alpha) after calibration
    q = mtq.quantize(model, awq_cfg, lambda m: [m(b) for b in batches])
    linears = []
    for n, m in q.named_modules():
        if hasattr(m, "weight_quantizer") and m.weight_quantizer.is_enabled:
            linears.append(n)
            out[f"pre/{n}.weight"] = bits(m.weight)
            stages = list(m.weight_quantizer) if preset != "INT4_AWQ_CFG" else [m.weight_quantizer]
            out[f"pre/{n}.amax"] = bits(stages[0]._amax)
            if len(stages) > 1:  # W4A8: the FP8 stage's per-tensor amax and the collapsed input amax
                out[f"pre/{n}.amax2"] = bits(stages[1]._amax.float())
                out[f"pre/{n}.in_amax"] = bits(m.input_quantizer._amax.float())
                out[f"pre/{n}.in_amax_channels"] = bits(m.input_quantizer._amax_for_smoothing.float())
            out[f"pre/{n}.pre_quant_scale"] = bits(m.input_quantizer._pre_quant_scale)
            out[f"pre/{n}.best_alpha"] = np.array(float(m.awq_lite.best_alpha) if hasattr(m, "awq_lite") else -1.0)
            if capture is not None:
                h = m.awq_lite
                capture[f"ref/{n}.act_scale"] = bits(h.act_scale)
                capture[f"ref/{n}.weight_scale"] = bits(h.weight_scale)
                capture[f"ref/{n}.best_scale"] = bits(h.best_scale)
                capture[f"ref/{n}.loss"] = np.array([float(v) for v in h.loss.values()], dtype=np.float64)
                seen = calls[n]
                # cache pass and search pass feed every linear the same tensors (nothing quantizes in between)
                assert len(seen) == 2 * len(batches), (n, len(seen))
                for b in range(len(batches)):
                    assert torch.equal(seen[b], seen[b + len(batches)]), (n, b)
                    capture[f"in/{n}/{b}"] = seen[b]
        elif type(m).__name__.endswith("RMSNorm"):
            out[f"pre/{n}.weight"] = bits(m.weight)
    with tempfile.TemporaryDirectory() as d:
        export_hf_checkpoint(q, export_dir=d)
        dtypes = {}
        with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
            for k in f.keys():
                t = f.get_tensor(k)
                out[f"exp/{k}"] = bits(t)
                dtypes[k] = str(t.dtype)
        quant_cfg = json.load(open(os.path.join(d, "hf_quant_config.json")))
    out["cases"] = np.array(json.dumps(dict(config=cfgd, n_batches=len(batches), linears=linears, dtypes=dtypes,
                                             hf_quant_config=quant_cfg)))


def gen_export_w4a8(out):
    """W4A8_AWQ_BETA_CFG (INT4 blocks -> FP8 weights, FP8 inputs, awq_lite) on the SAME model, tokens and -- asserted
    here -- the same per-linear inputs as export_llama.npz / export_llama_replay.npz: the input quantizers are bypassed
    in both calibration passes (model_calib.py:1436-1444), so the replay data serves this run too.  Stored: what differs
    from the INT4 run -- the exported tensors, the quantizer state right before export and hf_quant_config."""
    full, cap = {}, {}
    gen_export(full, capture=cap, preset="W4A8_AWQ_BETA_CFG")
    base = np.load(os.path.join(HERE, "export_llama.npz"))
    replay = np.load(os.path.join(HERE, "export_llama_replay.npz"))
    rc = json.loads(str(replay["cases"]))
    for k in base.files:
        if k.startswith(("orig/", "tokens")):
            assert np.array_equal(base[k], full[k]), k
    for n in rc["linears"]:
        for b in range(rc["n_batches"]):
            owner = rc["alias"].get(n, n)
            assert np.array_equal(replay[f"in/{owner}/{b}"], bits(cap[f"in/{n}/{b}"])), (n, b)
        assert np.array_equal(replay[f"ref/{n}.act_scale"], cap[f"ref/{n}.act_scale"]), n
        assert float(full[f"pre/{n}.best_alpha"]) == float(base[f"pre/{n}.best_alpha"]), n
    for k, v in full.items():
        if k.startswith("exp/") or k == "cases" or k.endswith((".amax2", ".in_amax", ".in_amax_channels")):
            out[k] = v


def gen_awq_ragged(out):
    """INT4_AWQ_CFG by the reference on ONE bf16 linear whose input width (192) is not a multiple of the INT4 block
    (128): the last block of every row is zero-padded (get_weight_scale, model_calib.py:1453-1469; static block quantizer,
    tensor_quantizer.py:975-1043).  The linear is fed directly (no model GEMM upstream), so alpha, pre_quant_scale, the
    folded weight, the per-block amax and the fake-quantized output are reproducible bit for bit."""
    import copy as _copy

    import modelopt.torch.quantization as mtq

    class One(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc = torch.nn.Linear(192, 96, bias=False)

        def forward(self, x):
            return self.fc(x)

    model = One()
    with torch.no_grad():
        model.fc.weight.copy_(weight_like((96, 192), torch.float32, 4242).float())
    model = model.to(torch.bfloat16)
    batches = _calib_batches(192, torch.bfloat16, 4243, n=3)
    out["w"] = bits(model.fc.weight)
    for i, b in enumerate(batches):
        out[f"x{i}"] = bits(b)
    cfg = _copy.deepcopy(mtq.INT4_AWQ_CFG)
    cfg["algorithm"]["debug"] = True
    q = mtq.quantize(model, cfg, lambda m: [m(b) for b in batches])
    h = q.fc.awq_lite
    out["best_alpha"] = np.array(float(h.best_alpha))
    out["loss"] = np.array([float(v) for v in h.loss.values()], dtype=np.float64)
    out["weight_scale"], out["act_scale"] = bits(h.weight_scale), bits(h.act_scale)
    out["pre_quant_scale"] = bits(q.fc.input_quantizer._pre_quant_scale)
    out["folded"] = bits(q.fc.weight)
    out["amax"] = bits(q.fc.weight_quantizer._amax.float())
    with torch.no_grad():
        out["y0"] = bits(q(batches[0]))
    out["cases"] = np.array(json.dumps(dict(n_batches=len(batches), amax_shape=list(q.fc.weight_quantizer._amax.shape))))


def gen_export_replay(out):
    """Replay data for the end-to-end INT4-AWQ checkpoint test: the SAME reference run as export_llama.npz (checked
    array by array against that file), plus what every quantized linear received in each calibration batch and the
    search's intermediate statistics (act_scale, weight_scale, per-alpha losses, best_scale).  Feeding these inputs to
    the patched linears takes the model's own GEMMs (attention, MLP: library kernels whose summation order differs
    between machines) out of the comparison: everything the path computes from them must then equal the reference
    byte for byte.  Linears that read one tensor (q / k / v, gate / up) store it once (`alias`)."""
    full, cap = {}, {}
    gen_export(full, capture=cap)
    have = np.load(os.path.join(HERE, "export_llama.npz"))
    assert set(have.files) == set(full), "export_llama.npz is not from this run"
    for k in have.files:
        assert np.array_equal(have[k], full[k]), f"export_llama.npz[{k}] differs from this run"
    cases = json.loads(str(full["cases"]))
    alias, stored = {}, {}
    for n in cases["linears"]:
        for b in range(cases["n_batches"]):
            x = cap.pop(f"in/{n}/{b}")
            owner = next((o for o, t in stored.items() if o[1] == b and t.shape == x.shape and torch.equal(t, x)), None)
            if owner is None:
                stored[(n, b)] = x
                out[f"in/{n}/{b}"] = bits(x)
                out[f"in_shape/{n}/{b}"] = np.array(x.shape)
            else:
                assert alias.setdefault(n, owner[0]) == owner[0]
    for k, v in cap.items():
        out[k] = v
    out["cases"] = np.array(json.dumps(dict(alias=alias, linears=cases["linears"], n_batches=cases["n_batches"])))


def gen_export_fp8(out):
    """FP8 (per-tensor W + A, FP8_DEFAULT_CFG, max calibration) checkpoint export of the tiny bf16 Llama by the
    reference: original weights + tokens, calibrated amax of every quantizer right before export, every exported
    tensor of model.safetensors and hf_quant_config.json."""
    import tempfile

    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint
    from safetensors import safe_open
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfgd = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, vocab_size=128, max_position_embeddings=64)
    cfg = LlamaConfig(architectures=["LlamaForCausalLM"], **cfgd)
    model = LlamaForCausalLM(cfg).to(torch.bfloat16)
    batches = [torch.randint(0, 128, (4, 32), generator=torch.Generator().manual_seed(10 + i)) for i in range(3)]
    for k, v in model.state_dict().items():
        out[f"orig/{k}"] = bits(v)
    for i, b in enumerate(batches):
        out[f"tokens{i}"] = b.numpy()
    q = mtq.quantize(model, mtq.FP8_DEFAULT_CFG, lambda m: [m(b) for b in batches])
    linears = []
    for n, m in q.named_modules():
        if hasattr(m, "weight_quantizer") and m.weight_quantizer.is_enabled:
            linears.append(n)
            out[f"pre/{n}.w_amax"] = bits(m.weight_quantizer._amax.float())
            out[f"pre/{n}.in_amax"] = bits(m.input_quantizer._amax.float())
    with tempfile.TemporaryDirectory() as d:
        export_hf_checkpoint(q, export_dir=d)
        dtypes = {}
        with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
            for k in f.keys():
                t = f.get_tensor(k)
                out[f"exp/{k}"] = t.view(torch.uint8).numpy().copy() if t.dtype == torch.float8_e4m3fn else bits(t)
                dtypes[k] = str(t.dtype)
        quant_cfg = json.load(open(os.path.join(d, "hf_quant_config.json")))
    out["cases"] = np.array(json.dumps(dict(config=cfgd, n_batches=len(batches), linears=linears, dtypes=dtypes,
                                             hf_quant_config=quant_cfg)))


def gen_export_fp8_pc_pt(out):
    """FP8 per-channel weights + per-token dynamic inputs (FP8_PER_CHANNEL_PER_TOKEN_CFG, max calibration) checkpoint
    export of the tiny bf16 Llama by the reference: original weights + tokens, the per-channel weight amax right before
    export, every exported tensor of model.safetensors (E4M3 weights, fp32 [Cout] weight_scale, no input_scale) and
    hf_quant_config.json (quant_algo FP8_PER_CHANNEL_PER_TOKEN)."""
    import tempfile

    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint
    from safetensors import safe_open
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfgd = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, vocab_size=128, max_position_embeddings=64)
    cfg = LlamaConfig(architectures=["LlamaForCausalLM"], **cfgd)
    model = LlamaForCausalLM(cfg).to(torch.bfloat16)
    batches = [torch.randint(0, 128, (4, 32), generator=torch.Generator().manual_seed(10 + i)) for i in range(3)]
    # same seed, same config: the original weights and tokens ARE export_llama_fp8.npz's (checked, not stored twice)
    base = np.load(os.path.join(HERE, "export_llama_fp8.npz"), allow_pickle=False)
    for k, v in model.state_dict().items():
        assert np.array_equal(base[f"orig/{k}"], bits(v)), k
    for i, b in enumerate(batches):
        assert np.array_equal(base[f"tokens{i}"], b.numpy())
    q = mtq.quantize(model, mtq.FP8_PER_CHANNEL_PER_TOKEN_CFG, lambda m: [m(b) for b in batches])
    linears = []
    for n, m in q.named_modules():
        if hasattr(m, "weight_quantizer") and m.weight_quantizer.is_enabled:
            linears.append(n)
            out[f"pre/{n}.w_amax"] = bits(m.weight_quantizer._amax.float())
            assert getattr(m.input_quantizer, "_amax", None) is None  # dynamic per token: nothing calibrated
    with tempfile.TemporaryDirectory() as d:
        export_hf_checkpoint(q, export_dir=d)
        dtypes = {}
        with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
            for k in f.keys():
                t = f.get_tensor(k)
                out[f"exp/{k}"] = t.view(torch.uint8).numpy().copy() if t.dtype == torch.float8_e4m3fn else bits(t)
                dtypes[k] = str(t.dtype)
        quant_cfg = json.load(open(os.path.join(d, "hf_quant_config.json")))
    out["cases"] = np.array(json.dumps(dict(config=cfgd, n_batches=len(batches), linears=linears, dtypes=dtypes,
                                             hf_quant_config=quant_cfg)))


def gen_export_mxfp4(out):
    """MXFP4 (dynamic blocks of 32, E8M0 scales; MXFP4_DEFAULT_CFG, no calibration) export of the tiny bf16 Llama by
    the reference: original weights and every exported tensor (packed nibbles, E8M0 scale bytes)."""
    import tempfile

    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint
    from safetensors import safe_open
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfgd = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, vocab_size=128, max_position_embeddings=64)
    model = LlamaForCausalLM(LlamaConfig(architectures=["LlamaForCausalLM"], **cfgd)).to(torch.bfloat16)
    for k, v in model.state_dict().items():
        out[f"orig/{k}"] = bits(v)
    q = mtq.quantize(model, mtq.MXFP4_DEFAULT_CFG, None)
    with tempfile.TemporaryDirectory() as d:
        export_hf_checkpoint(q, export_dir=d)
        dtypes = {}
        with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
            for k in f.keys():
                t = f.get_tensor(k)
                out[f"exp/{k}"] = bits(t)
                dtypes[k] = str(t.dtype)
        quant_cfg = json.load(open(os.path.join(d, "hf_quant_config.json")))
    out["cases"] = np.array(json.dumps(dict(config=cfgd, dtypes=dtypes, hf_quant_config=quant_cfg)))


def _gen_export_uncalibrated(out, preset):
    """export_hf_checkpoint of the tiny bf16 Llama under a preset that needs no calibration (dynamic MX weights)."""
    import tempfile

    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint
    from safetensors import safe_open
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfgd = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, vocab_size=128, max_position_embeddings=64)
    model = LlamaForCausalLM(LlamaConfig(architectures=["LlamaForCausalLM"], **cfgd)).to(torch.bfloat16)
    for k, v in model.state_dict().items():
        out[f"orig/{k}"] = bits(v)
    q = mtq.quantize(model, getattr(mtq, preset), None)
    with tempfile.TemporaryDirectory() as d:
        export_hf_checkpoint(q, export_dir=d)
        dtypes = {}
        with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
            for k in f.keys():
                t = f.get_tensor(k)
                out[f"exp/{k}"] = bits(t)
                dtypes[k] = str(t.dtype)
        quant_cfg = json.load(open(os.path.join(d, "hf_quant_config.json")))
    out["cases"] = np.array(json.dumps(dict(config=cfgd, dtypes=dtypes, hf_quant_config=quant_cfg, preset=preset)))


def gen_export_w4a8_mxfp4_fp8(out):
    """W4A8_MXFP4_FP8_CFG (MXFP4 block weights, per-tensor FP8 inputs; presets/model/w4a8_mxfp4_fp8.yaml) export."""
    _gen_export_uncalibrated(out, "W4A8_MXFP4_FP8_CFG")


def gen_export_mxfp4_mlp(out):
    """MXFP4_MLP_WEIGHT_ONLY_CFG (MXFP4 on the MLP projections only: a partially quantized model, exclude_modules with
    prefix wildcards; presets/model/mxfp4_mlp_weight_only.yaml) export."""
    _gen_export_uncalibrated(out, "MXFP4_MLP_WEIGHT_ONLY_CFG")


def gen_export_configs(out):
    """What export_hf_checkpoint writes NEXT TO the tensors, per preset, on the tiny Llama: hf_quant_config.json, the
    `quantization_config` it embeds into config.json (convert_hf_quant_config_format, export/convert_hf_config.py) and the
    file list.  One run mixes formats (INT8 weight-only on the MLPs, FP8 elsewhere): MIXED_PRECISION."""
    import copy
    import tempfile

    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint
    from transformers import LlamaConfig, LlamaForCausalLM

    cfgd = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, vocab_size=128, max_position_embeddings=64)
    batches = [torch.randint(0, 128, (4, 32), generator=torch.Generator().manual_seed(10 + i)) for i in range(2)]
    mixed = copy.deepcopy(mtq.FP8_DEFAULT_CFG)
    qc = mixed["quant_cfg"]
    int8_w = {"num_bits": 8, "axis": 0}
    if isinstance(qc, list):
        qc += [{"quantizer_name": "*mlp*weight_quantizer", "cfg": int8_w}, {"quantizer_name": "*mlp*input_quantizer", "enable": False}]
    else:
        qc["*mlp*weight_quantizer"] = int8_w
        qc["*mlp*input_quantizer"] = {"enable": False}
    kv = copy.deepcopy(mtq.FP8_DEFAULT_CFG)
    from modelopt.torch.quantization.utils import update_quant_cfg_with_kv_cache_quant
    kv = update_quant_cfg_with_kv_cache_quant(kv, mtq.FP8_KV_CFG["quant_cfg"])
    runs = {"int4_awq": (mtq.INT4_AWQ_CFG, True), "fp8": (mtq.FP8_DEFAULT_CFG, True), "fp8_kv": (kv, True),
            "int8_sq": (mtq.INT8_SMOOTHQUANT_CFG, True), "int8_wo": (mtq.INT8_WEIGHT_ONLY_CFG, True),
            "w4a8_awq": (mtq.W4A8_AWQ_BETA_CFG, True), "fp8_pc_pt": (mtq.FP8_PER_CHANNEL_PER_TOKEN_CFG, True),
            "mxfp4": (mtq.MXFP4_DEFAULT_CFG, False), "w4a8_mxfp4_fp8": (mtq.W4A8_MXFP4_FP8_CFG, False),
            "mxfp4_mlp": (mtq.MXFP4_MLP_WEIGHT_ONLY_CFG, False), "fp8_2d": (mtq.FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG, False),
            "mixed_fp8_int8wo": (mixed, True)}
    cases = {"config": cfgd, "n_batches": len(batches), "runs": {}}
    for i, b in enumerate(batches):
        out[f"tokens{i}"] = b.numpy()
    torch.manual_seed(0)
    base = LlamaForCausalLM(LlamaConfig(architectures=["LlamaForCausalLM"], **cfgd)).to(torch.bfloat16)
    for k, v in base.state_dict().items():
        out[f"orig/{k}"] = bits(v)
    for name, (cfg, calib) in runs.items():
        q = mtq.quantize(copy.deepcopy(base), copy.deepcopy(cfg), (lambda m: [m(b) for b in batches]) if calib else None)
        with tempfile.TemporaryDirectory() as d:
            export_hf_checkpoint(q, export_dir=d)
            files = sorted(os.listdir(d))
            hfq = json.load(open(os.path.join(d, "hf_quant_config.json")))
            cj = json.load(open(os.path.join(d, "config.json")))
        cases["runs"][name] = dict(files=files, hf_quant_config=hfq, quantization_config=cj.get("quantization_config"),
                                   config_keys=sorted(cj))
    out["cases"] = np.array(json.dumps(cases))


def gen_export_fp8_kv(out):
    """FP8 W + A with the FP8 KV-cache quantizers (FP8_DEFAULT_CFG + FP8_KV_CFG merged as hf_ptq does with
    update_quant_cfg_with_kv_cache_quant, examples/hf_ptq/hf_ptq.py:552-556) on the tiny fp32 Llama: the k / v
    bmm-quantizer amax of every attention after calibration, the logits of one batch with KV fake-quant active, and
    the exported k_scale / v_scale + hf_quant_config."""
    import copy
    import tempfile

    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint
    from safetensors import safe_open
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfgd = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, vocab_size=128, max_position_embeddings=64)
    for impl in ("sdpa", "eager"):
        cfg = LlamaConfig(architectures=["LlamaForCausalLM"], **cfgd)
        cfg._attn_implementation = impl
        torch.manual_seed(0)
        model = LlamaForCausalLM(cfg).to(torch.float32)
        batches = [torch.randint(0, 128, (4, 32), generator=torch.Generator().manual_seed(10 + i)) for i in range(3)]
        if impl == "sdpa":
            for k, v in model.state_dict().items():
                out[f"orig/{k}"] = bits(v)
            for i, b in enumerate(batches):
                out[f"tokens{i}"] = b.numpy()
        qcfg = mtq.update_quant_cfg_with_kv_cache_quant(copy.deepcopy(mtq.FP8_DEFAULT_CFG),
                                                        copy.deepcopy(mtq.FP8_KV_CFG["quant_cfg"]))
        q = mtq.quantize(model, qcfg, lambda m: [m(b) for b in batches])
        attns = []
        for n, m in q.named_modules():
            if hasattr(m, "k_bmm_quantizer"):
                attns.append(n)
                for which in "qkv":
                    tqz = getattr(m, f"{which}_bmm_quantizer")
                    out[f"{impl}/{n}.{which}_enabled"] = np.array(bool(tqz.is_enabled))
                    if tqz.is_enabled:
                        out[f"{impl}/{n}.{which}_amax"] = bits(tqz._amax.float())
        with torch.no_grad():
            out[f"{impl}/logits"] = bits(q(batches[0]).logits)
        if impl == "sdpa":
            with tempfile.TemporaryDirectory() as d:
                export_hf_checkpoint(q, export_dir=d)
                dtypes = {}
                with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
                    for k in f.keys():
                        if "k_scale" in k or "v_scale" in k or "k_bias" in k or "v_bias" in k:
                            out[f"exp/{k}"] = bits(f.get_tensor(k))
                            dtypes[k] = str(f.get_tensor(k).dtype)
                    all_keys = sorted(f.keys())
                quant_cfg = json.load(open(os.path.join(d, "hf_quant_config.json")))
    out["cases"] = np.array(json.dumps(dict(config=cfgd, n_batches=3, attentions=attns, dtypes=dtypes,
                                             exported_keys=all_keys, hf_quant_config=quant_cfg)))


def gen_moe_fp8(out):
    """FP8_DEFAULT_CFG on a tiny fp32 Mixtral (transformers >= 5: fused 3-D expert weights, quantized by the
    reference's _QuantFusedExperts, plugins/huggingface.py:976-1115): every enabled quantizer's amax (per-expert
    weight quantizers, shared input quantizers), logits with fake-quant active, and the exported checkpoint
    (per-expert gate / up / down projections split out of the fused tensors, moe_utils.py:48-215)."""
    import tempfile

    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint
    from safetensors import safe_open
    from transformers import MixtralConfig, MixtralForCausalLM

    torch.manual_seed(0)
    cfgd = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, vocab_size=128, max_position_embeddings=64, num_local_experts=4,
                num_experts_per_tok=2)
    model = MixtralForCausalLM(MixtralConfig(architectures=["MixtralForCausalLM"], **cfgd)).to(torch.float32)
    with torch.no_grad():  # expert 3 of layer 1 never wins the routing: exercises the uncalibrated-expert fallbacks
        model.model.layers[1].mlp.gate.weight[3] = 0
        model.model.layers[1].mlp.gate.weight[3, 0] = -1e4
    batches = [torch.randint(0, 128, (4, 32), generator=torch.Generator().manual_seed(10 + i)) for i in range(3)]
    for k, v in model.state_dict().items():
        out[f"orig/{k}"] = bits(v)
    for i, b in enumerate(batches):
        out[f"tokens{i}"] = b.numpy()
    q = mtq.quantize(model, mtq.FP8_DEFAULT_CFG, lambda m: [m(b) for b in batches])
    quantizers = {}
    for n, m in q.named_modules():
        if type(m).__name__ == "TensorQuantizer":
            quantizers[n] = bool(m.is_enabled)
            if m.is_enabled and getattr(m, "_amax", None) is not None:
                out[f"amax/{n}"] = bits(m._amax.float())
    with torch.no_grad():
        out["logits"] = bits(q(batches[0]).logits)
    with tempfile.TemporaryDirectory() as d:
        export_hf_checkpoint(q, export_dir=d)
        dtypes = {}
        with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
            for k in f.keys():
                t = f.get_tensor(k)
                dtypes[k] = str(t.dtype)
                if ".mlp." in k or "block_sparse_moe" in k:
                    out[f"exp/{k}"] = t.view(torch.uint8).numpy().copy() if t.dtype == torch.float8_e4m3fn else bits(t)
        quant_cfg = json.load(open(os.path.join(d, "hf_quant_config.json")))
    out["cases"] = np.array(json.dumps(dict(config=cfgd, n_batches=len(batches), quantizers=quantizers, dtypes=dtypes,
                                             hf_quant_config=quant_cfg)))


def gen_calibrate_weights(out):
    """calib.calibrate_weights (calib/histogram.py:346-433) on fp32 linears: per-channel and per-tensor percentile
    amax, "max", and the per-channel histograms numpy produced on the way (re-derived with the same np.histogram call)."""
    from modelopt.torch.quantization import calib as ref_calib
    from modelopt.torch.quantization import nn as qnn

    cases = {}
    gen = torch.Generator().manual_seed(21)
    for idx, (co, ci, kind) in enumerate([(16, 300, "normal"), (8, 4096, "heavy"), (5, 64, "grid"), (4, 33, "zero_row"),
                                          (6, 512, "large")]):
        lin = qnn.QuantLinear(ci, co, bias=False)
        with torch.no_grad():
            w = torch.randn(co, ci, generator=gen) * 0.05
            if kind == "heavy":
                w = torch.where(torch.rand(co, ci, generator=gen) < 0.002, w * 20, w)
            if kind == "grid":
                w = torch.randint(-8, 9, (co, ci), generator=gen).float() * 0.125  # values ON bin edges
            if kind == "zero_row":
                w[2] = 0
            if kind == "large":  # mean |w| above the bit width: the non-degenerate branch of the "mse" threshold
                w = w * torch.tensor([300.0, 400.0, 600.0, 1000.0, 150.0, 2000.0]).reshape(-1, 1)
            lin.weight.copy_(w)
        k = f"c{idx}"
        cases[k] = dict(cout=co, cin=ci, kind=kind)
        out[f"{k}_w"] = bits(lin.weight.detach())
        for tag, kw in [("pc9999", dict(method="percentile", perchannel=True)),
                        ("pc99", dict(method="percentile", perchannel=True, percentile=99.0)),
                        ("pt999", dict(method="percentile", perchannel=False, percentile=99.9)),
                        ("pcmax", dict(method="max", perchannel=True)),
                        ("pc512", dict(method="percentile", perchannel=True, percentile=99.5, num_bins=512)),
                        ("pcmse", dict(method="mse", perchannel=True, num_bins=512)),
                        ("ptmse", dict(method="mse", perchannel=False, num_bins=512))]:
            if kind == "zero_row" and "max" not in tag and "pt" not in tag:
                pass  # all-zero channel: numpy widens the range to (-0.5, 0.5); still a defined result
            lin.weight_quantizer.reset_amax()
            if kind == "zero_row" and tag == "pcmse":
                # the all-zero channel's histogram spans (-0.5, 0.5): candidate centres below zero, and the reference's
                # quantizer refuses a negative amax ("Negative values in amax")
                try:
                    ref_calib.calibrate_weights(lin, **kw)
                    raise AssertionError("expected the reference to refuse the negative candidates")
                except ValueError as e:
                    assert "Negative values in amax" in str(e)
                continue
            ref_calib.calibrate_weights(lin, **kw)
            out[f"{k}_{tag}"] = bits(lin.weight_quantizer.amax.float())
        hists = [np.histogram(r.abs().numpy(), bins=2048, range=(0, r.abs().numpy().max()))[0] for r in lin.weight.detach()]
        out[f"{k}_hist"] = np.stack(hists).astype(np.int32)
    out["cases"] = np.array(json.dumps(cases))


def gen_export_fp8_2d(out):
    """FP8 2-D blockwise weight-only (FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG, 128 x 128 tiles) export of a tiny bf16 Llama
    whose intermediate size (640) and head layout exercise several tile grids: per-tile amax of every linear, the
    exported e4m3 bytes and [R/128, 1, C/128, 1] scales; plus FP8QTensor.quantize / dequantize with blocks on both
    axes incl. padding (fp8_tensor.py:60-151)."""
    import tempfile

    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint
    from modelopt.torch.quantization.qtensor import FP8QTensor
    from safetensors import safe_open
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfgd = dict(hidden_size=256, intermediate_size=640, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, vocab_size=128, max_position_embeddings=64)
    model = LlamaForCausalLM(LlamaConfig(architectures=["LlamaForCausalLM"], **cfgd)).to(torch.bfloat16)
    for k, v in model.state_dict().items():
        out[f"orig/{k}"] = bits(v)
    q = mtq.quantize(model, mtq.FP8_2D_BLOCKWISE_WEIGHT_ONLY_CFG, None)
    for n, m in q.named_modules():
        if hasattr(m, "weight_quantizer") and m.weight_quantizer.is_enabled:
            out[f"pre/{n}.w_amax"] = bits(m.weight_quantizer._amax.float())
    with tempfile.TemporaryDirectory() as d:
        export_hf_checkpoint(q, export_dir=d)
        dtypes = {}
        with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
            for k in f.keys():
                t = f.get_tensor(k)
                out[f"exp/{k}"] = t.view(torch.uint8).numpy().copy() if t.dtype == torch.float8_e4m3fn else bits(t)
                dtypes[k] = str(t.dtype)
        quant_cfg = json.load(open(os.path.join(d, "hf_quant_config.json")))
    # the QTensor class on its own: computed scales (tensor dtype) and given fp32 scales, with padding on both axes
    qt_cases = {}
    gen = torch.Generator().manual_seed(5)
    for idx, (shape, blocks, dn, given) in enumerate([((256, 384), {-1: 128, -2: 128}, "bf16", False),
                                                      ((200, 300), {-1: 128, -2: 64}, "f16", False),
                                                      ((64, 96), {-1: 32, -2: 16}, "f32", False),
                                                      ((256, 256), {-1: 128, -2: 128}, "bf16", True)]):
        x = (torch.randn(*shape, generator=gen) * torch.exp(torch.randn(shape[0], 1, generator=gen))).to(DT[dn])
        scales = None
        if given:
            scales = (torch.rand(2, 2, generator=gen) * 0.01 + 1e-3).float()
        qt, sc = FP8QTensor.quantize(x, scales, block_sizes=blocks)
        deq = qt.dequantize(DT[dn], scale=sc, block_sizes=blocks)
        k = f"qt{idx}"
        qt_cases[k] = dict(shape=shape, blocks={str(a): b for a, b in blocks.items()}, dtype=dn, given=given,
                           scale_dtype=str(sc.dtype))
        out[f"{k}_x"], out[f"{k}_q"] = bits(x), qt._quantized_data.view(torch.uint8).numpy().copy()
        out[f"{k}_scales"], out[f"{k}_deq"] = bits(sc), bits(deq)
    out["cases"] = np.array(json.dumps(dict(config=cfgd, dtypes=dtypes, hf_quant_config=quant_cfg, qt=qt_cases)))


def gen_export_int8_sq(out):
    """INT8 SmoothQuant (INT8_SMOOTHQUANT_CFG: per-channel INT8 weights, per-tensor INT8 inputs, alpha = 1.0) checkpoint
    export of the tiny bf16 Llama by the reference: the calibrated state right before export (smoothed weights,
    per-channel weight amax, input amax, pre_quant_scale, norm weights) and every exported tensor (W8A8_SQ_PER_CHANNEL)."""
    import tempfile

    import modelopt.torch.quantization as mtq
    from modelopt.torch.export import export_hf_checkpoint
    from safetensors import safe_open
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfgd = dict(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                num_key_value_heads=2, vocab_size=128, max_position_embeddings=64)
    model = LlamaForCausalLM(LlamaConfig(architectures=["LlamaForCausalLM"], **cfgd)).to(torch.bfloat16)
    batches = [torch.randint(0, 128, (4, 32), generator=torch.Generator().manual_seed(10 + i)) for i in range(3)]
    for k, v in model.state_dict().items():
        out[f"orig/{k}"] = bits(v)
    for i, b in enumerate(batches):
        out[f"tokens{i}"] = b.numpy()
    q = mtq.quantize(model, mtq.INT8_SMOOTHQUANT_CFG, lambda m: [m(b) for b in batches])
    linears = []
    for n, m in q.named_modules():
        if hasattr(m, "weight_quantizer") and m.weight_quantizer.is_enabled:
            linears.append(n)
            out[f"pre/{n}.weight"] = bits(m.weight)
            out[f"pre/{n}.w_amax"] = bits(m.weight_quantizer._amax.float())
            out[f"pre/{n}.in_amax"] = bits(m.input_quantizer._amax.float())
            out[f"pre/{n}.pre_quant_scale"] = bits(m.input_quantizer._pre_quant_scale)
        elif type(m).__name__.endswith("RMSNorm"):
            out[f"pre/{n}.weight"] = bits(m.weight)
    with tempfile.TemporaryDirectory() as d:
        export_hf_checkpoint(q, export_dir=d)
        dtypes = {}
        with safe_open(os.path.join(d, "model.safetensors"), "pt") as f:
            for k in f.keys():
                t = f.get_tensor(k)
                out[f"exp/{k}"] = t.view(torch.uint8).numpy().copy() if t.dtype == torch.int8 else bits(t)
                dtypes[k] = str(t.dtype)
        quant_cfg = json.load(open(os.path.join(d, "hf_quant_config.json")))
    out["cases"] = np.array(json.dumps(dict(config=cfgd, n_batches=len(batches), linears=linears, dtypes=dtypes,
                                             hf_quant_config=quant_cfg)))


def gen_sq_mxfp4(out, have=None):
    """BASELINE configs[4] -- SmoothQuant scaling composed with MXFP4: the reference's smoothquant only acts
    on INT8 quantizers and its MX fake quantization needs the CUDA extension, so the composition is pinned from its two
    runnable halves.  (1) The scale math + fold: format independent, taken from the reference's INT8 SmoothQuant run of the
    tiny Llama (export_llama_int8_sq.npz: pre_quant_scale and smoothed weight of every linear -- what a composed flow
    must reproduce bit for bit from the ORIGINAL weights and tokens).  (2) The MXFP4 real quantization of THOSE smoothed
    weights by the reference's MXFP4QTensor.quantize (qtensor/mxfp4_tensor.py:37-81, CPU branch): packed E2M1 nibbles and
    E8M0 scale bytes, as export_hf_checkpoint stores them for an MXFP4 model.  alpha = 1.0 (the INT8 preset's)."""
    from modelopt.torch.quantization.qtensor import MXFP4QTensor

    # (`have`: the INT8 SmoothQuant arrays to build on -- the committed fixture, or, for tests/conftest.py's live fallback on a
    # host whose bf16 forward differs from the fixture host's, the same generator's output on that host)
    have = np.load(os.path.join(HERE, "export_llama_int8_sq.npz")) if have is None else have
    cases = json.loads(str(have["cases"]))
    for n in cases["linears"]:
        raw = have[f"pre/{n}.weight"]
        w = torch.from_numpy(raw.view(np.int16).copy()).view(torch.bfloat16)
        qt, e8m0 = MXFP4QTensor.quantize(w, 32)
        out[f"mx/{n}.weight"] = qt._quantized_data.view(torch.uint8).numpy().copy()
        out[f"mx/{n}.weight_scale"] = e8m0.view(torch.uint8).numpy().copy()
    out["cases"] = np.array(json.dumps(dict(linears=cases["linears"], alpha=1.0, block=32)))


def gen_mxfp8(out):
    """MXFP8QTensor (qtensor/mxfp8_tensor.py:26-268) on CPU: E8M0 scale bytes, E4M3 bytes and the dequantised tensor
    for 2-D / 3-D weights, a ragged last dim, all-zero blocks, block maxima ON the 448 * 2^k boundary and tiny /
    huge magnitudes (exponent clamps)."""
    from modelopt.torch.quantization.qtensor import MXFP8QTensor

    cases = {}
    idx = 0
    for dn, dt in DT.items():
        for shape, kind in [((48, 256), "normal"), ((3, 16, 64), "moe3d"), ((5, 70), "ragged"), ((8, 128), "edges")]:
            w = weight_like(shape, dt, 1700 + idx)
            if kind == "normal":
                w[1, :32] = 0
                w[2, 32:64] = w[2, 32:64] * 1e4
            if kind == "edges":
                w[0, :32] = 0
                w[1, 0] = 448.0          # descale == 1 exactly
                w[2, 0] = 1.75           # 448 * 2^-8
                w[3, 0] = 3.5 * 2.0 ** -20
                w[4, :32] = w[4, :32] * 1e-30 if dt != torch.float16 else w[4, :32] * 1e-4
                w[5, 0] = 6e4 if dt == torch.float16 else 3e38
                w[6, 0] = 449.0 if dt == torch.float32 else 450.0
                w[7, 5] = -448.0
            qt, e8 = MXFP8QTensor.quantize(w)
            deq = qt.dequantize(scale=e8)
            k = f"{dn}_{kind}"
            out[f"{k}_x"], out[f"{k}_q"] = bits(w), qt._quantized_data.view(torch.uint8).numpy().copy()
            out[f"{k}_e8m0"], out[f"{k}_deq"] = e8.numpy().copy(), bits(deq)
            cases[k] = dict(dtype=dn, shape=list(shape), kind=kind)
            idx += 1
    out["cases"] = np.array(json.dumps(cases))


def main():
    torch.manual_seed(1234)
    only = sys.argv[1:] or None
    single = {"hist": gen_hist, "mse": gen_mse, "export_llama": gen_export, "awq_clip": gen_awq_clip, "qtensor": gen_qtensor, "w4a8": gen_w4a8, "sgpt": gen_sgpt, "gptq": gen_gptq, "local_hessian": gen_local_hessian, "affine_bias": gen_affine_bias, "gptq_llama": gen_gptq_llama, "block2d": gen_block2d, "export_llama_fp8": gen_export_fp8, "export_llama_mxfp4": gen_export_mxfp4, "export_configs": gen_export_configs, "export_llama_w4a8_mxfp4_fp8": gen_export_w4a8_mxfp4_fp8, "export_llama_mxfp4_mlp": gen_export_mxfp4_mlp, "export_llama_fp8_kv": gen_export_fp8_kv, "moe_fp8": gen_moe_fp8, "calibrate_weights": gen_calibrate_weights, "export_llama_fp8_2d": gen_export_fp8_2d, "export_llama_int8_sq": gen_export_int8_sq, "mxfp8": gen_mxfp8, "export_llama_replay": gen_export_replay, "sq_mxfp4": gen_sq_mxfp4, "export_llama_w4a8": gen_export_w4a8, "awq_ragged": gen_awq_ragged,
              "export_llama_fp8_pc_pt": gen_export_fp8_pc_pt}
    for name, fn in [(only[0], single[only[0]])] if only and only[0] in single else [("int_fq", gen_int_fq), ("fp8_fq", gen_fp8), ("amax", gen_amax),
                     ("tq_block", gen_tq_block), ("hist", gen_hist), ("mask24", gen_mask),
                     ("int4", gen_int4), ("awq", gen_awq), ("model_flows", gen_model_flows), ("mse", gen_mse),
                     ("export_llama", gen_export), ("awq_clip", gen_awq_clip), ("qtensor", gen_qtensor), ("w4a8", gen_w4a8), ("sgpt", gen_sgpt), ("gptq", gen_gptq), ("local_hessian", gen_local_hessian), ("affine_bias", gen_affine_bias), ("gptq_llama", gen_gptq_llama), ("block2d", gen_block2d), ("export_llama_fp8", gen_export_fp8), ("export_llama_mxfp4", gen_export_mxfp4), ("export_configs", gen_export_configs), ("export_llama_w4a8_mxfp4_fp8", gen_export_w4a8_mxfp4_fp8), ("export_llama_mxfp4_mlp", gen_export_mxfp4_mlp), ("export_llama_fp8_kv", gen_export_fp8_kv), ("moe_fp8", gen_moe_fp8), ("calibrate_weights", gen_calibrate_weights), ("export_llama_fp8_2d", gen_export_fp8_2d), ("export_llama_int8_sq", gen_export_int8_sq), ("mxfp8", gen_mxfp8), ("export_llama_replay", gen_export_replay), ("sq_mxfp4", gen_sq_mxfp4), ("export_llama_w4a8", gen_export_w4a8), ("awq_ragged", gen_awq_ragged),
                     ("export_llama_fp8_pc_pt", gen_export_fp8_pc_pt)]:
        out = {}
        fn(out)
        path = os.path.join(HERE, f"{name}.npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")
    if only:
        return
    mx = extract_mx_vectors()
    with open(os.path.join(HERE, "mx_vectors.json"), "w") as f:
        json.dump(mx, f)
    print(f"mx_vectors: {len(mx)} cases")


if __name__ == "__main__":
    main()
