"""GPU parity of the fused MSE amax sweep (moq_mse_sweep) and the MSE calibration flow against the oracle and the
reference-generated fixtures (tests/golden/mse.npz).

The per-candidate losses are fp32 sums whose order differs from torch's, so they carry a stated tolerance
(rtol 2e-5 against the fp64 oracle / the reference's fp32 sums); what the calibration produces -- the chosen
amax per tensor / channel / block -- is discrete and must equal the reference's bit for bit.
"""

import copy
from functools import partial

import pytest
import torch

import _moa_import
from conftest import DT

pytestmark = pytest.mark.gpu

moa = _moa_import.load()
ops = moa.ops
from oracle import oracle  # noqa: E402  (the checker)
from test_gpu_host import TinyMLP  # noqa: E402

DEV = "cuda:0"
MULT = torch.linspace(0.25, 4.0, steps=39)


def _rel(a, b):
    return ((a.double() - b.double()).abs() / b.double().abs().clamp_min(1e-20)).max().item()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("layout", ["tensor", "channel", "block128", "block16", "tensor_ragged", "channel_ragged"])
@pytest.mark.parametrize("fmt", [4, 8, (4, 3)])
def test_mse_sweep_vs_oracle(dtype, layout, fmt):
    torch.manual_seed(hash((str(dtype), layout, str(fmt))) % 1000)
    shape = {"tensor": (64, 1024), "channel": (48, 4096), "block128": (32, 1024), "block16": (8, 256),
             "tensor_ragged": (37, 531), "channel_ragged": (19, 4099)}[layout]
    w = (torch.randn(*shape) * 0.02).to(dtype)
    w[0, 0] = 0.5  # an outlier so that the best multiplier is not trivially 1
    if layout.startswith("block"):
        g = int(layout[5:])
        wv, reduce_axis = w.reshape(-1, g), (1,)
        outer, axis_size, inner = 1, wv.shape[0], g
    elif layout.startswith("channel"):
        wv, reduce_axis = w, (1,)
        outer, axis_size, inner = 1, w.shape[0], w.shape[1]
    else:
        wv, reduce_axis = w, None
        outer, axis_size, inner = 1, 1, w.numel()
    init = wv.float().abs().amax(dim=1) if reduce_axis else wv.float().abs().amax().reshape(1)
    cand = MULT[:, None] * init[None, :]
    fp8 = not isinstance(fmt, int)
    want = oracle.mse_sweep(wv.contiguous(), cand, outer, axis_size, inner, fp8=fp8, num_bits=8 if fp8 else fmt)
    got = ops.mse_sweep(wv.to(DEV), cand.to(DEV), reduce_axis, fmt, False, False).cpu()
    assert got.shape == want.shape
    assert _rel(got, want) < 2e-5, f"max rel loss diff {_rel(got, want):.2e}"
    # accumulate: a second collect adds
    acc = ops.mse_sweep(wv.to(DEV), cand.to(DEV), reduce_axis, fmt, False, False)
    ops.mse_sweep(wv.to(DEV), cand.to(DEV), reduce_axis, fmt, False, False, loss=acc)
    assert _rel(acc.cpu(), 2 * want) < 2e-5


def test_mse_calibrator_matches_reference(golden):
    """MseCalibrator driven like the reference's mse_calibrate: losses within tolerance, chosen amax identical."""
    g = golden("mse")
    tq = moa.tensor_quantizer
    for name, c in g.cases.items():
        if name.startswith("flow_"):
            continue
        cfg = dict(c["cfg"])
        if isinstance(cfg["num_bits"], list):
            cfg["num_bits"] = tuple(cfg["num_bits"])
        if "block_sizes" in cfg:
            cfg["block_sizes"] = {(int(k) if k.lstrip("-").isdigit() else k): v for k, v in cfg["block_sizes"].items()}
        w = g.t(f"{name}_w", DT[c["dtype"]]).to(DEV)
        # a bare quantizer, exactly as the fixture generator drives the reference (gen_golden.gen_mse): max
        # calibration, then an MseCalibrator around the quantizer's own fake quant
        q = tq.TensorQuantizer(tq.QuantizerAttributeConfig(narrow_range=False, **cfg))
        moa.model_calib.max_calibrate(q, lambda qq: qq(w), distributed_sync=False)
        init = q._amax.clone().detach()
        assert str(init.dtype) == c["init_dtype"] and list(init.shape) == c["init_shape"]
        nb = q._num_bits
        cal = moa.calib.MseCalibrator(amax=init, axis=q._calibrator._axis, step_size=0.1, start_multiplier=0.25,
                                      stop_multiplier=4.0,
                                      quant_func=partial(moa.model_calib._mse_quant_func, quantizer=q),
                                      fused_format=(nb, q._unsigned, q._narrow_range))
        q._calibrator = cal
        q.disable_quant()
        q.enable_calib()
        q(w)
        seen = {"losses": torch.stack([l.reshape(-1) for l in cal._losses_sum]).float().cpu(),
                "amax": cal.compute_amax()}
        want_l = g.t(f"{name}_losses")
        assert _rel(seen["losses"], want_l) < 5e-5, f"{name}: losses rel {_rel(seen['losses'], want_l):.2e}"
        got = seen["amax"]  # compute_amax(): fp32 (1-D candidates x amax promote), before the buffer copy
        assert str(got.dtype) == c["amax_dtype"] and list(got.shape) == c["amax_shape"], \
            f"{name}: amax {got.dtype} {tuple(got.shape)} vs {c['amax_dtype']} {c['amax_shape']}"
        assert torch.equal(got.float().cpu().reshape(-1), g.t(f"{name}_amax").reshape(-1)), f"{name}: chosen amax differs"
        # the unfused per-candidate path (quant_func loop on our QDQ kernels) must agree with the fused kernel
        cal2 = moa.calib.MseCalibrator(amax=init, axis=cal._axis, step_size=0.1, start_multiplier=0.25,
                                       stop_multiplier=4.0,
                                       quant_func=partial(moa.model_calib._mse_quant_func, quantizer=q))
        cal2.collect(q._process_for_blockquant(w) if q.is_static_block_quant else w)
        l2 = torch.stack([l.reshape(-1) for l in cal2._losses_sum]).float().cpu()
        assert _rel(l2, want_l) < 5e-5, f"{name}: unfused losses rel {_rel(l2, want_l):.2e}"


@pytest.mark.parametrize("name", ["flow_int8_mse", "flow_int4blk_mse_bf16"])
def test_quantize_mse_flow_matches_reference(golden, name):
    g, gm = golden("mse"), golden("model_flows")
    c = g.cases[name]
    dn = c["dtype"]
    dt = DT[dn]
    model = TinyMLP(gm.t(f"{dn}_w1", dt), gm.t(f"{dn}_w2", dt), gm.t(f"{dn}_b2", dt)).to(DEV)
    batches = [gm.t(f"{dn}_x{i}", dt).to(DEV) for i in range(c["n_batches"])]
    mq = moa.model_quant
    cfg = copy.deepcopy(mq.INT8_DEFAULT_CFG if name == "flow_int8_mse" else mq.INT4_BLOCKWISE_WEIGHT_ONLY_CFG)
    cfg["algorithm"] = {"method": "mse"}
    q = moa.quantize(model, cfg, lambda m: [m(b) for b in batches])
    for lname in ("fc1", "fc2"):
        a = getattr(q, lname).weight_quantizer._amax
        assert list(a.shape) == c[f"{lname}_amax_shape"]
        assert torch.equal(a.float().cpu().reshape(-1), g.t(f"{name}_{lname}_weight_amax").reshape(-1)), \
            f"{name} {lname}: MSE-calibrated weight amax differs from the reference"
