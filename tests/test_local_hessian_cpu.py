"""local_hessian_calibrate (model_calib.py:1005-1127 of the reference) on CPU through the host-memory stand-in of the
C-ABI: the refined amax of both linears of the tiny MLP against the reference run (tests/golden/local_hessian.npz), next to
max and plain mse."""

import copy
import warnings

import pytest
import torch

import _moa_import
import hostmem_backend
from conftest import assert_bits_equal

moa = _moa_import.load()
DT = {"float32": torch.float32, "bfloat16": torch.bfloat16}


@pytest.fixture
def hostmem(monkeypatch):
    return hostmem_backend.install(monkeypatch, moa)


def _model(g, name, dt):
    class TinyMLP(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.fc1 = torch.nn.Linear(128, 128, bias=False)
            self.fc2 = torch.nn.Linear(128, 128, bias=True)

        def forward(self, x):
            return self.fc2(torch.nn.functional.gelu(self.fc1(x)))

    m = TinyMLP().to(dt)
    m.fc1.weight.data.copy_(g.t(f"{name}_w1", dt))
    m.fc2.weight.data.copy_(g.t(f"{name}_w2", dt))
    m.fc2.bias.data.copy_(g.t(f"{name}_b2", dt))
    return m


def _cfg(alg):
    cfg = copy.deepcopy(moa.model_quant.INT4_BLOCKWISE_WEIGHT_ONLY_CFG)
    cfg["quant_cfg"]["*weight_quantizer"] = {"num_bits": 4, "block_sizes": {-1: 16, "type": "static"}, "enable": True}
    cfg["algorithm"] = alg
    return cfg


@pytest.mark.parametrize("name", ["lh_f32", "lh_bf16"])
def test_local_hessian_amax_equals_the_reference_run(golden, hostmem, name):
    g = golden("local_hessian")
    dt = DT[g.cases[name]["dtype"]]
    batches = [g.t(f"{name}_x{i}", dt) for i in range(g.cases[name]["n_batches"])]
    picked = {}
    for alg_name, alg in (("max", "max"), ("mse", {"method": "mse"}),
                          ("local_hessian", {"method": "local_hessian", "fp8_scale_sweep": False, "block_size": 16})):
        model = moa.quantize(_model(g, name, dt), _cfg(alg), lambda m: [m(b) for b in batches])
        for lname in ("fc1", "fc2"):
            got = getattr(model, lname).weight_quantizer._amax.float().reshape(-1)
            want = g.t(f"{name}_{alg_name}_{lname}_amax").reshape(-1)
            if alg_name == "local_hessian":
                # the per-block Hessian comes out of a batched GEMM: on a near-tie of two multipliers the pick can flip
                same = (got == want).float().mean().item()
                assert same >= 0.98, f"{alg_name} {lname}: {same:.4f} of the amax entries equal the reference's"
            else:
                assert_bits_equal(got, want, f"{alg_name} {lname} amax")
            picked[(alg_name, lname)] = got
    # the Hessian-weighted search is a different search: it must not collapse onto plain mse or max
    assert not torch.equal(picked[("local_hessian", "fc1")], picked[("mse", "fc1")])
    assert not torch.equal(picked[("local_hessian", "fc1")], picked[("max", "fc1")])


def test_local_hessian_default_sweep_leaves_non_nvfp4_quantizers_at_max(golden, hostmem):
    """fp8_scale_sweep=True (the reference's default) searches static NVFP4 quantizers only: every other weight keeps its
    max-calibrated amax (_make_weight_mse_calibrator, model_calib.py:695-718)."""
    g = golden("local_hessian")
    dt = torch.float32
    batches = [g.t(f"lh_f32_x{i}", dt) for i in range(g.cases["lh_f32"]["n_batches"])]
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        model = moa.quantize(_model(g, "lh_f32", dt), _cfg({"method": "local_hessian"}), lambda m: [m(b) for b in batches])
    assert any("fp8_scale_sweep" in str(x.message) for x in w)
    for lname in ("fc1", "fc2"):
        assert_bits_equal(getattr(model, lname).weight_quantizer._amax.float().reshape(-1),
                          g.t(f"lh_f32_max_{lname}_amax").reshape(-1), f"{lname} amax stays at max")


def test_local_hessian_shares_one_accumulator_between_linears_that_read_one_tensor(hostmem):
    class QKV(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q, self.k = torch.nn.Linear(64, 64, bias=False), torch.nn.Linear(64, 32, bias=False)

        def forward(self, x):
            return self.q(x), self.k(x)

    torch.manual_seed(1)
    batches = [torch.randn(40, 64) for _ in range(3)]
    model = moa.quantize(QKV(), _cfg({"method": "local_hessian", "fp8_scale_sweep": False, "block_size": 16, "debug": True}),
                         lambda m: [m(b) for b in batches])
    accs = model._local_hessian_accumulators
    assert len(accs) == 2 and len({id(a) for a in accs.values()}) == 1
    acc = next(iter(accs.values()))
    assert acc.num_samples == 120
    x = torch.cat(batches).float().T.reshape(4, 16, -1)
    assert torch.allclose(acc.normalized_hessian(), (x @ x.transpose(-1, -2)) / 120, rtol=1e-5, atol=1e-6)
