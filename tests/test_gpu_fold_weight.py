"""model_quant.fold_weight (mtq.fold_weight, quantization/model_quant.py:728-736): the whole-model multi-tensor launches
must give exactly what every weight quantizer's own forward gives (the reference's tensor-by-tensor fold), for every
format group; the live comparison with the reference itself runs on CPU (tests/test_differential_cpu.py)."""

import copy

import pytest
import torch

import _moa_import

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import model_quant  # noqa: E402

DEV = "cuda:0"


class Net(torch.nn.Module):
    def __init__(self, dtype):
        super().__init__()
        g = torch.Generator().manual_seed(0)
        dims = [(256, 512), (512, 256), (384, 512), (128, 1024)]
        self.linears = torch.nn.ModuleList(torch.nn.Linear(ci, co, bias=False) for co, ci in dims)
        with torch.no_grad():
            for lin in self.linears:
                lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * 0.03)
        self.to(dtype)

    def forward(self, xs):
        return [lin(x) for lin, x in zip(self.linears, xs)]


PER_TENSOR_INT8 = {"quant_cfg": {"*weight_quantizer": {"num_bits": 8, "axis": None}, "*input_quantizer": {"enable": False}},
                   "algorithm": "max"}
UNCALIBRATED_INT4_BLOCKS = {"quant_cfg": {"*weight_quantizer": {"num_bits": 4, "block_sizes": {-1: 128, "type": "static"}},
                                          "*input_quantizer": {"enable": False}}, "algorithm": None}


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("cfg_name", ["FP8_DEFAULT_CFG", "PER_TENSOR_INT8", "MXFP4_DEFAULT_CFG", "UNCALIBRATED_INT4_BLOCKS",
                                      "INT4_BLOCKWISE_WEIGHT_ONLY_CFG", "INT8_DEFAULT_CFG"])
def test_fold_weight_equals_each_quantizers_own_forward(cfg_name, dtype):
    cfg = copy.deepcopy(globals().get(cfg_name) or getattr(model_quant, cfg_name))
    model = Net(dtype).to(DEV)
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(16, lin.in_features, generator=g).to(dtype).to(DEV) for lin in model.linears]
    with torch.no_grad():
        moa.quantize(model, cfg, (lambda m: m(xs)) if cfg["algorithm"] else None)
        want = [lin.weight_quantizer(lin.weight).clone() for lin in model.linears]
        out_before = model(xs)
        batched = sum(model_quant._fold_kind(lin.weight.data, lin.weight_quantizer) is not None for lin in model.linears)
        model_quant.fold_weight(model)
        out_after = model(xs)
    # which formats take the whole-model launches: everything but quantizers holding a per-block / per-channel amax
    assert batched == (0 if cfg_name in ("INT4_BLOCKWISE_WEIGHT_ONLY_CFG", "INT8_DEFAULT_CFG") else len(model.linears))
    for lin, w in zip(model.linears, want):
        assert torch.equal(lin.weight.view(torch.int32 if dtype == torch.float32 else torch.int16),
                           w.view(torch.int32 if dtype == torch.float32 else torch.int16))
        wq = lin.weight_quantizer
        assert not wq.is_enabled and not hasattr(wq, "_amax")
    for a, b in zip(out_before, out_after):
        assert torch.equal(a, b)


def test_fold_weight_keep_attrs():
    model = Net(torch.bfloat16).to(DEV)
    with torch.no_grad():
        moa.quantize(model, copy.deepcopy(model_quant.FP8_DEFAULT_CFG),
                     lambda m: m([torch.randn(4, lin.in_features, device=DEV).to(torch.bfloat16) for lin in m.linears]))
        amax = [lin.weight_quantizer._amax.clone() for lin in model.linears]
        model_quant.fold_weight(model, keep_attrs=True)
    for lin, a in zip(model.linears, amax):
        assert not lin.weight_quantizer.is_enabled and torch.equal(lin.weight_quantizer._amax, a)


@pytest.mark.parametrize("cfg_name", ["FP8_DEFAULT_CFG", "MXFP4_DEFAULT_CFG"])
def test_fold_weight_is_idempotent_at_llama_layer_size(cfg_name):
    """One Llama-3-8B decoder layer's seven weights (218 M elements) through the whole-model launches: a fake-quantized
    weight is a fixed point of its own quantizer (same per-tensor amax / same block scales), so folding a second time with
    re-enabled quantizers must change nothing -- a size-independent check where per-element oracles are too slow."""
    h, i, kv = 4096, 14336, 1024
    torch.manual_seed(2)

    class Layer(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.linears = torch.nn.ModuleList(torch.nn.Linear(ci, co, bias=False, device=DEV, dtype=torch.bfloat16)
                                               for co, ci in [(h, h), (kv, h), (kv, h), (h, h), (i, h), (i, h), (h, i)])

        def forward(self, x):
            return [lin(x[..., :lin.in_features]) for lin in self.linears]

    model = Layer()
    with torch.no_grad():
        for lin in model.linears:
            lin.weight.mul_(0.02 / lin.weight.std())
        moa.quantize(model, copy.deepcopy(getattr(model_quant, cfg_name)),
                     (lambda m: m(torch.randn(4, i, device=DEV, dtype=torch.bfloat16))) if cfg_name == "FP8_DEFAULT_CFG" else None)
        amax = [getattr(lin.weight_quantizer, "_amax", None) for lin in model.linears]
        amax = [a.clone() if a is not None else None for a in amax]
        model_quant.fold_weight(model, keep_attrs=True)
        once = [lin.weight.detach().clone() for lin in model.linears]
        for lin in model.linears:
            lin.weight_quantizer.enable()
        model_quant.fold_weight(model, keep_attrs=True)
    for lin, w, a in zip(model.linears, once, amax):
        # value equality: the MX conversion maps a -0 it produced itself to +0 on the second pass (sign of an element
        # that rounds to zero is kept, the sign of a zero INPUT is not: tensor_quant_mx.h's sign * magnitude form)
        assert torch.equal(lin.weight.float(), w.float())
        if cfg_name == "FP8_DEFAULT_CFG":
            assert torch.equal(lin.weight.view(torch.int16), w.view(torch.int16))
        if a is not None:
            assert torch.equal(lin.weight_quantizer._amax, a)
