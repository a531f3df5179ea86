"""Layer-by-layer calibration: same statistics as a whole-model pass, checkpoint / resume."""

import copy

import pytest
import torch

import _moa_import

pytestmark = pytest.mark.gpu
moa = _moa_import.load()
from model_optimizer_amd import layerwise, model_calib, model_quant  # noqa: E402

DEV = "cuda:0"


class Block(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.fc1 = torch.nn.Linear(d, 2 * d, bias=False)
        self.fc2 = torch.nn.Linear(2 * d, d, bias=False)

    def forward(self, h, scale=1.0):
        return h + scale * self.fc2(torch.nn.functional.gelu(self.fc1(h)))


class Stack(torch.nn.Module):
    def __init__(self, d=128, n=4):
        super().__init__()
        self.embed = torch.nn.Linear(d, d, bias=False)
        self.layers = torch.nn.ModuleList([Block(d) for _ in range(n)])

    def forward(self, x):
        h = self.embed(x)
        for layer in self.layers:
            h = layer(h, scale=0.5)
        return h


def _setup(cfg):
    torch.manual_seed(0)
    model = Stack().to(DEV).to(torch.bfloat16)
    batches = [torch.randn(16, 128, device=DEV).to(torch.bfloat16) for _ in range(3)]
    moa.nn.replace_quant_module(model)
    q_cfg = {k: v for k, v in cfg["quant_cfg"].items()}
    q_cfg["*embed*"] = {"enable": False}
    model_quant.set_quantizer_by_cfg(model, q_cfg)
    return model, batches


def _amax(model):
    return {n: q._amax.detach().float().cpu().clone() for n, q in model.named_modules()
            if isinstance(q, moa.TensorQuantizer) and hasattr(q, "_amax")}


@pytest.mark.parametrize("cfg", [model_quant.FP8_DEFAULT_CFG, model_quant.INT4_BLOCKWISE_WEIGHT_ONLY_CFG])
def test_layerwise_max_equals_whole_model_max(cfg):
    model, batches = _setup(cfg)
    whole = copy.deepcopy(model)
    model_calib.max_calibrate(whole, lambda m: [m(b) for b in batches])
    n = layerwise.layerwise_calibrate(model, lambda m: [m(b) for b in batches], model_calib.max_calibrate)
    assert n == 4
    a, b = _amax(whole), _amax(model)
    assert set(a) == set(b) and len(a) >= 8
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k}: layerwise amax differs from the whole-model pass"


def test_layerwise_resume(tmp_path):
    cfg = model_quant.FP8_DEFAULT_CFG
    model, batches = _setup(cfg)
    ref = copy.deepcopy(model)
    layerwise.layerwise_calibrate(ref, lambda m: [m(b) for b in batches], model_calib.max_calibrate)
    calls = {"n": 0}

    def flaky(layer, loop, **kw):
        if calls["n"] == 2:
            raise KeyboardInterrupt
        calls["n"] += 1
        model_calib.max_calibrate(layer, loop, **kw)

    m1 = copy.deepcopy(model)
    with pytest.raises(KeyboardInterrupt):
        layerwise.layerwise_calibrate(m1, lambda m: [m(b) for b in batches], flaky, checkpoint_dir=str(tmp_path))
    m2 = copy.deepcopy(model)  # a fresh process: nothing calibrated yet
    n = layerwise.layerwise_calibrate(m2, lambda m: [m(b) for b in batches], model_calib.max_calibrate,
                                      checkpoint_dir=str(tmp_path))
    assert n == 2, "resume must start at the first unfinished layer"
    a, b = _amax(ref), _amax(m2)
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k}: resumed run differs"
    assert layerwise.layerwise_calibrate(m2, lambda m: [m(b) for b in batches], model_calib.max_calibrate,
                                         checkpoint_dir=str(tmp_path)) == 0


@pytest.mark.parametrize("search", ["gram", "gemm"])
def test_layerwise_awq_lite_equals_whole_model(search):
    """awq_lite as the per-layer calibration function (how a model whose Gram matrices do not all fit at once --
    Llama-3-70B: 263 GB -- still takes the Gram search: one layer's matrices are alive at a time).  The layer inputs
    come from the un-quantized previous layers, as in the whole-model pass, so alphas and folded weights agree."""
    model, batches = _setup(model_quant.INT4_AWQ_CFG)
    whole = copy.deepcopy(model)
    hw = model_calib.awq_lite(whole, lambda m: [m(b) for b in batches], search=search)
    n = layerwise.layerwise_calibrate(model, lambda m: [m(b) for b in batches], model_calib.awq_lite, search=search)
    assert n == 4 and len(hw) == 8
    for (name, a), (_, b) in zip(whole.layers.named_modules(), model.layers.named_modules()):
        if hasattr(a, "awq_lite"):
            assert a.awq_lite.best_alpha == b.awq_lite.best_alpha, name
            assert torch.equal(a.weight, b.weight), name
            assert torch.equal(a.input_quantizer.pre_quant_scale, b.input_quantizer.pre_quant_scale), name
            assert torch.equal(a.weight_quantizer.amax, b.weight_quantizer.amax), name


def test_layerwise_resume_after_an_interrupted_checkpoint_write(tmp_path):
    """A crash between the write of next_inputs.pt (inputs of layer N + 1) and the manifest leaves the manifest at
    completed = N: the saved inputs name the layer they feed, a resume that finds another layer's inputs re-captures them
    instead of calibrating layer N on layer N + 1's activations; files are moved into place atomically."""
    import json
    import os

    cfg = model_quant.FP8_DEFAULT_CFG
    model, batches = _setup(cfg)
    ref = copy.deepcopy(model)
    layerwise.layerwise_calibrate(ref, lambda m: [m(b) for b in batches], model_calib.max_calibrate)
    calls = {"n": 0}

    def flaky(layer, loop, **kw):
        if calls["n"] == 3:
            raise KeyboardInterrupt
        calls["n"] += 1
        model_calib.max_calibrate(layer, loop, **kw)

    m1 = copy.deepcopy(model)
    with pytest.raises(KeyboardInterrupt):
        layerwise.layerwise_calibrate(m1, lambda m: [m(b) for b in batches], flaky, checkpoint_dir=str(tmp_path))
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp")]
    blob = torch.load(os.path.join(tmp_path, "next_inputs.pt"), weights_only=False)
    assert blob["for_layer"] == 3
    # the simulated crash: next_inputs.pt already holds layer 3's inputs, the manifest still says 2 layers are done
    with open(os.path.join(tmp_path, "manifest.json"), "w") as f:
        json.dump({"num_layers": 4, "completed": 2}, f)
    m2 = copy.deepcopy(model)
    with pytest.warns(UserWarning, match="interrupted checkpoint write"):
        n = layerwise.layerwise_calibrate(m2, lambda m: [m(b) for b in batches], model_calib.max_calibrate,
                                          checkpoint_dir=str(tmp_path))
    assert n == 2
    a, b = _amax(ref), _amax(m2)
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), f"{k}: run resumed from re-captured inputs differs"


def test_layerwise_runs_without_autograd(tmp_path):
    """calib_func, the input capture, the hand-over to the next layer and the replay after an interrupted checkpoint
    write all run under no_grad (no autograd graph through every decoder-layer forward on the 70B flow)."""
    model, batches = _setup(model_quant.FP8_DEFAULT_CFG)
    seen = []

    class Probe(torch.nn.Module):
        def forward(self, x):
            seen.append(torch.is_grad_enabled())
            return x

    for layer in model.layers:
        layer.fc1 = torch.nn.Sequential(Probe(), layer.fc1)

    def calib(layer, loop, **kw):
        seen.append(torch.is_grad_enabled())
        model_calib.max_calibrate(layer, loop, **kw)

    with torch.enable_grad():
        layerwise.layerwise_calibrate(model, lambda m: [m(b) for b in batches], calib, checkpoint_dir=str(tmp_path))
        inputs = layerwise._replay_to_layer(model, model.layers, 2, lambda m: [m(b) for b in batches])
    assert seen and not any(seen), "autograd was enabled inside the layerwise flow"
    assert all(not a.requires_grad for args, _ in inputs for a in args if isinstance(a, torch.Tensor))


def test_parent_walk_hands_every_block_its_own_arguments():
    """The default capture mode runs the PARENT's forward for every layer (layerwise.DecoderWalk: finished blocks are
    meta placeholders, the previous block replays its recorded inputs, the target records and stops), so arguments the
    parent computes per block reach each block; the hand-over of the first call's arguments does not see them."""

    class PerLayer(Stack):
        def forward(self, x):
            h = self.embed(x)
            for i, layer in enumerate(self.layers):
                h = layer(h, scale=0.25 * (i + 1))
            return h

    torch.manual_seed(0)
    model = PerLayer().to(DEV).to(torch.bfloat16)
    batches = [torch.randn(16, 128, device=DEV).to(torch.bfloat16) for _ in range(3)]
    moa.nn.replace_quant_module(model)
    model_quant.set_quantizer_by_cfg(model, {**model_quant.FP8_DEFAULT_CFG["quant_cfg"], "*embed*": {"enable": False}})
    whole, handed = copy.deepcopy(model), copy.deepcopy(model)
    model_calib.max_calibrate(whole, lambda m: [m(b) for b in batches])
    assert layerwise.layerwise_calibrate(model, lambda m: [m(b) for b in batches], model_calib.max_calibrate) == 4
    a, b = _amax(whole), _amax(model)
    assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert all("forward" not in m.__dict__ for m in model.modules())
    layerwise.layerwise_calibrate(handed, lambda m: [m(b) for b in batches], model_calib.max_calibrate, capture="handover")
    c = _amax(handed)
    assert any(not torch.equal(a[k], c[k]) for k in a if ".layers.0." not in k)
