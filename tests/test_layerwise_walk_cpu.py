"""Layer-by-layer calibration driven through the PARENT model's forward (layerwise.DecoderWalk = the reference's skip /
run / capture strategy, utils/layerwise_calib.py) on CPU, host through tests/hostmem_backend.py:

* a decoder stack whose parent hands every block DIFFERENT arguments (alternating attention masks, a per-layer scale
  computed between the blocks) -- the whole-model statistics are reproduced by the parent walk and NOT by the hand-over
  of the first call's arguments;
* every algorithm of the path takes `layerwise` through quantize() / calibrate(), with the reference's refusals;
* checkpoint windows (`save_every`), resume inside a window, configuration drift;
* LIVE against the reference's own layerwise_calibrate on the same stack (build container only)."""

import copy
import json
import os
import sys

import pytest
import torch

import _moa_import
import hostmem_backend
from conftest import GOLDEN

moa = _moa_import.load()
from model_optimizer_amd import layerwise, model_calib, model_quant  # noqa: E402

sys.path.insert(0, GOLDEN)
import ref_shim  # noqa: E402


@pytest.fixture
def hostmem(monkeypatch):
    return hostmem_backend.install(monkeypatch, moa)


class Block(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.fc1 = torch.nn.Linear(d, 2 * d, bias=False)
        self.fc2 = torch.nn.Linear(2 * d, d, bias=False)

    def forward(self, h, mask=None, scale=1.0):
        x = h if mask is None else h * mask
        return (h + scale * self.fc2(torch.nn.functional.gelu(self.fc1(x))),)


class Stack(torch.nn.Module):
    """The parent computes per-layer arguments BETWEEN the blocks: blocks alternate between two masks (the sliding /
    full attention pattern of Gemma-2-style stacks) and take a scale that depends on the layer index."""

    def __init__(self, d=64, n=5):
        super().__init__()
        self.embed = torch.nn.Linear(d, d, bias=False)
        self.layers = torch.nn.ModuleList([Block(d) for _ in range(n)])
        self.head = torch.nn.Linear(d, 8, bias=False)
        self.calls = 0

    def forward(self, x):
        self.calls += 1
        h = self.embed(x)
        even = (torch.arange(h.shape[-1]) % 2 == 0).to(h.dtype)
        masks = (even, 1.0 - 0.5 * even)
        for i, layer in enumerate(self.layers):
            h = layer(h, mask=masks[i % 2], scale=0.25 * (i + 1))[0]
        return self.head(h)


def _setup(cfg, dtype=torch.float32, seed=0):
    torch.manual_seed(seed)
    model = Stack().to(dtype)
    batches = [torch.randn(6, 64).to(dtype) * (1 + i) for i in range(3)]
    moa.nn.replace_quant_module(model)
    q_cfg = dict(cfg["quant_cfg"])
    q_cfg["*embed*"] = {"enable": False}
    q_cfg["*head*"] = {"enable": False}
    model_quant.set_quantizer_by_cfg(model, q_cfg)
    return model, batches


def _amax(model):
    return {n: q._amax.detach().float().clone() for n, q in model.named_modules()
            if isinstance(q, moa.TensorQuantizer) and hasattr(q, "_amax")}


def _same(a, b):
    assert set(a) == set(b) and a
    return [k for k in a if not torch.equal(a[k], b[k])]


@pytest.mark.parametrize("cfg", ["FP8_DEFAULT_CFG", "INT8_DEFAULT_CFG"])
def test_parent_walk_reproduces_the_whole_model_statistics(hostmem, cfg):
    model, batches = _setup(getattr(model_quant, cfg))
    loop = lambda m: [m(b) for b in batches]  # noqa: E731
    whole = copy.deepcopy(model)
    model_calib.max_calibrate(whole, loop)
    handed = copy.deepcopy(model)
    assert layerwise.layerwise_calibrate(model, loop, model_calib.max_calibrate) == 5
    assert not _same(_amax(whole), _amax(model)), "parent walk differs from the whole-model pass"
    # the first call's arguments replayed for every block miss the per-layer masks / scales: inputs of layers >= 1 differ
    layerwise.layerwise_calibrate(handed, loop, model_calib.max_calibrate, capture="handover")
    off = _same(_amax(whole), _amax(handed))
    assert off and all(".layers.0." not in k for k in off), off
    # nothing is left on the instances
    assert all("forward" not in m.__dict__ for m in model.modules())


def test_the_walk_skips_finished_layers_and_runs_each_block_once_per_batch(hostmem):
    """Cost of the parent mode: per layer one run of the parent per batch in which ONLY the previous block computes
    (finished blocks return meta placeholders), plus the calibration function's own replays."""
    model, batches = _setup(model_quant.FP8_DEFAULT_CFG)
    ran = {i: 0 for i in range(5)}
    for i, layer in enumerate(model.layers):  # the block's first linear: counts COMPUTE, not calls of the block
        layer.fc1.register_forward_hook(lambda m, a, o, i=i: ran.__setitem__(i, ran[i] + 1))
    model.calls = 0
    layerwise.layerwise_calibrate(model, lambda m: [m(b) for b in batches], model_calib.max_calibrate)
    assert model.calls == 5 * len(batches)  # one parent run per layer per batch
    # a block computes for its calibration (3 batches) and once more per batch to feed its successor; the last does not
    assert [ran[i] for i in range(5)] == [6, 6, 6, 6, 3]


def test_qdq_from_the_previous_layer_feeds_quantized_outputs(hostmem):
    model, batches = _setup(model_quant.INT8_DEFAULT_CFG)
    loop = lambda m: [m(b) for b in batches]  # noqa: E731
    plain = copy.deepcopy(model)
    layerwise.layerwise_calibrate(plain, loop, model_calib.max_calibrate)
    layerwise.layerwise_calibrate(model, loop, model_calib.max_calibrate, get_qdq_activations_from_prev_layer=True)
    off = _same(_amax(plain), _amax(model))
    assert off and all(".layers.0." not in k for k in off)
    # what it must equal: a sequential whole-model calibration in which every finished layer quantizes
    seq2, _ = _setup(model_quant.INT8_DEFAULT_CFG)
    for i, layer in enumerate(seq2.layers):
        others = [q for j, l2 in enumerate(seq2.layers) if j > i for q in l2.modules() if isinstance(q, moa.TensorQuantizer)]
        saved = [q._disabled for q in others]
        for q in others:
            q._disabled = True
        model_calib.max_calibrate(layer, lambda _l: loop(seq2))
        for q, d in zip(others, saved):
            q._disabled = d
    assert not _same(_amax(seq2), _amax(model))


def test_inconsistent_forward_loops_are_refused(hostmem):
    model, batches = _setup(model_quant.FP8_DEFAULT_CFG)
    n = {"runs": 0}

    def shrinking(m):
        n["runs"] += 1
        for b in batches[: 3 if n["runs"] == 1 else 2]:
            m(b)

    with pytest.raises(RuntimeError, match="less often"):
        layerwise.layerwise_calibrate(model, shrinking, model_calib.max_calibrate)
    assert all("forward" not in m.__dict__ for m in model.modules())
    n["runs"] = 0

    def growing(m):
        n["runs"] += 1
        for b in batches[: 2 if n["runs"] == 1 else 3]:
            m(b)

    with pytest.raises(RuntimeError, match="more often"):
        layerwise.layerwise_calibrate(model, growing, model_calib.max_calibrate)


def test_a_parent_that_computes_between_blocks_gets_a_clear_error(hostmem):
    class Glue(Stack):
        def forward(self, x):
            h = self.embed(x)
            for layer in self.layers:
                h = layer(h)[0]
                if float(h.abs().max()) > 1e30:  # real-device arithmetic on the hidden state between two blocks
                    h = h.clamp(-1e30, 1e30)
            return self.head(h)

    torch.manual_seed(0)
    model = Glue()
    moa.nn.replace_quant_module(model)
    model_quant.set_quantizer_by_cfg(model, dict(model_quant.FP8_DEFAULT_CFG["quant_cfg"]))
    batches = [torch.randn(4, 64)]
    with pytest.raises(RuntimeError, match="capture='handover'"):
        layerwise.layerwise_calibrate(model, lambda m: [m(b) for b in batches], model_calib.max_calibrate)
    assert all("forward" not in m.__dict__ for m in model.modules())
    assert layerwise.layerwise_calibrate(model, lambda m: [m(b) for b in batches], model_calib.max_calibrate,
                                         capture="handover") == 5


# ---------------------------------------------------------------------------------- quantize(algorithm.layerwise)
@pytest.mark.parametrize("method,preset", [("max", "FP8_DEFAULT_CFG"), ("mse", "INT8_DEFAULT_CFG"),
                                           ("smoothquant", "INT8_SMOOTHQUANT_CFG"), ("awq_lite", "INT4_AWQ_CFG"),
                                           ("awq_clip", "INT4_AWQ_CFG")])
def test_every_algorithm_takes_the_layerwise_option(hostmem, method, preset):
    """mode.py:215-277: the wrapper is the same for all algorithms.  Amax-only algorithms and SmoothQuant / AWQ see the
    activations of the un-quantized predecessors, so the per-layer run equals the whole-model run."""
    model, batches = _setup(getattr(model_quant, preset))
    loop = lambda m: [m(b) for b in batches]  # noqa: E731
    whole = copy.deepcopy(model)
    model_quant.calibrate(whole, {"method": method}, loop)
    model_quant.calibrate(model, {"method": method, "layerwise": {"enable": True}}, loop)
    assert not _same(_amax(whole), _amax(model))
    for (n, a), (_, b) in zip(whole.named_parameters(), model.named_parameters()):
        assert torch.equal(a, b), n
    for (n, a), (_, b) in zip(whole.named_modules(), model.named_modules()):
        if isinstance(a, moa.TensorQuantizer) and a.pre_quant_scale is not None:
            assert torch.equal(a.pre_quant_scale, b.pre_quant_scale), n


def test_layerwise_option_refusals(hostmem, tmp_path):
    model, batches = _setup(model_quant.INT4_AWQ_CFG)
    loop = lambda m: [m(b) for b in batches]  # noqa: E731
    with pytest.raises(ValueError, match="requires layerwise.enable=True"):
        model_quant.calibrate(model, {"method": "max", "layerwise": {"checkpoint_dir": str(tmp_path)}}, loop)
    for method in ("awq_lite", "smoothquant", "gptq"):
        with pytest.raises(ValueError, match="mutates layer weights"):
            model_quant.calibrate(model, {"method": method, "layerwise": {"enable": True, "calib_mutates_weights": False}}, loop)
    with pytest.raises(ValueError, match="forward_loop is required"):
        model_quant.calibrate(model, {"method": "max", "layerwise": True}, None)
    with pytest.raises(ValueError, match="unknown layerwise option"):
        model_quant.calibrate(model, {"method": "max", "layerwise": {"enable": True, "every": 2}}, loop)
    # amax-only algorithms may leave the weights out of their checkpoints
    model_quant.calibrate(model, {"method": "max", "layerwise": {"enable": True, "calib_mutates_weights": False,
                                                                 "checkpoint_dir": str(tmp_path)}}, loop)
    blob = torch.load(os.path.join(tmp_path, "layer_0000.pt"), weights_only=False)
    assert blob["weights"] is None and blob["output"][0] == "tuple"


# ------------------------------------------------------------------------------------------- checkpoint windows
def test_save_every_commits_windows_and_resumes_at_their_start(hostmem, tmp_path):
    model, batches = _setup(model_quant.FP8_DEFAULT_CFG)
    loop = lambda m: [m(b) for b in batches]  # noqa: E731
    ref = copy.deepcopy(model)
    layerwise.layerwise_calibrate(ref, loop, model_calib.max_calibrate)
    calls = {"n": 0}

    def flaky(layer, lp, **kw):
        if calls["n"] == 3:  # layers 0, 1 (a committed window) and 2 (inside the next window) are done
            raise KeyboardInterrupt
        calls["n"] += 1
        model_calib.max_calibrate(layer, lp, **kw)

    m1 = copy.deepcopy(model)
    with pytest.raises(KeyboardInterrupt):
        layerwise.layerwise_calibrate(m1, loop, flaky, checkpoint_dir=str(tmp_path), save_every=2)
    assert all("forward" not in m.__dict__ for m in m1.modules())
    with open(os.path.join(tmp_path, "manifest.json")) as f:
        man = json.load(f)
    assert man == {"num_layers": 5, "completed": 2, "save_every": 2, "calib_mutates_weights": True}
    assert os.path.exists(os.path.join(tmp_path, "layer_0002.pt"))  # written, not committed
    assert torch.load(os.path.join(tmp_path, "next_inputs.pt"), weights_only=False)["for_layer"] == 2
    m2 = copy.deepcopy(model)
    assert layerwise.layerwise_calibrate(m2, loop, model_calib.max_calibrate, checkpoint_dir=str(tmp_path), save_every=2) == 3
    assert not _same(_amax(ref), _amax(m2))
    with open(os.path.join(tmp_path, "manifest.json")) as f:
        assert json.load(f)["completed"] == 5  # the last layer always commits
    # settings other than the directory's are refused
    for kw, key in (({"save_every": 1}, "save_every"), ({"save_every": 2, "calib_mutates_weights": False}, "calib_mutates_weights")):
        with pytest.raises(ValueError, match=f"Checkpoint {key} mismatch"):
            layerwise.layerwise_calibrate(copy.deepcopy(model), loop, model_calib.max_calibrate, checkpoint_dir=str(tmp_path), **kw)
    short = copy.deepcopy(model)
    del short.layers[4]
    with pytest.raises(ValueError, match="Checkpoint num_layers mismatch"):
        layerwise.layerwise_calibrate(short, loop, model_calib.max_calibrate, checkpoint_dir=str(tmp_path), save_every=2)


def test_resume_of_a_weight_mutating_calibration_equals_the_uninterrupted_run(hostmem, tmp_path):
    """AWQ folds scales into the weights: the finished layers' weights come back from the checkpoint, the walk skips
    them (placeholders of the stored output shapes) and continues on the stored inputs."""
    model, batches = _setup(model_quant.INT4_AWQ_CFG)
    loop = lambda m: [m(b) for b in batches]  # noqa: E731
    ref = copy.deepcopy(model)
    layerwise.layerwise_calibrate(ref, loop, model_calib.awq_lite)
    calls = {"n": 0}

    def flaky(layer, lp, **kw):
        if calls["n"] == 2:
            raise KeyboardInterrupt
        calls["n"] += 1
        model_calib.awq_lite(layer, lp, **kw)

    with pytest.raises(KeyboardInterrupt):
        layerwise.layerwise_calibrate(copy.deepcopy(model), loop, flaky, checkpoint_dir=str(tmp_path))
    m2 = copy.deepcopy(model)
    ran = {i: 0 for i in range(5)}
    for i, layer in enumerate(m2.layers):
        layer.fc1.register_forward_hook(lambda m, a, o, i=i: ran.__setitem__(i, ran[i] + 1))
    assert layerwise.layerwise_calibrate(m2, loop, model_calib.awq_lite, checkpoint_dir=str(tmp_path)) == 3
    assert ran[0] == 0 and ran[1] == 0, "finished layers must not compute after a resume"
    assert not _same(_amax(ref), _amax(m2))
    for (n, a), (_, b) in zip(ref.named_parameters(), m2.named_parameters()):
        assert torch.equal(a, b), n


# ------------------------------------------------------------------------------------------- live, against the reference
@pytest.mark.skipif(not ref_shim.reference_available(), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("preset,algorithm,dtype", [
    ("FP8_DEFAULT_CFG", {"method": "max"}, torch.float32),
    ("INT8_DEFAULT_CFG", {"method": "max", "layerwise": {"get_qdq_activations_from_prev_layer": True}}, torch.float32),
    ("INT8_DEFAULT_CFG", {"method": "mse"}, torch.float32),
    ("INT8_SMOOTHQUANT_CFG", {"method": "smoothquant"}, torch.bfloat16),
    ("INT4_AWQ_CFG", {"method": "awq_lite"}, torch.bfloat16)])  # 16-bit weights: as the other live AWQ comparisons
def test_layerwise_flows_equal_the_reference_run_live(hostmem, preset, algorithm, dtype):
    """The reference's own layerwise_calibrate (model_calib.py:2051-2188) on the same stack: its decoder-layer registry
    gets a discoverer for the stack, then `mtq.quantize(..., algorithm {..., layerwise: enable})` on both sides."""
    ref_shim.install()
    import modelopt.torch.quantization as mtq
    from modelopt.torch.quantization.nn import TensorQuantizer as RefQuantizer
    from modelopt.torch.quantization.utils.layerwise_calib import LayerActivationCollector

    LayerActivationCollector.register_decoder_layer_support(_is_stack, _stack_layers)
    algo = copy.deepcopy(algorithm)
    algo["layerwise"] = {**algo.get("layerwise", {}), "enable": True}
    torch.manual_seed(3)
    ours = Stack().to(dtype)
    theirs = copy.deepcopy(ours)
    batches = [(torch.randn(6, 64) * (1 + i)).to(dtype) for i in range(3)]
    loop = lambda m: [m(b) for b in batches]  # noqa: E731
    cfg = copy.deepcopy(getattr(mtq, preset))
    cfg["quant_cfg"] = _with_disabled(cfg["quant_cfg"], ("*embed*", "*head*"))
    cfg["algorithm"] = copy.deepcopy(algo)
    mtq.quantize(theirs, cfg, loop)
    want = {n: m._amax.detach().float().clone() for n, m in theirs.named_modules()
            if isinstance(m, RefQuantizer) and getattr(m, "_amax", None) is not None}
    mine = copy.deepcopy(getattr(model_quant, preset))
    mine["quant_cfg"] = _with_disabled(mine["quant_cfg"], ("*embed*", "*head*"))
    mine["algorithm"] = copy.deepcopy(algo)
    moa.quantize(ours, mine, loop)
    got = _amax(ours)
    assert set(got) == set(want) and len(got) >= 10
    for k in want:
        assert torch.equal(got[k].reshape(-1), want[k].reshape(-1)), k
    for (n, a), (_, b) in zip(theirs.named_parameters(), ours.named_parameters()):
        assert torch.equal(a, b), n


def _is_stack(model):
    return isinstance(model, Stack)


def _stack_layers(model):
    return model.layers


def _with_disabled(quant_cfg, patterns):
    off = {"enable": False}
    if isinstance(quant_cfg, dict):
        return {**quant_cfg, **{p: off for p in patterns}}
    return list(quant_cfg) + [{"quantizer_name": p, **off} for p in patterns]


# ------------------------------------------------------------------------------------------- Hugging Face stacks
@pytest.mark.parametrize("arch", ["gemma2", "llama"])
def test_hugging_face_stacks_walked_through_their_parent(hostmem, arch):
    """Gemma-2 alternates sliding-window and full attention: its parent hands every block the mask of the block's own
    type.  The parent walk reproduces the whole-model statistics; the hand-over of the first block's arguments gives
    the full-attention blocks a sliding mask.  Llama calls every block alike: both modes agree there."""
    import transformers as tf

    cfg = dict(hidden_size=64, intermediate_size=128, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
               vocab_size=96, max_position_embeddings=64)
    torch.manual_seed(11)
    if arch == "gemma2":
        model = tf.Gemma2ForCausalLM(tf.Gemma2Config(architectures=["Gemma2ForCausalLM"], head_dim=16, sliding_window=4, **cfg))
    else:
        model = tf.LlamaForCausalLM(tf.LlamaConfig(architectures=["LlamaForCausalLM"], **cfg))
    model = model.float().eval()
    batches = [torch.randint(0, 96, (2, 20), generator=torch.Generator().manual_seed(5 + i)) for i in range(2)]
    loop = lambda m: [m(b) for b in batches]  # noqa: E731
    quant_cfg = copy.deepcopy(model_quant.FP8_DEFAULT_CFG)
    whole = moa.quantize(copy.deepcopy(model), quant_cfg, loop)
    walked = moa.quantize(copy.deepcopy(model), {**quant_cfg, "algorithm": {"method": "max", "layerwise": {"enable": True}}}, loop)
    handed = moa.quantize(copy.deepcopy(model), {**quant_cfg, "algorithm": {"method": "max", "layerwise": {"enable": True, "capture": "handover"}}}, loop)
    assert not _same(_amax(whole), _amax(walked))
    off = _same(_amax(whole), _amax(handed))
    if arch == "gemma2":
        assert off and all(".layers.0." not in k for k in off), off
    else:
        assert not off
    with torch.no_grad():  # the walked model is left as a whole-model calibration leaves it
        assert torch.equal(whole(batches[0]).logits, walked(batches[0]).logits)
