"""forward_loop.create_forward_loop / run_batch (dataset_utils.py:963-1255): the calibration loop of the path's callers.
LIVE against the reference's `_process_batch` for the out-of-memory splitting (same sub-batches, same order, same
learned size), use_cache handling, and the loop driving a real calibration."""

import copy
import sys
import types

import pytest
import torch

import _moa_import
import hostmem_backend
from conftest import GOLDEN

moa = _moa_import.load()
from model_optimizer_amd import forward_loop as fl  # noqa: E402

sys.path.insert(0, GOLDEN)
import ref_shim  # noqa: E402


class _Probe:
    """An inference method that runs out of memory above a limit which may change from call to call."""

    def __init__(self, limits):
        self.limits, self.calls, self.attempts = list(limits), [], 0

    def __call__(self, input_ids=None, attention_mask=None):
        limit = self.limits[min(self.attempts, len(self.limits) - 1)]
        self.attempts += 1
        if input_ids.shape[0] > limit:
            raise torch.cuda.OutOfMemoryError("probe")
        assert attention_mask is None or attention_mask.shape[0] == input_ids.shape[0]
        self.calls.append(input_ids[:, 0].tolist())


def _batches(sizes, none_masks=False):
    out, start = [], 0
    for n in sizes:
        ids = torch.arange(start, start + n).reshape(n, 1).repeat(1, 3)
        out.append({"input_ids": ids, "attention_mask": None if none_masks and n % 2 else torch.ones(n, 3)})
        start += n
    return out


CASES = [([16, 16, 7], [5]), ([9], [4]), ([12, 12], [100, 100, 3]), ([8, 8, 8], [8, 3, 3, 3, 8]), ([5, 1, 6], [2]),
         ([32], [32]), ([10, 10], [6, 6, 2, 6])]


@pytest.mark.parametrize("sizes,limits", CASES)
def test_out_of_memory_splitting_visits_the_reference_sub_batches_live(sizes, limits):
    if not ref_shim.reference_available():
        pytest.skip("reference checkout not present (GPU box)")
    ref_shim.install()
    from modelopt.torch.utils import dataset_utils as ref

    mine, theirs = _Probe(limits), _Probe(limits)
    a = b = None
    with pytest.warns(UserWarning) if any(n > limits[0] for n in sizes) else _nullcontext():
        for batch in _batches(sizes):
            a = fl.run_batch(batch, mine, a)
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for batch in _batches(sizes):
            b = ref._process_batch(batch, theirs, b)
    assert mine.calls == theirs.calls and mine.attempts == theirs.attempts and a == b


class _nullcontext:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def test_splitting_properties_without_the_reference():
    probe = _Probe([5])
    known = None
    with pytest.warns(UserWarning, match="out of memory with batch size 16, trying with batch size 8"):
        for batch in _batches([16, 16, 7], none_masks=True):  # (a None entry stays None in every piece)
            known = fl.run_batch(batch, probe, known)
    seen = [s for call in probe.calls for s in call]
    assert seen == list(range(39)), "every sample exactly once, in order"
    assert known == 4 and max(len(c) for c in probe.calls) <= 5
    with pytest.raises(AssertionError, match="single sample"):
        fl.run_batch(_batches([1])[0], _Probe([0]))
    with pytest.raises(AssertionError, match="must be tensors or None"):
        fl.run_batch({"input_ids": torch.zeros(2, 3), "meta": "text"}, _Probe([9]))
    fl.run_batch({"input_ids": torch.zeros(2, 3, dtype=torch.long), "attention_mask": None}, _Probe([9]))


def test_use_cache_is_off_inside_the_loop_and_put_back():
    class M(torch.nn.Module):
        def __init__(self, config):
            super().__init__()
            self.config, self.seen = config, []

        def forward(self, input_ids=None):
            self.seen.append((getattr(self.config, "use_cache", "absent"), getattr(getattr(self.config, "text_config", None), "use_cache", "absent"),
                              torch.is_grad_enabled()))

    text = types.SimpleNamespace()
    m = M(types.SimpleNamespace(use_cache=True, text_config=text))
    loop = fl.create_forward_loop(dataloader=[{"input_ids": torch.zeros(2, 4, dtype=torch.long)}] * 2)
    loop(m)
    assert m.seen == [(False, False, False)] * 2
    assert m.config.use_cache is True and not hasattr(text, "use_cache")
    loop(M(None))  # a model without a config
    with pytest.raises(ValueError, match="pass `dataloader=`"):
        fl.create_forward_loop(model=m, dataset_name="cnn_dailymail")

    class T5ish(M):
        def generate(self, input_ids=None):
            self.seen.append("generate")

    t = T5ish(types.SimpleNamespace())
    loop(t)
    assert t.seen == ["generate", "generate"] and not hasattr(t.config, "use_cache")


def test_the_loop_calibrates_a_hugging_face_model(monkeypatch):
    """create_forward_loop over synthetic token batches == the hand-written loop over the same batches."""
    import transformers as tf

    hostmem_backend.install(monkeypatch, moa)
    torch.manual_seed(5)
    cfg = tf.LlamaConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4, num_key_value_heads=2,
                         vocab_size=96, max_position_embeddings=64, architectures=["LlamaForCausalLM"])
    model = tf.LlamaForCausalLM(cfg).float().eval()
    batches = fl.synthetic_token_batches(96, num_samples=10, max_sample_length=12, batch_size=4)
    assert [b["input_ids"].shape for b in batches] == [(4, 12), (4, 12), (2, 12)]
    assert torch.equal(batches[0]["input_ids"], fl.synthetic_token_batches(96, 10, 12, 4)[0]["input_ids"])
    a = moa.quantize(copy.deepcopy(model), moa.model_quant.FP8_DEFAULT_CFG, fl.create_forward_loop(dataloader=batches))
    b = moa.quantize(copy.deepcopy(model), moa.model_quant.FP8_DEFAULT_CFG, lambda m: [m(**x) for x in batches])
    amax = lambda mod: {n: q._amax.clone() for n, q in mod.named_modules() if isinstance(q, moa.TensorQuantizer) and hasattr(q, "_amax")}  # noqa: E731
    x, y = amax(a), amax(b)
    assert set(x) == set(y) and len(x) > 20 and all(torch.equal(x[k], y[k]) for k in x)


class _FakeGpu:
    """torch.cuda's memory queries over a scripted device: every model call of batch b takes b * per_sample bytes (kept
    as the allocator's peak), calls above `oom_above` samples run out of memory."""

    def __init__(self, total, used, per_sample, oom_above):
        self.total, self.free, self.per_sample, self.oom_above, self.peak, self.calls = total, total - used, per_sample, oom_above, 0, []

    def install(self, mp):
        mp.setattr(torch.cuda, "get_device_properties", lambda d=0: types.SimpleNamespace(total_memory=self.total))
        mp.setattr(torch.cuda, "device_count", lambda: 1)
        mp.setattr(torch.cuda, "mem_get_info", lambda d=0: (self.free, self.total))
        mp.setattr(torch.cuda, "max_memory_allocated", lambda d=0: self.peak)
        mp.setattr(torch.cuda, "empty_cache", lambda: None)

    def model(self):
        gpu = self

        class M(torch.nn.Module):
            device = torch.device("cpu")

            def forward(self, x):
                gpu.calls.append(x.shape[0])
                if x.shape[0] > gpu.oom_above:
                    raise torch.cuda.OutOfMemoryError("probe")
                gpu.peak = max(gpu.peak, x.shape[0] * gpu.per_sample)
                gpu.free = min(gpu.free, gpu.total - 1000 - x.shape[0] * gpu.per_sample) if x.shape[0] == 1 else gpu.free

        return M()


@pytest.mark.parametrize("total,used,per_sample,oom_above,want", [
    (100_000, 1000, 10_000, 10**9, 8), (100_000, 1000, 30_000, 10**9, 2), (100_000, 1000, 60_000, 10**9, 1),
    (10**9, 1000, 1000, 100, 60), (10**9, 1000, 100, 10**9, 512)])
def test_get_max_batch_size_over_a_scripted_device(monkeypatch, total, used, per_sample, oom_above, want):
    gpu = _FakeGpu(total, used, per_sample, oom_above)
    gpu.install(monkeypatch)
    got = fl.get_max_batch_size(gpu.model(), max_sample_length=8)
    assert got == want, (got, gpu.calls)
    if ref_shim.reference_available():  # the reference's probe over the same script
        ref_shim.install()
        from modelopt.torch.utils import dataset_utils as ref

        twin = _FakeGpu(total, used, per_sample, oom_above)
        twin.install(monkeypatch)
        assert ref.get_max_batch_size(twin.model(), max_sample_length=8) == got and twin.calls == gpu.calls
