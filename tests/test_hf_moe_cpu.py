"""Sparse MoE blocks with separate expert modules (hf_moe.py = the reference's _QuantSparseSequentialMoe rules) on CPU:
shared input amax after max calibration, experts without tokens, `sync_expert_weight_amax`, `moe_calib_experts_ratio` --
LIVE against the reference on a block both sides detect structurally."""

import copy
import sys

import pytest
import torch

import _moa_import
import hostmem_backend
from conftest import GOLDEN

moa = _moa_import.load()
from model_optimizer_amd import hf_moe, model_quant  # noqa: E402

sys.path.insert(0, GOLDEN)
import ref_shim  # noqa: E402


class Router(torch.nn.Module):
    """The (logits, scores, indices) router of transformers >= 5 with `top_k` / `num_experts` on it."""

    def __init__(self, d, n, k):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(n, d) * 0.5)
        self.register_buffer("offset", torch.zeros(n))
        self.top_k, self.num_experts = k, n

    def forward(self, x):
        logits = torch.nn.functional.linear(x, self.weight) + self.offset.to(x.dtype)
        scores = logits.float().softmax(-1)
        top, idx = torch.topk(scores, self.top_k, dim=-1)
        return logits, (top / top.sum(-1, keepdim=True)).to(x.dtype), idx


class Expert(torch.nn.Module):
    def __init__(self, d):
        super().__init__()
        self.up = torch.nn.Linear(d, 2 * d, bias=False)
        self.down = torch.nn.Linear(2 * d, d, bias=False)

    def forward(self, x):
        return self.down(torch.nn.functional.silu(self.up(x)))


class MoeBlock(torch.nn.Module):
    def __init__(self, d=32, n=6, k=2):
        super().__init__()
        self.gate = Router(d, n, k)
        self.experts = torch.nn.ModuleList([Expert(d) for _ in range(n)])

    def forward(self, hidden_states):
        x = hidden_states.reshape(-1, hidden_states.shape[-1])
        _, weights, idx = self.gate(x)
        out = torch.zeros_like(x)
        for e, expert in enumerate(self.experts):
            tok, slot = torch.where(idx == e)
            if tok.numel():
                out.index_add_(0, tok, expert(x[tok]) * weights[tok, slot, None])
        return out.reshape(hidden_states.shape)


class Net(torch.nn.Module):
    def __init__(self, d=32):
        super().__init__()
        self.inp = torch.nn.Linear(d, d, bias=False)
        self.moe = MoeBlock(d)
        self.out = torch.nn.Linear(d, d, bias=False)

    def forward(self, x):
        return self.out(self.moe(self.inp(x)))


def _net(starve=True):
    torch.manual_seed(21)
    net = Net()
    if starve:  # expert 5 never wins: it sees no calibration token
        net.moe.gate.offset[5] = -1e4
    return net


def _batches():
    g = torch.Generator().manual_seed(3)
    return [torch.randn(2, 9, 32, generator=g) * (1 + i) for i in range(3)]


def _amax(model, cls):
    return {n: q._amax.detach().float().clone() for n, q in model.named_modules() if isinstance(q, cls) and getattr(q, "_amax", None) is not None}


@pytest.fixture
def hostmem(monkeypatch):
    return hostmem_backend.install(monkeypatch, moa)


def test_experts_share_their_input_amax_and_starved_experts_get_a_weight_amax(hostmem):
    net, batches = _net(), _batches()
    assert [n for n, _ in hf_moe.sparse_moe_blocks(net)] == ["moe"]
    seen = [0] * 6
    for i, e in enumerate(net.moe.experts):
        e.register_forward_hook(lambda m, a, o, i=i: seen.__setitem__(i, seen[i] + 1))
    moa.quantize(net, model_quant.FP8_DEFAULT_CFG, lambda m: [m(b) for b in batches])
    assert seen[5] == 0 and all(seen[:5]), seen
    ups = [e.up.input_quantizer.amax for e in net.moe.experts]
    assert all(a is not None and torch.equal(a, ups[0]) for a in ups), "one input amax per projection for the block"
    downs = [e.down.input_quantizer.amax for e in net.moe.experts]
    assert all(torch.equal(a, downs[0]) for a in downs) and not torch.equal(ups[0], downs[0])
    ws = [e.up.weight_quantizer.amax for e in net.moe.experts]
    assert all(w is not None for w in ws) and len({float(w) for w in ws}) > 1, "weight amax stays per expert by default"
    for e in net.moe.experts:
        assert torch.equal(e.up.weight_quantizer.amax.float(), e.up.weight.abs().max().float())
    shared = copy.deepcopy(_net())
    moa.quantize(shared, {**model_quant.FP8_DEFAULT_CFG, "algorithm": {"method": "max", "sync_expert_weight_amax": True}},
                 lambda m: [m(b) for b in batches])
    ws = [e.up.weight_quantizer.amax for e in shared.moe.experts]
    assert all(torch.equal(w, ws[0]) for w in ws)
    assert float(ws[0]) == max(float(e.up.weight.abs().max()) for e in shared.moe.experts)


def test_widened_routing_calibrates_more_experts_and_leaves_the_output_alone(hostmem):
    net, batches = _net(starve=False), _batches()
    plain = copy.deepcopy(net)
    with torch.no_grad():
        want = plain(batches[0])
    cfg = {**model_quant.FP8_DEFAULT_CFG, "algorithm": {"method": "max", "moe_calib_experts_ratio": 0.5}}
    moa.quantize(net, cfg, lambda m: [m(b) for b in batches])
    assert net.moe._moe_calib_experts_ratio == 0.5 and net.moe.gate.top_k == 2
    n_tokens = sum(b.shape[0] * b.shape[1] for b in batches[1:])  # (the batch that sets the counter up is not counted)
    assert int(net.moe.expert_token_count.sum()) == n_tokens * 3, "tokens counted at the widened top-k (3 of 6 experts)"
    ups = [float(e.up.input_quantizer.amax) for e in net.moe.experts]
    assert len(set(ups)) > 1, "with a ratio every expert keeps its own statistics"
    for q in net.modules():
        if isinstance(q, moa.TensorQuantizer):
            q.disable()
    with torch.no_grad():
        assert torch.equal(net(batches[0]), want), "outside calibration the block routes as configured"
    # a deep copy runs on ITS OWN experts (the shadowed forward is rebound to the copied block)
    twin = copy.deepcopy(net)
    assert twin.moe.forward.block is twin.moe and twin.moe.gate._forward_hooks and \
        all(h.block is twin.moe for h in twin.moe.gate._forward_hooks.values())
    with torch.no_grad():
        for p in twin.moe.experts.parameters():
            p.zero_()
        assert not torch.equal(net(batches[0]), twin(batches[0])) and torch.equal(net(batches[0]), want)
    assert hf_moe.set_moe_calib_experts_ratio(net, None) == 1 and "forward" not in net.moe.__dict__
    with pytest.raises(AssertionError, match="Invalid moe_calib_experts_ratio"):
        hf_moe.set_moe_calib_experts_ratio(net, 1.5)


@pytest.mark.skipif(not ref_shim.reference_available(), reason="reference checkout not present (GPU box)")
@pytest.mark.parametrize("algorithm", [{"method": "max"}, {"method": "max", "sync_expert_weight_amax": True},
                                       {"method": "max", "moe_calib_experts_ratio": 0.5},
                                       {"method": "max", "moe_calib_experts_ratio": 1.0}])
def test_moe_block_calibration_equals_the_reference_live(hostmem, algorithm):
    ref_shim.install()
    import modelopt.torch.quantization as mtq
    from modelopt.torch.quantization.nn import TensorQuantizer as RefQuantizer

    batches = _batches()
    loop = lambda m: [m(b) for b in batches]  # noqa: E731
    theirs = _net()
    cfg = copy.deepcopy(mtq.FP8_DEFAULT_CFG)
    cfg["algorithm"] = copy.deepcopy(algorithm)
    mtq.quantize(theirs, cfg, loop)
    assert type(theirs.moe).__name__.startswith("Quant"), "the reference did not take the block for a sparse MoE"
    want = _amax(theirs, RefQuantizer)
    ours = _net()
    mine = copy.deepcopy(model_quant.FP8_DEFAULT_CFG)
    mine["algorithm"] = copy.deepcopy(algorithm)
    moa.quantize(ours, mine, loop)
    got = _amax(ours, moa.TensorQuantizer)
    assert set(got) == set(want) and len(want) >= 26, set(got) ^ set(want)
    for k in want:
        assert torch.equal(got[k].reshape(-1), want[k].reshape(-1)), k
    if algorithm.get("moe_calib_experts_ratio", 1.0) < 1.0:
        assert torch.equal(ours.moe.expert_token_count, theirs.moe.expert_token_count)
    with torch.no_grad():
        assert torch.equal(ours(batches[0]), theirs(batches[0]))
