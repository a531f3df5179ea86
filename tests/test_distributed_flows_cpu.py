"""world_size-2 gloo END-TO-END tests of the data-parallel calibration flows on the host-memory stand-in
backend: every rank holds a replica of the model and calibrates on its share of the batches (rank r: batches r, r+2, ...),
the weight-side statistics are dealt round-robin over the ranks, and the result must equal a SINGLE-rank run over all
batches -- the reference's property (tests/unit/torch/quantization/test_dist.py:27-47: after quantize every amax equals
its all_reduce(MAX)), extended to histograms (SUM of int64 counts), AWQ activation scales (average,
model_calib.py:1588-1594) and the chosen alpha (per-alpha losses SUMmed so every rank takes the same decision).

Each worker first computes the single-rank result on its own (before the process group exists), then joins the group and
runs the sharded flow.  On the GPU node the same code runs over RCCL (backend "nccl")."""

import copy
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _moa_import

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class MLP(torch.nn.Module):
    def __init__(self, d=128, h=256, dtype=torch.float32, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.fc1 = torch.nn.Linear(d, h, bias=False)
        self.fc2 = torch.nn.Linear(h, d, bias=True)
        self.fc3 = torch.nn.Linear(d, d, bias=False)
        with torch.no_grad():
            for lin in (self.fc1, self.fc2, self.fc3):
                lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * 0.05)
            self.fc2.bias.copy_(torch.randn(d, generator=g) * 0.01)
        self.to(dtype)

    def forward(self, x):
        return self.fc3(self.fc2(torch.relu(self.fc1(x))))


def _n_batches(world):
    """an EVEN split for every world (the reference averages the ranks' act-scale means: equal to the single-rank mean only then)"""
    return 4 if world <= 2 else 2 * world


def _batches(d, dtype, n=4):
    g = torch.Generator().manual_seed(5)
    ch = torch.exp(torch.randn(d, generator=g))
    ch[:3] *= 20
    out = [(torch.randn(24, d, generator=g) * ch).to(dtype) for _ in range(n)]
    out[2][0, 5] = 400.0  # the largest activation sits in a LATER batch of rank 0's share: histograms must grow
    out[1][3, 7] = 250.0  # and rank 1's first batch exceeds rank 0's first-batch range
    return out


def _install_backend(moa):
    sys.path.insert(0, HERE)
    import hostmem_backend

    mpatch = pytest.MonkeyPatch()
    hostmem_backend.install(mpatch, moa)
    from model_optimizer_amd import model_calib

    model_calib._WeightCacheBudget.host_bytes = 1 << 30  # Gram search on the host stand-in
    return mpatch


def _amaxes(model):
    return {n: m._amax.clone() for n, m in model.named_modules() if hasattr(m, "_amax")}


def _job_max_and_smoothquant(rank, world, moa, single):
    """FP8 per-tensor + INT8 per-channel max calibration and SmoothQuant: amax of weights (sharded over the ranks) and
    activations (batches sharded) bit-equal to the single-rank run."""
    mq = moa.model_quant
    for cfg in (mq.FP8_DEFAULT_CFG, mq.INT8_DEFAULT_CFG, mq.INT8_SMOOTHQUANT_CFG):
        batches = _batches(128, torch.float32, n=_n_batches(world))
        if single:
            model = moa.quantize(MLP(), copy.deepcopy(cfg), lambda m: [m(b) for b in batches])
        else:
            model = moa.quantize(MLP(), copy.deepcopy(cfg), lambda m: [m(b) for b in batches[rank::world]])
        yield {"amax": _amaxes(model), "w": {n: p.detach().clone() for n, p in model.named_parameters()},
               "pqs": {n: m._pre_quant_scale.clone() for n, m in model.named_modules() if hasattr(m, "_pre_quant_scale")}}


def _job_weight_side(rank, world, moa, single):
    """The weight-side flows dealt over the replicas (distributed.declare_data_parallel): 2:4 magnitude masks,
    SparseGPT (Hessians combined on their owner, masks broadcast), fold_weight (FP8, dynamic MXFP4 blocks and static INT4
    blocks) and the sharded checkpoint export (each rank packs and writes its own linears; the union of the shards is
    the single-rank checkpoint, byte for byte)."""
    import tempfile

    from safetensors.torch import load_file

    mq, sp, ex = moa.model_quant, moa.sparsity, moa.export
    batches = _batches(128, torch.float32, n=_n_batches(world))
    mine = batches if single else batches[rank::world]
    # 2:4 magnitude masks
    model = sp.sparsify(MLP(), "sparse_magnitude")
    yield {"mask": {n: m._weight_mask.clone() for n, m in model.named_modules() if hasattr(m, "_weight_mask")},
           "w": {n: p.detach().clone() for n, p in model.named_parameters()}}
    # SparseGPT: the Hessian over ALL batches whichever rank saw them
    model = sp.sparsify(MLP(), "sparsegpt", forward_loop=lambda m: [m(b) for b in mine])
    yield {"mask~": {n: m._weight_mask.clone() for n, m in model.named_modules() if hasattr(m, "_weight_mask")}}
    # fold_weight after calibration
    for preset in ("FP8_DEFAULT_CFG", "MXFP4_DEFAULT_CFG", "INT4_BLOCKWISE_WEIGHT_ONLY_CFG", "INT8_SMOOTHQUANT_CFG"):
        model = moa.quantize(MLP(), copy.deepcopy(getattr(mq, preset)), lambda m: [m(b) for b in mine])
        state = ex.export_state_dict(model, torch.float32)
        if not single:
            with tempfile.TemporaryDirectory() as d:
                box = [d]
                dist.broadcast_object_list(box, src=0)  # one directory for all ranks: rank 0's
                sub = os.path.join(box[0], preset)
                ex.save_checkpoint(state, sub, ex.hf_quant_config(model))
                dist.barrier()
                files = sorted(f for f in os.listdir(sub) if f.endswith(".safetensors"))
                assert files == [f"model-{r + 1:05d}-of-{world:05d}.safetensors" for r in range(world)], files
                import json

                index = json.load(open(os.path.join(sub, "model.safetensors.index.json")))["weight_map"]
                merged = {}
                for f in files:
                    part = load_file(os.path.join(sub, f))
                    assert not set(part) & set(merged)
                    assert all(index[k] == f for k in part)
                    merged.update(part)
                assert set(index) == set(merged)
                dist.barrier()
            state = merged
        else:
            state = {k: v.detach().cpu() for k, v in state.items()}
        mq.fold_weight(model)
        yield {"ckpt": state, "w": {n: p.detach().clone() for n, p in model.named_parameters()},
               "off": {n: torch.tensor(float(q.is_enabled)) for n, q in model.named_modules() if n.endswith("weight_quantizer")}}


def _job_undeclared(rank, world, moa, single):
    """No declare_data_parallel: ranks holding DIFFERENT weights (tensor parallel, FSDP) calibrate all of their own -- the
    amax of every weight quantizer is the MAX over the ranks' own values, never one rank's shard alone."""
    mq = moa.model_quant
    batches = _batches(128, torch.float32)
    if single:
        values = []
        for r in range(world):
            m = moa.quantize(MLP(seed=r), copy.deepcopy(mq.FP8_DEFAULT_CFG), lambda mm: [mm(b) for b in batches])
            values.append(_amaxes(m))
        yield {"amax": {k: torch.stack([v[k] for v in values]).amax(0) for k in values[0]}}
    else:
        m = moa.quantize(MLP(seed=rank), copy.deepcopy(mq.FP8_DEFAULT_CFG), lambda mm: [mm(b) for b in batches])
        yield {"amax": _amaxes(m)}


def _job_histogram(rank, world, moa, single):
    """Histogram calibrators on the activations (percentile and entropy): int64 counts, bin edges and the amax equal
    the single-rank run over all batches."""
    from model_optimizer_amd import model_calib

    mq = moa.model_quant
    for method, kw in (("percentile", {"percentile": 99.9}), ("entropy", {})):
        cfg = copy.deepcopy(mq.INT8_DEFAULT_CFG)
        cfg["quant_cfg"]["*input_quantizer"] = {"num_bits": 8, "axis": None, "calibrator": "histogram"}
        cfg["algorithm"] = None
        model = moa.quantize(MLP(), cfg, None)
        batches = _batches(128, torch.float32)
        mine = batches if single else batches[rank::world]
        model_calib.histogram_calibrate(model, lambda m: [m(b) for b in mine], method=method, **kw)
        cals = {n: m._calibrator for n, m in model.named_modules() if n.endswith("input_quantizer")}
        yield {"amax": _amaxes(model), "hist": {n: c._calib_hist.clone() for n, c in cals.items()},
               "edges": {n: c._calib_bin_edges.clone() for n, c in cals.items()}}


def _job_awq(rank, world, moa, single):
    """awq_lite (default search: Gram scores + re-scored near-ties; and the error-GEMM engine): act_scale is the average
    of the ranks' means, the per-alpha losses are summed, every rank picks the single-rank run's alpha and folds the
    same weights."""
    mq = moa.model_quant
    for search, dtype, preset in (("auto", torch.bfloat16, "INT4_AWQ_CFG"), ("gemm", torch.bfloat16, "INT4_AWQ_CFG"),
                                  ("gram", torch.float32, "INT4_AWQ_CFG"), ("auto", torch.bfloat16, "W4A8_AWQ_BETA_CFG")):
        # W4A8: the per-channel input amax collected in the cache pass is MAX-synchronised before it collapses
        cfg = copy.deepcopy(getattr(mq, preset))
        cfg["algorithm"] = {"method": "awq_lite", "search": search}
        batches = _batches(128, dtype, n=_n_batches(world))
        mine = batches if single else batches[rank::world]
        model = moa.quantize(MLP(dtype=dtype), cfg, lambda m: [m(b) for b in mine])
        hs = {n: m.awq_lite for n, m in model.named_modules() if hasattr(m, "awq_lite")}
        yield {"alpha": {n: h.best_alpha for n, h in hs.items()}, "act_scale": {n: h.act_scale.clone() for n, h in hs.items()},
               "loss": {n: h.loss_buf.clone() for n, h in hs.items()}, "amax": _amaxes(model),
               "w": {n: p.detach().clone() for n, p in model.named_parameters()},
               "contenders": {n: h.contenders for n, h in hs.items()},
               "chan_amax": {n: m.input_quantizer._amax_for_smoothing.clone() for n, m in model.named_modules()
                             if hasattr(m, "awq_lite") and hasattr(m.input_quantizer, "_amax_for_smoothing")},
               "scored_here": {n: (h.use_gram, h.scored_here) for n, h in hs.items()}}


class _DecoderBlock(torch.nn.Module):
    def __init__(self, d, g):
        super().__init__()
        self.qkv = torch.nn.Linear(d, d, bias=False)
        self.up = torch.nn.Linear(d, 2 * d, bias=False)
        self.down = torch.nn.Linear(2 * d, d, bias=False)
        with torch.no_grad():
            for lin in (self.qkv, self.up, self.down):
                lin.weight.copy_(torch.randn(lin.weight.shape, generator=g) * 0.05)

    def forward(self, h):
        m = h + 0.5 * self.qkv(h)
        return m + self.down(torch.nn.functional.gelu(self.up(m)))


class _DecoderStack(torch.nn.Module):
    def __init__(self, d=128, n=3, dtype=torch.bfloat16):
        super().__init__()
        g = torch.Generator().manual_seed(11)
        self.embed = torch.nn.Linear(d, d, bias=False)
        with torch.no_grad():  # (every process must build the SAME replica: nothing from the global generator)
            self.embed.weight.copy_(torch.randn(d, d, generator=g) * 0.1)
        self.layers = torch.nn.ModuleList([_DecoderBlock(d, g) for _ in range(n)])
        self.to(dtype)

    def forward(self, x):
        h = self.embed(x)
        for layer in self.layers:
            h = layer(h)
        return h


def _job_awq_layer_local(rank, world, moa, single):
    """awq_lite walking a decoder stack one layer at a time (one forward per layer, near-ties re-scored from the stored
    activations) under data parallelism: every layer's statistics are reduced before the next layer runs, every rank picks
    the single-rank alphas and folds the same weights."""
    mq = moa.model_quant
    cfg = copy.deepcopy(mq.INT4_AWQ_CFG)
    cfg["quant_cfg"]["*embed*"] = {"enable": False}
    cfg["quant_cfg"]["*weight_quantizer"] = {"num_bits": 4, "block_sizes": {-1: 32, "type": "static"}, "enable": True}
    cfg["algorithm"] = {"method": "awq_lite", "search": "auto", "layer_local": True, "tie_margin": 0.05}
    batches = _batches(128, torch.bfloat16, n=_n_batches(world))
    mine = batches if single else batches[rank::world]
    model = moa.quantize(_DecoderStack(), cfg, lambda m: [m(b) for b in mine])
    from model_optimizer_amd import model_calib

    st = dict(model_calib.AWQ_LITE_STATS)
    assert st.get("layer_local") and st["passes"] == 1 and st["rescored_candidates"] > 0, st
    hs = {n: m.awq_lite for n, m in model.named_modules() if hasattr(m, "awq_lite")}
    yield {"alpha": {n: h.best_alpha for n, h in hs.items()}, "act_scale": {n: h.act_scale.clone() for n, h in hs.items()},
           "loss": {n: h.loss_buf.clone() for n, h in hs.items()}, "amax": _amaxes(model),
           "w": {n: p.detach().clone() for n, p in model.named_parameters()},
           "contenders": {n: h.contenders for n, h in hs.items()}}


def _job_layerwise(rank, world, moa, single):
    """`algorithm.layerwise` under data parallelism: the walk runs on every rank over its own batches, each layer's
    calibration function reduces its statistics across the ranks before the next layer is walked -- amax (max: bit for
    bit) and AWQ alphas / folded weights equal the single-rank layer-by-layer run.  A checkpoint directory is refused in a
    multi-process job, as in the reference (utils/layerwise_calib.py:574-579)."""
    mq = moa.model_quant
    for preset, method in (("FP8_DEFAULT_CFG", "max"), ("INT4_AWQ_CFG", "awq_lite")):
        cfg = copy.deepcopy(getattr(mq, preset))
        cfg["quant_cfg"]["*embed*"] = {"enable": False}
        if method == "awq_lite":
            cfg["quant_cfg"]["*weight_quantizer"] = {"num_bits": 4, "block_sizes": {-1: 32, "type": "static"}, "enable": True}
        cfg["algorithm"] = {"method": method, "layerwise": {"enable": True}}
        batches = _batches(128, torch.bfloat16, n=_n_batches(world))
        mine = batches if single else batches[rank::world]
        model = moa.quantize(_DecoderStack(), cfg, lambda m: [m(b) for b in mine])
        out = {"amax": _amaxes(model)}
        if method == "awq_lite":
            out["alpha"] = {n: m.awq_lite.best_alpha for n, m in model.named_modules() if hasattr(m, "awq_lite")}
            out["w"] = {n: p.detach().clone() for n, p in model.named_parameters()}
        yield out
    if not single:
        import tempfile

        cfg = copy.deepcopy(mq.FP8_DEFAULT_CFG)
        cfg["algorithm"] = {"method": "max", "layerwise": {"enable": True, "checkpoint_dir": tempfile.mkdtemp()}}
        with pytest.raises(RuntimeError, match="multi-process"):
            moa.quantize(_DecoderStack(), cfg, lambda m: [m(b) for b in mine])


def _job_gptq(rank, world, moa, single):
    """GPTQ under data parallelism: every rank accumulates the Hessians of its share of the batches, the distinct Hessians
    are combined sample-weighted on one owner each, the owner updates the linears that read them and broadcasts the
    weights.  The combined Hessian is a differently associated sum than the single-rank running mean: single weights may
    land on the neighbouring level (key "w~": at most 2 % of a tensor), and every rank must hold the same weights."""
    mq = moa.model_quant
    for preset in ("INT4_BLOCKWISE_WEIGHT_ONLY_CFG", "FP8_DEFAULT_CFG"):
        cfg = copy.deepcopy(getattr(mq, preset))
        if "INT4" in preset:
            cfg["quant_cfg"]["*weight_quantizer"] = {"num_bits": 4, "block_sizes": {-1: 32, "type": "static"}, "enable": True}
        cfg["algorithm"] = {"method": "gptq", "perc_damp": 0.01, "block_size": 64}
        batches = _batches(128, torch.float32, n=_n_batches(world))
        mine = batches if single else batches[rank::world]
        model = moa.quantize(MLP(), cfg, lambda m: [m(b) for b in mine])
        from model_optimizer_amd import gptq

        st = gptq.GPTQ_STATS
        assert st["linears"] == 3 and (st["kernel_linears"] == 3 if single else st["kernel_linears"] <= 3), st
        yield {"w~": {n: p.detach().clone() for n, p in model.named_parameters()}, "amax": _amaxes(model)}


class _TPShard(torch.nn.Module):
    """Rank r's shard of the MLP under tensor parallelism: fc1 column parallel (rows r * H/W ..), fc2 row parallel (the
    matching input columns); the partial outputs are not combined -- only the calibration statistics matter here."""

    def __init__(self, full: MLP, rank, world):
        super().__init__()
        h = full.fc1.out_features // world
        self.fc1 = torch.nn.Linear(full.fc1.in_features, h, bias=False)
        self.fc2 = torch.nn.Linear(h, full.fc2.out_features, bias=False)
        with torch.no_grad():
            self.fc1.weight.copy_(full.fc1.weight[rank * h:(rank + 1) * h])
            self.fc2.weight.copy_(full.fc2.weight[:, rank * h:(rank + 1) * h])

    def forward(self, x):
        return self.fc2(torch.relu(self.fc1(x)))


class _TPFull(torch.nn.Module):
    def __init__(self, full: MLP):
        super().__init__()
        self.fc1, self.fc2 = full.fc1, torch.nn.Linear(full.fc2.in_features, full.fc2.out_features, bias=False)
        with torch.no_grad():
            self.fc2.weight.copy_(full.fc2.weight)

    def forward(self, x):
        return self.fc2(torch.relu(self.fc1(x)))


def _job_tensor_parallel(rank, world, moa, single):
    """The tensor-parallel amax rules (distributed.sync_amax_tensor_parallel; model_calib.py:408-485): after calibrating
    the rank's shard and the TP sync, per-tensor amaxes equal the unsharded model's, the row-parallel per-channel weight
    amax equals the unsharded rows', the column-parallel per-channel weight amax stays the shard's own rows."""
    mq = moa.model_quant
    for preset in ("FP8_DEFAULT_CFG", "INT8_DEFAULT_CFG"):
        cfg = copy.deepcopy(getattr(mq, preset))
        cfg["algorithm"] = {"method": "max", "distributed_sync": False}  # every rank sees every batch: no DP here
        batches = _batches(128, torch.float32)
        full = MLP()
        if single:
            model = moa.quantize(_TPFull(full), cfg, lambda m: [m(b) for b in batches])
            yield {"amax": _amaxes(model)}
            continue
        model = moa.quantize(_TPShard(full, rank, world), cfg, lambda m: [m(b) for b in batches])
        picked = moa.distributed.sync_amax_tensor_parallel(model, None, lambda n, m: n == "fc1", lambda n, m: n == "fc2")
        yield {"amax": _amaxes(model), "picked": len(picked), "rank": rank, "world": world}


def _compare_tp(want, got):
    for a, b in zip(want, got):
        r, w = b["rank"], b["world"]
        for name, full in a["amax"].items():
            mine = b["amax"][name]
            if full.numel() == 1:
                assert torch.equal(mine.reshape(()), full.reshape(())), f"tp {name}: per-tensor amax differs"
            elif name.startswith("fc1"):  # column parallel, per output channel: the shard's own rows, untouched
                h = full.shape[0] // w
                assert torch.equal(mine, full[r * h:(r + 1) * h]), f"tp {name}"
            else:  # row parallel, per output channel: MAX over the input shards = the unsharded row amax
                assert torch.equal(mine, full), f"tp {name}"
        assert b["picked"] == (4 if all(v.numel() == 1 for v in a["amax"].values()) else 3)


def _compare(kind, want, got):
    for i, (a, b) in enumerate(zip(want, got)):
        for key in a:
            for name in a[key]:
                x, y = a[key][name], b[key][name]
                if key == "scored_here":
                    continue  # which rank scored a linear is checked across ranks in the worker
                if key == "mask~":
                    # SparseGPT: the combined Hessian is a differently associated fp32 sum than the single-rank running
                    # mean, which can flip near-ties of the pruning scores (and everything downstream in that row)
                    assert x.shape == y.shape and (x == y).float().mean() >= 0.97, f"{kind}[{i}] {name}"
                    assert torch.equal(y.view(-1, 4).sum(1), torch.full((y.numel() // 4,), 2)), f"{kind}[{i}] {name}: not 2:4"
                    continue
                if key == "w~":
                    assert x.shape == y.shape and (x == y).float().mean() >= 0.98, f"{kind}[{i}] {name}: {(x != y).float().mean():.4f} differ"
                    continue
                if key in ("alpha", "contenders"):
                    assert x == y, f"{kind}[{i}] {key} {name}: {x} vs {y}"
                elif key in ("act_scale", "loss") or ((kind.startswith("awq") or (kind == "layerwise" and i == 1)) and key in ("amax", "w")):
                    # averages / sums over ranks associate differently from the single-rank order (fp32)
                    tol = 1e-6 if key == "act_scale" else 2e-2
                    assert torch.allclose(x.float(), y.float(), rtol=tol, atol=0), f"{kind}[{i}] {key} {name}"
                else:
                    assert x.shape == y.shape and torch.equal(x, y), f"{kind}[{i}] {key} {name}: not bit-equal"


def _worker(rank, world, port, kind, ret):
    try:
        moa = _moa_import.load()
        _install_backend(moa)
        job = globals()[f"_job_{kind}"]
        with torch.no_grad():
            want = list(job(rank, world, moa, single=True))  # no process group yet: plain single-rank flow
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            if kind not in ("tensor_parallel", "undeclared"):
                moa.distributed.declare_data_parallel()  # the ranks are replicas: weight-side work may be dealt out
            got = list(job(rank, world, moa, single=False))
        if kind == "tensor_parallel":
            _compare_tp(want, got)
            ret[rank] = "ok"
            return
        _compare(kind, want, got)
        if kind == "awq":
            # Gram-scored linears: every Gram matrix was reduced to ONE rank, which alone evaluated the 11 quadratic forms
            for g in got:
                mine = g["scored_here"]
                everyone = [None] * world
                dist.all_gather_object(everyone, mine)
                for name, (use_gram, _) in mine.items():
                    n_scorers = sum(int(e[name][1]) for e in everyone)
                    assert n_scorers == (1 if use_gram else 0), f"{name}: scored on {n_scorers} ranks"
                if any(u for u, _ in mine.values()):
                    assert any(s for _, s in mine.values()) or any(s for e in everyone for _, s in e.values())
        # and every rank holds the same state (the reference's property)
        for g in got:
            for key in ("amax", "hist", "mask", "mask~", "w", "w~"):
                for name, t in g.get(key, {}).items():
                    ref = t.clone().float()
                    dist.all_reduce(ref, op=dist.ReduceOp.MAX)
                    assert torch.equal(ref, t.float()), f"{kind} {key} {name} differs between ranks"
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        import traceback

        ret[rank] = f"{type(e).__name__}: {e}\n{traceback.format_exc()}"
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["max_and_smoothquant", "histogram", "awq", "awq_layer_local", "tensor_parallel", "weight_side",
                                  "undeclared", "gptq", "layerwise"])
def test_data_parallel_flow_equals_single_rank(kind):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), kind, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}, "\n".join(f"rank {r}: {v}" for r, v in dict(ret).items())


@pytest.mark.parametrize("kind", ["weight_side", "awq", "max_and_smoothquant", "gptq"])
def test_three_replicas_uneven_shards(kind):
    """world = 3 over a model with fewer linears than that: some rank owns no weight of a given pass (empty shard, nothing to
    broadcast), the calibration batches split unevenly -- results still equal the single-rank run."""
    world = 3
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), kind, ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}, "\n".join(f"rank {r}: {v}" for r, v in dict(ret).items())


def test_forced_single_rank_walks_every_collective_site():
    """tests/dist_nccl_world1.py (the GPU suite runs it on RCCL) dry-run on gloo + the host-memory stand-in: with
    MOQ_FORCE_DIST=1 a world of one executes every collective call site and must reproduce the plain run bit for bit."""
    import json
    import subprocess

    env = dict(os.environ, MOQ_TEST_DEVICE="cpu", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.pop("MOQ_FORCE_DIST", None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "dist_nccl_world1.py")], capture_output=True, text=True,
                       timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["ok"], line["mismatches"]
    for name in ("all_reduce", "reduce", "broadcast", "all_gather_object", "broadcast_object_list", "barrier"):
        assert line["calls"].get(name, 0) > 0, line["calls"]


# ------------------------------------------------------------------------------------------------ DP subgroups
def _subgroup_worker(rank, world, port, ret):
    """4 ranks = 2 data-parallel groups {0, 1} and {2, 3} (as the DP groups of a DP x TP layout are): the replicas of a
    group hold the SAME model (seeded by the group), the two groups different ones.  declare_data_parallel(own group):
    every weight-side unit must have exactly one owner INSIDE the group (shard_list numbers the ranks by the group, like
    owner_rank / broadcast_from_owners), statistics stay inside the group, and every rank ends with the single-rank result
    of ITS group's model."""
    try:
        moa = _moa_import.load()
        _install_backend(moa)
        mq, sp = moa.model_quant, moa.sparsity
        gid = rank // 2
        batches = _batches(128, torch.float32, n=4)

        def flows(mine):
            out = []
            m = moa.quantize(MLP(seed=10 + gid), copy.deepcopy(mq.FP8_DEFAULT_CFG), lambda mm: [mm(b) for b in mine])
            out.append({"amax": _amaxes(m)})
            mq.fold_weight(m)
            out.append({"w": {n: p.detach().clone() for n, p in m.named_parameters()}})
            m = sp.sparsify(MLP(seed=10 + gid), "sparse_magnitude")
            out.append({"mask": {n: mod._weight_mask.clone() for n, mod in m.named_modules() if hasattr(mod, "_weight_mask")},
                        "w": {n: p.detach().clone() for n, p in m.named_parameters()}})
            m = moa.quantize(MLP(seed=10 + gid), copy.deepcopy(mq.INT4_BLOCKWISE_WEIGHT_ONLY_CFG), None)
            mq.fold_weight(m)
            out.append({"w": {n: p.detach().clone() for n, p in m.named_parameters()}})
            return out

        with torch.no_grad():
            want = flows(batches)
            os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
            dist.init_process_group("gloo", rank=rank, world_size=world)
            groups = [dist.new_group([0, 1]), dist.new_group([2, 3])]  # (every rank creates every group)
            moa.distributed.declare_data_parallel(groups[gid])
            assert moa.distributed.shard_list(list(range(5))) == list(range(5))[rank % 2::2]
            got = flows(batches[rank % 2::2])
        _compare("subgroup", want, got)
        ret[rank] = "ok"
    except Exception as e:  # noqa: BLE001
        import traceback

        ret[rank] = f"{type(e).__name__}: {e}\n{traceback.format_exc()}"
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_two_data_parallel_subgroups_of_a_four_rank_world():
    world = 4
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_subgroup_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {r: "ok" for r in range(world)}, "\n".join(f"rank {r}: {v}" for r, v in dict(ret).items())
