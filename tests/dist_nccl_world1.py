"""Run by tests/test_gpu_dist_nccl.py in a subprocess: one rank, backend "nccl" (RCCL), MOQ_FORCE_DIST=1 -- every
collective call site of the data-parallel flows executes on device tensors through RCCL (all_reduce MAX / SUM buckets,
chunked reduce of the Gram matrices and Hessians, broadcast_from_owners, all_gather / all_gather_object,
broadcast_object_list, barrier), and with a world of one the results must equal the plain single-process run bit for bit.
Prints one JSON line."""

import copy
import json
import os
import sys
import tempfile

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import _moa_import  # noqa: E402

from test_distributed_flows_cpu import MLP, _amaxes, _batches  # noqa: E402

DEV = os.environ.get("MOQ_TEST_DEVICE", "cuda:0")  # "cpu": dry run of this script on gloo + the host-memory stand-in


def flows(moa):
    from model_optimizer_amd import model_calib

    mq, sp, ex = moa.model_quant, moa.sparsity, moa.export
    out = {}
    dt = torch.bfloat16
    batches = [b.to(DEV) for b in _batches(128, dt)]
    loop = lambda m: [m(b) for b in batches]  # noqa: E731
    for preset in ("FP8_DEFAULT_CFG", "INT8_SMOOTHQUANT_CFG", "MXFP4_SMOOTHQUANT_CFG"):
        m = moa.quantize(MLP(dtype=dt).to(DEV), copy.deepcopy(getattr(mq, preset)), loop)
        out[f"{preset}.amax"] = _amaxes(m)
        out[f"{preset}.w"] = {n: p.detach().clone() for n, p in m.named_parameters()}
        with tempfile.TemporaryDirectory() as d:
            box = [d]
            if dist.is_initialized():
                dist.broadcast_object_list(box, src=0)
            state = ex.export_state_dict(m, dt)
            ex.save_checkpoint(state, box[0], ex.hf_quant_config(m))
            out[f"{preset}.files"] = {f: torch.tensor(os.path.getsize(os.path.join(box[0], f)))
                                      for f in sorted(os.listdir(box[0])) if f.endswith(".safetensors")}
        out[f"{preset}.ckpt"] = {k: v.detach().clone() for k, v in state.items()}
        mq.fold_weight(m)
        out[f"{preset}.folded"] = {n: p.detach().clone() for n, p in m.named_parameters()}
    # histogram calibrators: range broadcast, int64 SUM bucket
    for method, kw in (("percentile", {"percentile": 99.9}), ("entropy", {})):
        cfg = copy.deepcopy(mq.INT8_DEFAULT_CFG)
        cfg["quant_cfg"]["*input_quantizer"] = {"num_bits": 8, "axis": None, "calibrator": "histogram"}
        cfg["algorithm"] = None
        m = moa.quantize(MLP(dtype=dt).to(DEV), cfg, None)
        model_calib.histogram_calibrate(m, loop, method=method, **kw)
        out[f"hist.{method}"] = _amaxes(m)
    # AWQ-lite: act-scale / loss / contender buckets, Gram matrices reduced to their owner in chunks
    for search in ("auto", "gemm", "gram"):
        cfg = copy.deepcopy(mq.INT4_AWQ_CFG)
        cfg["algorithm"] = {"method": "awq_lite", "search": search, "tie_margin": 0.02 if search == "auto" else None}
        m = moa.quantize(MLP(dtype=dt).to(DEV), cfg, loop)
        hs = {n: mod.awq_lite for n, mod in m.named_modules() if hasattr(mod, "awq_lite")}
        out[f"awq.{search}.loss"] = {n: h.loss_buf.clone() for n, h in hs.items()}
        out[f"awq.{search}.alpha"] = {n: torch.tensor(h.best_alpha) for n, h in hs.items()}
        out[f"awq.{search}.w"] = {n: p.detach().clone() for n, p in m.named_parameters()}
    # 2:4 masks dealt over the ranks and broadcast; SparseGPT Hessians combined on their owner
    m = sp.sparsify(MLP(dtype=dt).to(DEV), "sparse_magnitude")
    out["mask.magnitude"] = {n: mod._weight_mask.clone() for n, mod in m.named_modules() if hasattr(mod, "_weight_mask")}
    m = sp.sparsify(MLP(dtype=dt).to(DEV), "sparsegpt", forward_loop=loop)
    out["mask.sparsegpt"] = {n: mod._weight_mask.clone() for n, mod in m.named_modules() if hasattr(mod, "_weight_mask")}
    # GPTQ: Hessians combined on their owner, updated weights broadcast
    cfg = copy.deepcopy(mq.INT4_BLOCKWISE_WEIGHT_ONLY_CFG)
    cfg["quant_cfg"]["*weight_quantizer"] = {"num_bits": 4, "block_sizes": {-1: 32, "type": "static"}, "enable": True}
    cfg["algorithm"] = {"method": "gptq", "block_size": 64}
    m = moa.quantize(MLP(dtype=dt).to(DEV), cfg, loop)
    out["gptq.w"] = {n: p.detach().clone() for n, p in m.named_parameters()}
    return out


def main():
    moa = _moa_import.load()
    if DEV == "cpu":
        from test_distributed_flows_cpu import _install_backend

        _install_backend(moa)
    else:
        assert torch.cuda.is_available()
        torch.cuda.set_device(0)
    with torch.no_grad():
        want = flows(moa)
        os.environ["MOQ_FORCE_DIST"] = "1"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if DEV == "cpu":
            dist.init_process_group("gloo", rank=0, world_size=1)
        else:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
        moa.distributed.declare_data_parallel()
        moa.distributed.MAX_COLLECTIVE_BYTES = 16 << 10  # several calls per Gram matrix / weight even at test sizes
        calls = {}
        for name in ("all_reduce", "reduce", "broadcast", "all_gather_object", "broadcast_object_list", "barrier"):
            orig = getattr(dist, name)

            def counted(*a, _orig=orig, _name=name, **k):
                calls[_name] = calls.get(_name, 0) + 1
                return _orig(*a, **k)

            setattr(dist, name, counted)
        got = flows(moa)
        if DEV != "cpu":
            torch.cuda.synchronize()
    bad = []
    for key in want:
        if key.endswith(".files"):
            continue  # file names differ by design (one file vs rank shards)
        for name in want[key]:
            a, b = want[key][name], got[key].get(name)
            if b is None or a.shape != b.shape or not torch.equal(a.cpu(), b.cpu()):
                bad.append(f"{key}/{name}")
    extra_files = sorted(got["FP8_DEFAULT_CFG.files"])
    dist.destroy_process_group()
    print(json.dumps({"ok": not bad, "mismatches": bad[:20], "compared": sum(len(v) for v in want.values()),
                      "calls": calls, "sharded_files": extra_files}))


if __name__ == "__main__":
    main()
