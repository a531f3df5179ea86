"""bench.py -- GB/s of weights calibrated + quantize-dequantized on MI355X (BASELINE.json metric).

One step = one pass of the hot path over ALL linear weights of a synthetic model, already resident in HBM:
    fp8      (default): per-tensor abs-max of every weight (one multi-tensor launch),
             [N>1: ONE bucketed RCCL all-reduce(MAX) of all amax values -- the reference's
             sync_amax_across_distributed_group, model_calib.py:390-407], per-tensor FP8-E4M3 QDQ of every
             weight (one multi-tensor launch).      algorithmic HBM bytes: 2 + 4 = 6 B/element
    int4g128 (BASELINE configs[2] weight side / north-star kernel): fused per-group(128) abs-max + INT4 QDQ,
             one launch, 4 + 4/128 B/element.
    mxfp4, mask24, int8 : the other formats of the path, for the record.
    fp8-mask24 (BASELINE configs[3] as ONE step): 2:4 magnitude masks + the masked weights in place + their per-tensor abs-max
             from one pass (5 B/element), [the amax bucket], FP8 QDQ of the sparse weights (4 B/element).
    mxfp4-sq (BASELINE configs[4]): SmoothQuant fold W <- dtype(W * (1/s)[col]) of every weight (model_calib.
             apply_pre_quant_scale_and_smooth) composed with the MXFP4 g = 32 quantize-dequantize, the whole model in ONE
             launch (multi_tensor.fold_mx_fused = moq_mt_fold_mx_fused: what smoothquant(fold_weights=True) runs): one read
             + one write, 4 B/element (+ the column scales, 4 B per column, L2-resident).  Rounds 2-5 ran it as 560
             scale_cols launches + the MX launch: 8 B/element, 96.0 ms/step on Llama-3-70B.
`value` = weight bytes of the WHOLE pool (2 B/element) / wall time per step.

Which model (when --model is not given):
    N = 1 : Llama-3-8B (224 tensors, 13.96 GB) -- BASELINE configs[1], the configuration the metric is quoted on.
    N > 1 : STRONG scaling of the multi-GPU configuration BASELINE.json names for the format -- ONE model's tensors dealt
            over the ranks (largest first, round-robin: every rank gets the same number of tensors of every shape), no
            data-path collective, one amax bucket all-reduce(MAX) in flight under the QDQ launch:
              fp8 / int8 / mask24        -> Mixtral-8x7B (configs[3]; 896 tensors, 92.9 GB: 11.6 GB per rank at N = 8)
              int4g128 / mxfp4 / mxfp4-sq -> Llama-3-70B (configs[4]; 560 tensors, 136.9 GB: 17.1 GB per rank at N = 8)
            Both fit ONE MI355X in place, so `--model mixtral-8x7b` / `--model llama3-70b` at N = 1 is the base of that
            curve; the default N = 1 line carries it as extra.scale_base_n1 (same step, same model, one GPU).
            `--scaling weak` (every rank holds one model's worth; the pool grows with N) is kept as a switch and the
            weak leg of the default run is reported in extra.weak_scaling -- it is not a BASELINE configuration.

`extra` (rank 0, outside the timed region) carries the other half of BASELINE.json's metric and the north-star
target: the INT4-AWQ PTQ wall-clock of the full synthetic Llama-3-8B (tools/awq_bench.py, calibration batches sharded
over the ranks) and, at N = 1, the fused per-group amax + INT4 QDQ kernel over all Llama-3-70B weights in place.
The CPU baseline runs LAST and in a process of its own (`--cpu-baseline-only`): its OpenMP runtime never shares a
process with a timed flow.

Contract: python bench.py --gpus N --steps K --warmup W ; one JSON line on rank 0.  `--gpus N` without a launcher
(no WORLD_SIZE in the environment) re-executes itself under `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1`; under torchrun (the driver's form) it runs as the rank it is.

RULE: no edit of this file ships without ONE full default run (`python bench.py --gpus 1 --steps 20 --warmup 5`, all
extras) after it, its line committed under profiles/ -- round 3 shipped an order change that was only ever run with
shortened extras and the driver's AWQ figure came out 2.4x the claimed one.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import _moa_import  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s achievable)

MODELS = {
    # (hidden, intermediate, layers, kv_dim[, experts])
    "llama3-8b": (4096, 14336, 32, 1024),
    "llama3-70b": (8192, 28672, 80, 1024),
    "mixtral-8x7b": (4096, 14336, 32, 1024, 8),  # BASELINE configs[3]: 8 experts per layer (w1, w3, w2 each)
}
MODEL_NAMES = {"llama3-8b": "Llama-3-8B", "llama3-70b": "Llama-3-70B", "mixtral-8x7b": "Mixtral-8x7B"}


def layer_shapes(model):
    h, i, _, kv = MODELS[model][:4]
    experts = MODELS[model][4] if len(MODELS[model]) > 4 else 1
    return [(h, h), (kv, h), (kv, h), (h, h)] + [(i, h), (i, h), (h, i)] * experts  # q k v o, (gate up down) x experts


def deal(shapes, rank, world):
    """Strong-scaling deal of a model's tensors over the ranks: pool indices sorted largest tensor first (ties by index),
    dealt round-robin -- every rank gets the same number of tensors of every shape whenever the counts divide (they do
    for the three models at 2 / 4 / 8 ranks), so the ranks' bytes are equal; a plain index round-robin leaves Mixtral's
    28-tensor layers 1.7 % out of balance at 4 and 8 ranks.  Returns this rank's pool indices, ascending."""
    order = sorted(range(len(shapes)), key=lambda i: (-shapes[i][0] * shapes[i][1], i))
    return sorted(order[rank::world])


def make_weights(model, n_layers, device, seed=1234, rank=0, world=1, scaling="strong"):
    """bf16 N(0, 0.02^2) with 0.1% x8 outliers, generated on the GPU; tensor i of the pool is seeded by its
    index.  strong: the pool is the model's list, dealt by `deal` (size-balanced round-robin of the per-layer tensors).
    weak: the pool is `world` times the model's list and rank r holds entries [r * len, (r + 1) * len).
    Returns (tensors of this rank, their indices in the pool, number of tensors of the whole pool)."""
    shapes = [shape for _ in range(n_layers) for shape in layer_shapes(model)]
    if scaling == "weak" and world > 1:
        first, n_model = rank * len(shapes), len(shapes)
        shapes = shapes * world
        mine = range(first, first + n_model)
    else:
        mine = deal(shapes, rank, world)
    g = torch.Generator(device=device)
    ws, idx = [], []
    for i in mine:
        shape = shapes[i]
        g.manual_seed(seed + i)
        w = torch.randn(shape, generator=g, device=device, dtype=torch.float32) * 0.02
        m = torch.rand(shape, generator=g, device=device) < 0.001
        ws.append(torch.where(m, w * 8, w).to(torch.bfloat16))
        idx.append(i)
        del w, m
    return ws, idx, len(shapes)


PMC_KERNEL = {"fp8": "mt_map_kernel<2, moq::OpFp8Qdq>", "int8": "mt_map_kernel<2, moq::OpIntQdq>",
              "int4g128": "mt_group_kernel<2, 16>", "mask24": "mt_mask24_kernel<2>", "mxfp4": "mt_mx_kernel<2, 4, 6>",
              "mxfp4-sq": "mt_fold_mx_kernel<2, 4, 6>"}


def pmc_traffic(workload, model, n_layers):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC profile of this same
    command (tools/profile_bench.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled per
    MI355X_MICROARCH.md's gfx950 note, WRITE_SIZE x 1024 B verified against a kernel with known write bytes).
    Counters cannot be read from inside the timed process, so the bench line quotes the profile; null when no
    profile of this workload / model size is committed."""
    if n_layers != MODELS[model][2] or workload not in PMC_KERNEL:
        return None, None
    path = None
    # newest committed profile of this workload ("r05h": round 5's; the files tagged plain "r05" were written late in round 4);
    # profiles of a model other than the default carry its name (round 6: configs[4] on Llama-3-70B)
    suffix = "" if model == "llama3-8b" else f"_{model}"
    for tag in ("r06", "r05h", "r05", "r04", "r03", "r02", "r01c", "r01b", "r01"):
        cand = os.path.join(ROOT, "profiles", f"{tag}_{workload}{suffix}_pmc.json")
        if os.path.exists(cand):
            path = cand
            break
    if path is None:
        return None, None
    try:
        with open(path) as f:
            prof = json.load(f)
        for name, v in prof.items():
            if PMC_KERNEL[workload] in name and v.get("hbm_read_bytes") is not None:
                return int(v["hbm_read_bytes"] + (v.get("hbm_write_bytes_raw") or 0)), os.path.relpath(path, ROOT)
    except (OSError, ValueError):
        pass
    return None, None


def node_probe(dev, budget_ms=25.0):
    """What THIS box gives the library's own streams on a fresh 2 GiB arena (one tensor, carved like a weight): the read-only
    abs-max and the in-place FP8 quantize-dequantize (read + write, the dominant kernel's shape), each repeated for
    ~budget_ms, HIP-event timed.  Reported beside the roofline so that a line from a slow box says so itself (rounds 1-3 used
    a torch copy_ here, which read 5.0 TB/s on every box whatever the kernel did -- VERDICT r03 weak #2); not part of any
    timed region."""
    from model_optimizer_amd.multi_tensor import SegmentTable

    n = 1 << 30  # bf16 elements = 2 GiB
    src = torch.empty(n, dtype=torch.bfloat16, device=dev).normal_(0, 0.02)
    tab = SegmentTable([src], outputs=[src])
    tab.calibrate_amax()
    out = {}
    for name, fn, nbytes in (("node_read_GBs", lambda: tab.calibrate_amax(), 2.0 * n),
                             ("node_copy_GBs", lambda: tab.fake_quant_e4m3(), 4.0 * n)):
        fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 0
        t0 = time.perf_counter()
        a.record()
        while True:
            fn()
            reps += 1
            if reps % 8 == 0:
                torch.cuda.synchronize()
                if (time.perf_counter() - t0) * 1e3 > budget_ms:
                    break
        b.record()
        torch.cuda.synchronize()
        out[name] = round(nbytes * reps / (a.elapsed_time(b) * 1e-3) / 1e9, 1)
    out["node_probe"] = "library kernels on a fresh 2 GiB tensor: abs-max (read) / in-place FP8 QDQ (read + write)"
    del tab, src
    return out


def cpu_baseline(workload, budget_s=12.0):
    """Time the CPU oracle (a C restatement of the reference's eager path, OpenMP on every host CPU this process
    may use) on a bounded sample: repeated 4096x4096 bf16 weights until ~budget_s of CPU work."""
    from oracle import oracle

    threads = oracle.usable_cpus()  # visible CPUs capped by affinity and the cgroup quota (16 of 256 on the GPU boxes)
    oracle.set_threads(threads)
    w = (torch.randn(4096, 4096, generator=torch.Generator().manual_seed(1234)) * 0.02).to(torch.bfloat16)
    n_bytes = w.numel() * 2
    sq_scale = torch.exp(0.5 * torch.randn(4096, generator=torch.Generator().manual_seed(99)))

    def one():
        if workload == "fp8":
            a = oracle.reduce_amax(w)
            oracle.fake_quant_e4m3(w, a.reshape(1))
        elif workload == "int4g128":
            oracle.amax_qdq_int_group(w, 128, num_bits=4, narrow_range=False)
        elif workload == "int8":
            a = oracle.reduce_amax(w)
            oracle.fake_quant_int(w, a.reshape(1), 8, False, True)
        elif workload == "mxfp4":
            oracle.mx_fused_amax_convert(w, 32, "E2M1")
        elif workload == "mxfp4-sq":
            oracle.mx_fused_amax_convert(oracle.scale_cols(w, sq_scale), 32, "E2M1")
        else:
            oracle.mask_2to4(w)

    one()  # warm (builds the .so, faults pages)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < budget_s:
        one()
        reps += 1
    dt = time.perf_counter() - t0
    out = {"value": round(reps * n_bytes / dt / 1e9, 4), "unit": "GB/s", "cores": threads, "kind": "port",
           "sample": f"{reps} x (4096x4096 bf16 weight, {workload} calibrate+QDQ), C oracle with OpenMP, "
                     f"{dt:.1f} s"}
    # The reference's OWN eager CPU path on the same sample, on THIS host ("re-measure on the GPU
    # box's host"): its Python package is importable from /root/reference (build container) or from the gitignored
    # archive tools/stage_reference.sh packs under oracle/_ref/ (the GPU box).  When it is, it IS the baseline
    # (kind "reference") and the C port rides beside it; otherwise the port stays the reported number.
    ref = reference_cpu_baseline(workload, w, threads, budget_s=min(budget_s, 10.0))
    if ref is not None and ref.get("value"):
        out = dict(ref, port={k: out[k] for k in ("value", "unit", "cores", "sample")})
    elif ref is not None:
        out["reference_eager"] = ref
    return out


def reference_cpu_baseline(workload, w, threads, budget_s=10.0):
    """SURVEY.md 8(d) "CPU baseline": max_calibrate(TensorQuantizer, lambda q: q(w)) + one QDQ forward q(w) per weight,
    bf16 input, torch.set_num_threads(<usable host CPUs>) -- the reference's own code (model_calib.py:310-498,
    nn/modules/tensor_quantizer.py:1119-1221, tensor_quant.py:46-59, :607-645), repeated for ~budget_s.  None when no
    reference is present or the workload has no CPU implementation there (the MX formats are CUDA-only)."""
    cfgs = {"fp8": dict(num_bits=(4, 3), axis=None), "int8": dict(num_bits=8, axis=None),
            "int4g128": dict(num_bits=4, block_sizes={-1: 128, "type": "static"})}
    if workload not in cfgs and workload != "mask24":
        return None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import ref_shim

        source = ref_shim.reference_source()
        if source is None:
            return None
        ref_shim.install()
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            from modelopt.torch.quantization.config import QuantizerAttributeConfig
            from modelopt.torch.quantization.model_calib import max_calibrate
            from modelopt.torch.quantization.nn import TensorQuantizer
            from modelopt.torch.sparsity.weight_sparsity.magnitude import create_asp_mask
        torch.set_num_threads(threads)

        def one():
            with torch.no_grad():
                if workload == "mask24":
                    create_asp_mask(w, "2:4 sparsity")
                    return
                q = TensorQuantizer(QuantizerAttributeConfig(**cfgs[workload]))
                max_calibrate(q, lambda qq: qq(w), distributed_sync=False)
                q(w)

        one()
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < budget_s:
            one()
            reps += 1
        dt = time.perf_counter() - t0
        return {"value": round(reps * w.numel() * 2 / dt / 1e9, 4), "unit": "GB/s", "cores": threads, "kind": "reference",
                "sample": f"{reps} x (4096x4096 bf16 weight, {workload}: the reference's max_calibrate(TensorQuantizer) + one "
                          f"QDQ forward, eager CPU path, torch threads = {threads}), {dt:.1f} s on this host",
                "reference_source": source, "torch": torch.__version__}
    except Exception as e:  # a reported baseline never costs the line
        return {"value": None, "kind": "reference", "sample": f"failed: {type(e).__name__}: {e}"}


# the multi-GPU configuration BASELINE.json names for each format (configs[3]: Mixtral-8x7B FP8 + 2:4; configs[4]:
# Llama-3-70B MXFP4 g32 + SmoothQuant; the per-group INT4 pass is the north-star kernel "over Llama-3-70B weight tensors")
SCALE_MODEL = {"fp8": "mixtral-8x7b", "int8": "mixtral-8x7b", "mask24": "mixtral-8x7b", "fp8-mask24": "mixtral-8x7b",
               "int4g128": "llama3-70b", "mxfp4": "llama3-70b", "mxfp4-sq": "llama3-70b"}
ALG_BYTES_PER_ELEM = {"fp8": 4.0, "int8": 4.0, "int4g128": 4.0 + 4.0 / 128, "mxfp4": 4.0, "mxfp4-sq": 4.0, "mask24": 3.0,
                      "fp8-mask24": 5.0}
DOM_KERNEL = {"fp8": "mt_map_kernel<bf16, OpFp8Qdq>", "int8": "mt_map_kernel<bf16, OpIntQdq>",
              "int4g128": "mt_group_kernel<bf16, 16>", "mxfp4": "mt_mx_kernel<bf16, 4, E2M1>",
              "mxfp4-sq": "mt_fold_mx_kernel<bf16, 4, E2M1>", "mask24": "mt_mask24_kernel<bf16>",
              "fp8-mask24": "mt_mask24_apply_kernel<bf16>"}


def awq_cold_subprocess(args, dev):
    """tools/awq_bench.py --cold in a process of its own (see the call site); returns its `cold` object plus the wall-clock of
    the whole child as this process saw it and how long the driver needed to have clean pages again."""
    import subprocess

    import gc

    torch.cuda.synchronize()
    gc.collect()  # (the flows' modules and closures refer to each other: without this their tensors pin the 160 GiB segment the
    reserved_before = torch.cuda.memory_reserved(dev)
    torch.cuda.empty_cache()  # warm run parked in torch's cache, and the child finds 120 GB free instead of 290)
    handed_back = max(0, reserved_before - torch.cuda.memory_reserved(dev))
    t_w, probes = time.perf_counter(), 0
    # The driver wipes what was just handed back in the background (~25 GB/s, tools/alloc_wipe_probe.py) and an allocation that
    # meets unwiped VRAM stalls for seconds.  Rounds 5-6 probed with 32 GiB allocations until one was prompt -- but the child
    # takes ~80 GB, a prompt 32 GiB only says 32 GiB of clean pages exist, and the probe's own free is wiped again: one lease
    # in five still put a 4 s stall into the child's first batch (profiles/r06g_*: linear 91, 4.15 s).  So: wait out the wipe of
    # what this process handed back (at 20 GB/s, bounded), then confirm with ONE small allocation.
    time.sleep(min(30.0, handed_back / 20e9 + 1.0))
    while time.perf_counter() - t_w < 45.0:
        t = time.perf_counter()
        block = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()
        took = time.perf_counter() - t
        del block
        torch.cuda.empty_cache()
        probes += 1
        if took < 0.05:
            break
        time.sleep(1.0)
    wipe_wait = round(time.perf_counter() - t_w, 3)
    free_b, total_b = torch.cuda.mem_get_info(dev)
    held = {"parent_allocated_GB": round(torch.cuda.memory_allocated(dev) / 1e9, 2),
            "parent_reserved_GB": round(torch.cuda.memory_reserved(dev) / 1e9, 2), "device_free_GB": round(free_b / 1e9, 1),
            "device_total_GB": round(total_b / 1e9, 1)}  # (what the child finds: this process keeps its context and whatever it still holds)
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK")}
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "awq_bench.py"), "--layers", str(args.awq_layers),
                        "--batches", str(args.awq_batches), "--cold"], capture_output=True, text=True, timeout=900, env=env)
    wall = round(time.perf_counter() - t0, 3)
    if r.returncode != 0:
        return {"failed": f"rc {r.returncode}: {r.stderr[-300:]}", "child_wall_s": wall}
    line = json.loads(r.stdout.strip().splitlines()[-1])
    return dict(line["cold"], child_wall_s=wall, wipe_wait_s=wipe_wait, wipe_probes=probes, handed_back_GB=round(handed_back / 1e9, 1), **held, passes=line.get("passes"),
                store_dropped=line.get("store_dropped"), stored_input_bytes=line.get("stored_input_bytes"),
                stages_s=line.get("stages_s"), forward_loop_calls=line.get("forward_loop_calls"),
                what="a fresh process: import, build the stack and 64 batches, ONE quantize(); no warm forward, no rehearsal, "
                     "no memory taken ahead of the clock")


def default_model(workload, world):
    return "llama3-8b" if world == 1 else SCALE_MODEL[workload]


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_command(n, argv, port=None):
    """The command `python bench.py --gpus N ...` turns itself into when no launcher started it."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), os.path.abspath(__file__), *argv]


def build_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="fp8", choices=["fp8", "int4g128", "int8", "mxfp4", "mxfp4-sq", "mask24", "fp8-mask24"])
    ap.add_argument("--model", default=None, choices=list(MODELS),
                    help="default: llama3-8b at N = 1 (BASELINE configs[1]); at N > 1 the multi-GPU configuration of the "
                         "format (mixtral-8x7b for fp8 / int8 / mask24, llama3-70b for int4g128 / mxfp4 / mxfp4-sq)")
    ap.add_argument("--layers", type=int, default=0, help="0 = all layers of the model")
    ap.add_argument("--scaling", default="strong", choices=["weak", "strong"],
                    help="N > 1: strong (default) = ONE model's tensors dealt over the ranks; weak = every rank holds one "
                         "model's worth of tensors (the pool grows with N).  The other leg is reported in extra")
    ap.add_argument("--group-mb", type=int, default=0,
                    help="fp8/int8: calibrate+QDQ in groups of <= this many MB of weights (second read from the "
                         "Infinity Cache) instead of two whole-model passes; 0 = off")
    ap.add_argument("--out-of-place", action="store_true",
                    help="QDQ outputs go to separate tensors (rounds 1-2 measured this).  Default since round 3: IN PLACE, "
                         "what the product's fold_weight / max_calibrate paths do (every SegmentTable of the package is built "
                         "with outputs = inputs; y == x is part of the C-ABI contract) -- same 4 B/element of traffic, and "
                         "the only form in which llama3-70b fits one GPU.  The default line carries the out-of-place time "
                         "of the same launch in extra.qdq_out_of_place")
    ap.add_argument("--inplace", action="store_true", help="(accepted for old command lines; in place is the default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-node-probe", action="store_true",
                    help="skip the 2 GiB probe launches after the timed region (tools/profile_bench.sh: the profiled process "
                         "then launches the dominant kernel on the workload's tensors only, so rocprofv3's per-kernel "
                         "average IS the launch the line times)")
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="time the CPU baseline of --workload, print its JSON object and exit (no GPU is touched): the "
                         "process the main run starts for it, last")
    ap.add_argument("--cpu-baseline-inproc-first", action="store_true",
                    help="DIAGNOSTIC (round 3's order): time the CPU baseline inside this process BEFORE the AWQ extras")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary measurements (other kernels, the "
                                                            "Llama-3-70B in-place pass, the INT4-AWQ wall-clock)")
    ap.add_argument("--no-hf", action="store_true", help="extra: skip the INT4-AWQ flow on the random-init HF Llama-3-8B")
    ap.add_argument("--awq-layers", type=int, default=32, help="extra: layers of the INT4-AWQ wall-clock run (0 = skip)")
    ap.add_argument("--awq-batches", type=int, default=64, help="extra: calibration batches of 4096 tokens in total")
    ap.add_argument("--awq-watchdog-s", type=float, default=240.0,
                    help="N > 1: print the line without the AWQ extra if that flow has not finished after this many seconds")
    return ap


class Pool:
    """The tensors one rank holds of a pool of per-layer weights, their segment tables, and one step of the workload."""

    def __init__(self, moa, wl, model, n_layers, dev, rank, world, scaling, inplace=True, group_mb=0, use_dist=False):
        from model_optimizer_amd.multi_tensor import SegmentTable

        self.moa, self.wl, self.dev, self.use_dist = moa, wl, dev, use_dist
        self.weights, owned, self.n_tensors = make_weights(model, n_layers, dev, rank=rank, world=world, scaling=scaling)
        self.n_local = sum(w.numel() for w in self.weights)
        self.n_model_elem = sum(r * c for _ in range(n_layers) for r, c in layer_shapes(model))  # one model
        self.weak = scaling == "weak" and world > 1
        self.n_elem = self.n_model_elem * (world if self.weak else 1)  # the whole pool
        self.owned_idx = torch.tensor(owned, dtype=torch.int64, device=dev)
        self.amax_all = torch.zeros(self.n_tensors, dtype=torch.float32, device=dev)  # every rank ends with every amax
        self.tab = SegmentTable(self.weights, outputs=self.weights if inplace else None,
                                group_size=128 if wl == "int4g128" else None)
        self.groups = None
        if group_mb and wl in ("fp8", "int8"):
            groups, cur, cur_b = [], [], 0
            for i, w in enumerate(self.weights):
                b = w.numel() * w.element_size()
                if cur and cur_b + b > group_mb * 1e6:
                    groups.append(cur)
                    cur, cur_b = [], 0
                cur.append(i)
                cur_b += b
            groups.append(cur)
            self.groups = [SegmentTable([self.weights[i] for i in g], outputs=[self.tab.outputs[i] for i in g])
                           for g in groups]
        self.fold_scales = self.fold_sides = None
        if wl == "mxfp4-sq":
            # per-channel SmoothQuant scales of each tensor (synthetic, log-normal); steps alternate s and 1/s so that the
            # in-place fold keeps the weights' magnitude over many steps
            gs = torch.Generator(device=dev).manual_seed(99)
            self.fold_scales = []
            for w in self.weights:
                sv = torch.exp(0.5 * torch.randn(w.shape[1], generator=gs, device=dev, dtype=torch.float32))
                self.fold_scales.append((sv, 1.0 / sv))
            # the side tables of the two alternating launches, built once (scale pointers + row lengths beside the segments)
            self.fold_sides = [self.tab.fold_side([sv[k] for sv in self.fold_scales], 32) for k in (0, 1)]
        self.step_no = 0
        self.masks = self.mask_tab = None
        if wl in ("mask24", "fp8-mask24"):
            self.masks = [torch.empty(w.shape, dtype=torch.bool, device=dev) for w in self.weights]
            self.mask_tab = SegmentTable(self.weights, outputs=self.masks)
        self.dom_events = []

    def release(self):
        self.tab = self.groups = self.weights = self.masks = self.mask_tab = self.fold_scales = self.fold_sides = None
        torch.cuda.empty_cache()

    def step(self, record, collective=True):
        import torch.distributed as dist

        wl, tab = self.wl, self.tab
        use_dist = self.use_dist and collective
        ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
        e0 = e1 = None

        def start():
            nonlocal e0, e1
            if record:
                e0, e1 = ev(), ev()
                e0.record()

        def stop():
            if record:
                e1.record()
                self.dom_events.append((e0, e1))

        if self.groups is not None:
            start()
            for gt in self.groups:
                gt.calibrate_amax()
                gt.fake_quant_e4m3() if wl == "fp8" else gt.fake_quant_int(8, False, True)
            stop()
            if use_dist:
                self.amax_all.zero_()
                self.amax_all[self.owned_idx] = torch.cat([gt.amax_flat for gt in self.groups])
                dist.all_reduce(self.amax_all, op=dist.ReduceOp.MAX)
        elif wl == "fp8" or wl == "int8":
            tab.calibrate_amax()
            work = None
            if use_dist:
                # one bucket: the owners' values reach every rank (zeros are the identity of abs-max).  The QDQ of a
                # rank's own tensors needs only its own statistics, so the exchange runs on RCCL's stream UNDER the QDQ
                # launch and is waited for at the end of the step (it is part of the step, not of the QDQ's inputs)
                self.amax_all.zero_()
                self.amax_all[self.owned_idx] = tab.amax_flat
                work = dist.all_reduce(self.amax_all, op=dist.ReduceOp.MAX, async_op=True)
            start()
            tab.fake_quant_e4m3() if wl == "fp8" else tab.fake_quant_int(8, False, True)
            stop()
            if work is not None:
                work.wait()  # the step's stream continues only after every rank's statistics have arrived
        elif wl == "fp8-mask24":
            # BASELINE configs[3] as ONE step: 2:4 magnitude masks, the masked weights in place and their per-tensor abs-max
            # from one pass (5 B/element), [the amax bucket], FP8 QDQ of the sparse weights (4 B/element)
            start()
            self.mask_tab.mask_2to4_apply(calibrate=True)
            stop()
            work = None
            if use_dist:
                self.amax_all.zero_()
                self.amax_all[self.owned_idx] = self.mask_tab.amax_flat
                work = dist.all_reduce(self.amax_all, op=dist.ReduceOp.MAX, async_op=True)
            tab.amax_flat.copy_(self.mask_tab.amax_flat)
            tab.fake_quant_e4m3()
            if work is not None:
                work.wait()
        elif wl == "int4g128":
            start()
            tab.amax_qdq_int_group(4, False, False)
            stop()
        elif wl == "mxfp4-sq":
            k = self.step_no & 1
            self.step_no += 1
            start()
            tab.fold_mx_fused(block_size=32, fmt="E2M1", side=self.fold_sides[k])  # fold + MXFP4 QDQ of every weight, in place
            stop()
        elif wl == "mxfp4":
            start()
            tab.mx_fused_amax_convert(32, "E2M1")  # every weight of the model in one launch
            stop()
        else:
            start()
            self.mask_tab.mask_2to4()  # every weight of the model in one launch
            stop()

    def run(self, steps, warmup, ramp_s=1.5):
        """Ramp (untimed), `warmup` untimed steps, then exactly `steps` steps between barrier + synchronize on both sides;
        returns the elapsed seconds, MAX over the ranks."""
        import torch.distributed as dist

        # Power-state ramp (not a measured step, not part of the W warm-up steps): a GPU that has been idle starts a
        # bandwidth-bound kernel ~15 % slower than one that has been busy for a second (measured: the same launch takes
        # 5.1 ms as the first work after process start and 4.4 ms later in the same session), so the device is kept busy
        # for ~1.5 s before the contractual warm-up + timed region.
        # The ramp is time-based, so ranks may run different numbers of iterations: it must not contain a collective.
        t_ramp = time.perf_counter()
        while time.perf_counter() - t_ramp < ramp_s:
            self.step(False, collective=False)
            torch.cuda.synchronize()
        for _ in range(warmup):
            self.step(False)

        def barrier():
            if self.use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        self.dom_events = []
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.step(True)
        torch.cuda.synchronize()
        # this rank's own K steps, before it waits for the others at the closing barrier (the step's collectives couple the
        # ranks inside the steps already; what differs here is a rank that is simply slower or idle)
        self.local_elapsed = time.perf_counter() - t0
        barrier()
        elapsed = time.perf_counter() - t0
        if self.use_dist:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = t.item()
        return elapsed


def main():
    args = build_parser().parse_args()
    if args.cpu_baseline_only:
        print(json.dumps(cpu_baseline(args.workload)), flush=True)
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # the contract command without a launcher: become the launcher (one rank per GPU; MOQ_BENCH_DEBUG_ONE_GPU=1 puts
        # all ranks on cuda:0 over gloo).  exec: the launcher's exit code and rank 0's stdout are this command's.
        cmd = launch_command(args.gpus, sys.argv[1:])
        sys.stdout.flush()
        os.execv(cmd[0], cmd)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # MOQ_BENCH_DEBUG_ONE_GPU=1: every rank uses cuda:0 and the collectives go through gloo -- lets the N > 1 control
    # flow (collective counts, barriers, rank-0 reporting) be exercised on a single-GPU box; never a measurement.
    one_gpu_debug = os.environ.get("MOQ_BENCH_DEBUG_ONE_GPU") == "1"
    if one_gpu_debug:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # MOQ_FORCE_DIST=1 with ONE rank: the process group is initialised (backend nccl = RCCL) and every collective of
    # the N > 1 path runs with a world of one -- the first multi-GPU run is then not this script's first RCCL run
    # (tests/test_gpu_dist_nccl.py does the same for the library flows)
    use_dist = world > 1 or os.environ.get("MOQ_FORCE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        # RCCL prints a version banner to the C stdout when the communicator comes up; through a pipe it is block
        # buffered and would surface at exit, AFTER the JSON line.  fd 1 points at stderr until the communicator exists
        # and the C buffers are flushed, so that the one line on stdout is the result line.
        import ctypes

        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if one_gpu_debug:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=dev)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)  # the first collective (lazy parts of the communicator, its banner)
            torch.cuda.synchronize()
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    moa = _moa_import.load()
    from model_optimizer_amd.multi_tensor import SegmentTable

    model_given = args.model is not None
    args.model = args.model or default_model(args.workload, world)
    n_layers = args.layers or MODELS[args.model][2]
    wl = args.workload
    args.inplace = not args.out_of_place
    if n_layers * len(layer_shapes(args.model)) < world:  # (every rank takes this exit together: the count is the same everywhere)
        raise SystemExit(f"bench.py: {n_layers} layer(s) of {args.model} hold {n_layers * len(layer_shapes(args.model))} weight "
                         f"tensors, fewer than the {world} ranks they would be dealt over")
    pool = Pool(moa, wl, args.model, n_layers, dev, rank, world, args.scaling, args.inplace, args.group_mb, use_dist)
    weights, tab, masks = pool.weights, pool.tab, pool.masks
    n_local, n_model_elem, n_elem, n_tensors, weak = pool.n_local, pool.n_model_elem, pool.n_elem, pool.n_tensors, pool.weak
    elapsed = pool.run(args.steps, args.warmup)
    ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731

    ms_per_step = elapsed / args.steps * 1e3
    value = n_elem * 2 / (elapsed / args.steps) / 1e9  # the whole pool's weights: what all ranks processed

    # dominant kernel: average launch duration from the HIP events recorded inside the timed region
    dom_all = [a.elapsed_time(b) for a, b in pool.dom_events]
    dom_ms = sum(dom_all) / len(dom_all)
    alg_bytes_per_elem = ALG_BYTES_PER_ELEM[wl]
    dom_name = DOM_KERNEL[wl]
    achieved = n_local * alg_bytes_per_elem / (dom_ms * 1e-3) / 1e9  # this rank's launch over this rank's tensors
    # PMC traffic comes from the committed single-GPU profile of the same launch over the whole model; a rank that
    # owns a share of the tensors moves that share of it (the kernel's traffic is proportional to its elements)
    traffic, traffic_src = pmc_traffic(wl, args.model, n_layers)
    if traffic is not None and n_local != n_model_elem:
        traffic = int(traffic * n_local / n_model_elem)
        traffic_src = f"{traffic_src}, scaled to rank 0's {n_local / n_model_elem:.4f} of the model's elements"
    roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                "alg_bytes_per_launch": int(n_local * alg_bytes_per_elem), "avg_launch_ms": round(dom_ms, 4),
                "min_launch_ms": round(min(dom_all), 4), "max_launch_ms": round(max(dom_all), 4)}
    try:
        if not args.no_node_probe:
            roofline.update(node_probe(dev))  # this node's own copy / read ceilings, after the timed region
    except Exception as e:  # a reported extra, never a reason to lose the main result
        roofline["node_probe_failed"] = f"{type(e).__name__}: {e}"
    if use_dist:
        # every rank's own launch against the roofline (rank order): the line's `roofline` is rank 0's
        mine = torch.tensor([achieved, dom_ms, roofline.get("node_copy_GBs", 0.0)], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        roofline["per_rank"] = [{"achieved": round(t[0].item(), 1), "frac": round(t[0].item() / HBM_PEAK_GBS, 4),
                                 "avg_launch_ms": round(t[1].item(), 4), "node_copy_GBs": round(t[2].item(), 1)}
                                for t in every]

    def describe(p, model):
        what = ("2:4 magnitude mask (1-byte masks written)" if wl == "mask24"
                else "2:4 mask + masked weights + their abs-max in one pass, then FP8 quantize-dequantize, in place" if wl == "fp8-mask24"
                else wl + " calibrate + quantize-dequantize" + (" in place" if args.inplace else ""))
        return (f"{model} all {p.n_tensors // (world if p.weak else 1)} linear weights"
                f"{f' x {world} (one set per GPU)' if p.weak else ''} ({p.n_elem * 2 / 1e9:.2f} GB bf16), {what}, "
                f"inputs resident in HBM")

    out = {
        "metric": f"GB/s weights calibrated+QDQ ({MODEL_NAMES[args.model]})",
        "value": round(value, 2),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak" if weak else "strong",  # (one GPU: the same thing; the N > 1 default is strong)
        "vs_baseline": None,
        "dtype": "bf16 storage, f32 arithmetic",
        "data": "synthetic",
        "config": {"workload": describe(pool, args.model),
                   "format": wl, "model": args.model, "layers": n_layers,
                   "baseline_config": ({"llama3-8b": "configs[1] (Llama-3-8B FP8 per-tensor, 1 GPU)" if wl == "fp8" else
                                        "configs[2] weight side" if wl == "int4g128" else None,
                                        "mixtral-8x7b": "configs[3] (Mixtral-8x7B FP8 + 2:4, weights sharded over the GPUs)",
                                        "llama3-70b": "configs[4] (Llama-3-70B per-group, weights sharded over the GPUs)"}
                                       [args.model]),
                   "parallelism": (f"a pool of {n_tensors} per-layer weight tensors partitioned over {world} GPUs "
                                   f"({len(weights)} on rank 0{', largest first round-robin' if not weak else ''}); one amax "
                                   f"bucket all-reduce(MAX) of {n_tensors} values, in flight under the QDQ launch")
                                  if world > 1 else "single GPU"},
        "roofline": roofline,
    }

    if use_dist:
        # A line that cannot mislead (no multi-GPU node has ever run this; the driver's will be the first): how many ranks the
        # collective backend REALLY joined (a SUM of ones), over which backend, on how many distinct devices, and every rank's
        # own time for the K steps.  A launch that silently ran one rank, N ranks on one GPU (the gloo debug mode), or with
        # an idle rank shows here -- `multi_gpu_valid` is the conjunction a reader would check by hand.
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        props = torch.cuda.get_device_properties(dev)
        ident = str(getattr(props, "uuid", None) or getattr(props, "pci_bus_id", None) or dev.index)
        mine = [None] * world
        dist.all_gather_object(mine, {"rank": rank, "device_index": dev.index, "device": ident, "name": props.name,
                                      "ms_per_step": round(pool.local_elapsed / args.steps * 1e3, 4),
                                      "tensors": len(weights), "elements": n_local})
        backend = dist.get_backend()
        distinct = len({m["device"] for m in mine})
        out["collective"] = {"backend": backend, "is_rccl": backend == "nccl", "world_size": dist.get_world_size(),
                             "rccl_ranks_seen": int(round(ones.item())), "distinct_devices": distinct, "per_rank": mine,
                             "multi_gpu_valid": bool(backend == "nccl" and int(round(ones.item())) == world == args.gpus
                                                     and distinct == world)}

    extra = {}

    def timed(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        a, b = ev(), ev()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    def other_pool_leg(model, scaling, steps):
        """The same step on another pool (every rank runs it: it contains the step's collectives): timed like the
        headline -- ramp, warm-up, barriers, max over ranks."""
        p = Pool(moa, wl, model, MODELS[model][2] if model != args.model else n_layers, dev, rank, world, scaling,
                 True, 0, use_dist)
        dt = p.run(steps, max(args.warmup, 2), ramp_s=0.5)
        dom = [a.elapsed_time(b) for a, b in p.dom_events]
        dms = sum(dom) / len(dom)
        leg = {"value": round(p.n_elem * 2 / (dt / steps) / 1e9, 2), "unit": "GB/s", "ms_per_step": round(dt / steps * 1e3, 4),
               "steps": steps, "n_gpus": world, "scaling": "weak" if p.weak else "strong",
               "workload": describe(p, model),
               "dominant_kernel_frac_of_8TBs": round(p.n_local * alg_bytes_per_elem / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "dominant_kernel_ms": round(dms, 4)}
        p.release()
        return leg

    if not args.no_extra and world == 1:
        # secondary measurements on the same resident weights (not part of `value`)
        # (order: the passes that leave the weights alone first; the in-place ones -- the product's form -- last, since
        # each rewrites the resident weights onto its own grid)
        if wl in ("fp8", "int8"):
            ms = timed(lambda: tab.calibrate_amax())
            extra["per_tensor_amax"] = {"ms": round(ms, 4), "hbm_GBs": round(n_elem * 2 / ms / 1e6, 1),
                                        "frac_of_8TBs": round(n_elem * 2 / ms / 1e6 / HBM_PEAK_GBS, 4)}
        if wl == "fp8" and args.inplace:
            # the same QDQ launch with separate output tensors (what rounds 1-2 timed): depends on where the allocator puts
            # the 14 GB of outputs relative to the inputs -- pools of one process differ by 10 % on the same box
            # (profiles/r03_output_placement.md) -- which the in-place default does not
            tab_o = SegmentTable(weights, outputs=None)
            tab_o.amax_flat.copy_(tab.amax_flat)
            ms = timed(lambda: tab_o.fake_quant_e4m3())
            extra["qdq_out_of_place"] = {"ms": round(ms, 4), "hbm_GBs": round(n_elem * 4 / ms / 1e6, 1),
                                         "frac_of_8TBs": round(n_elem * 4 / ms / 1e6 / HBM_PEAK_GBS, 4)}
            del tab_o
            torch.cuda.empty_cache()
        if wl == "fp8":
            # the other single-launch configurations of BASELINE.json on the same resident weights: configs[3] (2:4
            # magnitude mask: 2 B read + 1 B mask written per element) and configs[4]'s QDQ (MXFP4 g32: 2 B + 2 B)
            mk = [torch.empty(w.shape, dtype=torch.bool, device=dev) for w in weights]
            mtab = SegmentTable(weights, outputs=mk)
            ms = timed(lambda: mtab.mask_2to4())
            extra["mask_2to4"] = {"ms": round(ms, 4), "weights_GBs": round(n_elem * 2 / ms / 1e6, 1),
                                  "hbm_GBs": round(n_elem * 3 / ms / 1e6, 1),
                                  "frac_of_8TBs": round(n_elem * 3 / ms / 1e6 / HBM_PEAK_GBS, 4)}
            del mtab, mk
            ms = timed(lambda: tab.mx_fused_amax_convert(32, "E2M1"))
            extra["mxfp4_g32_qdq"] = {"ms": round(ms, 4), "weights_GBs": round(n_elem * 2 / ms / 1e6, 1),
                                      "hbm_GBs": round(n_elem * 4 / ms / 1e6, 1),
                                      "frac_of_8TBs": round(n_elem * 4 / ms / 1e6 / HBM_PEAK_GBS, 4)}
        if wl == "fp8" and args.inplace:
            # configs[3] as one step on the same resident weights: mask + apply + abs-max (5 B/element), FP8 QDQ (4 B/element)
            mk = [torch.empty(w.shape, dtype=torch.bool, device=dev) for w in weights]
            mtab = SegmentTable(weights, outputs=mk)

            def sparse_fp8_step():
                mtab.mask_2to4_apply(calibrate=True)
                tab.amax_flat.copy_(mtab.amax_flat)
                tab.fake_quant_e4m3()

            ms_step = timed(sparse_fp8_step)
            ms = timed(lambda: mtab.mask_2to4_apply(calibrate=True))
            extra["fp8_mask24_step"] = {"ms_per_step": round(ms_step, 4), "weights_GBs": round(n_elem * 2 / ms_step / 1e6, 1),
                                        "mask_apply_amax_ms": round(ms, 4),
                                        "mask_apply_amax_frac_of_8TBs": round(n_elem * 5 / ms / 1e6 / HBM_PEAK_GBS, 4)}
            del mtab, mk
        if wl != "int4g128":
            tabg = SegmentTable(weights, outputs=tab.outputs, group_size=128)
            ms = timed(lambda: tabg.amax_qdq_int_group(4, False, False))
            b = n_elem * (4.0 + 4.0 / 128)
            extra["int4g128_fused_amax_qdq"] = {"ms": round(ms, 4), "weights_GBs": round(n_elem * 2 / ms / 1e6, 1),
                                                "hbm_GBs": round(b / ms / 1e6, 1),
                                                "frac_of_8TBs": round(b / ms / 1e6 / HBM_PEAK_GBS, 4)}
            del tabg
    if not args.no_extra:
        # release the main workload's tensors: the extras below bring their own
        del tab, weights, masks
        pool.release()
    if not args.no_extra and world == 1:
        # north_star's "histogram collection": |x| histogram (2048 bins, range = the tensor's abs-max) of a calibration-sized
        # activation [131072 tokens, 8192] bf16 = 2.1 GB with per-channel spread and four massive channels -- 2 B/element;
        # round 6: LDS counters per |x| pattern, binned once per pattern at the flush (profiles/r06e_hist_pattern_counters.md
        # has the kernel-only durations at the 67 MB a forward presents per batch, where a Python event pair times the host)
        try:
            gh = torch.Generator(device=dev).manual_seed(7)
            chan = torch.exp(torch.randn(8192, generator=gh, device=dev))
            chan[:4] *= 50.0
            xh = (torch.randn(131072, 8192, generator=gh, device=dev) * chan).to(torch.bfloat16)
            edge = float(xh.float().abs().max())
            counts = torch.zeros(2048, dtype=torch.int64, device=dev)
            ms = timed(lambda: moa.ops.hist_abs(xh, 2048, edge, counts=counts), reps=10)
            ok = int(counts.sum()) == 11 * xh.numel()
            extra["hist_abs_2048bins"] = {"ms": round(ms, 4), "activation_GB": round(xh.numel() * 2 / 1e9, 2),
                                          "hbm_GBs": round(xh.numel() * 2 / ms / 1e6, 1),
                                          "frac_of_8TBs": round(xh.numel() * 2 / ms / 1e6 / HBM_PEAK_GBS, 4),
                                          "every_element_counted": ok}
            del xh, counts, chan
        except Exception as e:  # a reported extra, never a reason to lose the main result
            extra["hist_abs_2048bins"] = {"failed": f"{type(e).__name__}: {e}"}
    if not args.no_extra and use_dist and not weak and wl != "mxfp4-sq":
        # the WEAK leg of a multi-GPU run (every rank holds the whole model; not a BASELINE configuration)
        try:
            extra["weak_scaling"] = other_pool_leg(args.model, "weak", args.steps)
        except torch.cuda.OutOfMemoryError as e:  # (same sizes on every rank: all of them land here together)
            extra["weak_scaling"] = {"failed": f"{type(e).__name__}"}
    if not args.no_extra and world == 1 and not model_given and SCALE_MODEL[wl] != args.model:
        # N = 1 base of the strong-scaling curve the N > 1 default runs: the same step over the whole named model
        try:
            free, _ = torch.cuda.mem_get_info(dev)
            big = SCALE_MODEL[wl]
            need = sum(r * c for _ in range(MODELS[big][2]) for r, c in layer_shapes(big)) * 2
            if free > need * 1.08:
                extra["scale_base_n1"] = other_pool_leg(big, "strong", min(args.steps, 10))
        except Exception as e:  # a reported extra, never a reason to lose the main result
            extra["scale_base_n1"] = {"failed": f"{type(e).__name__}: {e}"}
    if not args.no_extra and world == 1 and args.model == "llama3-8b":
        # north-star target (BASELINE.json): per-group amax + QDQ over ALL Llama-3-70B weight tensors on one GPU, in
        # place (136.9 GB of weights; out of place would need 274 GB)
        try:
            free, _ = torch.cuda.mem_get_info(dev)
            if free > 150e9:
                w70, _, _ = make_weights("llama3-70b", MODELS["llama3-70b"][2], dev)
                n70 = sum(w.numel() for w in w70)
                t70 = SegmentTable(w70, outputs=w70, group_size=128)
                ms = timed(lambda: t70.amax_qdq_int_group(4, False, False), reps=3)
                b = n70 * (4.0 + 4.0 / 128)
                extra["llama3_70b_int4g128_inplace"] = {
                    "ms": round(ms, 3), "weights_GB": round(n70 * 2 / 1e9, 2), "weights_GBs": round(n70 * 2 / ms / 1e6, 1),
                    "hbm_GBs": round(b / ms / 1e6, 1), "frac_of_8TBs": round(b / ms / 1e6 / HBM_PEAK_GBS, 4)}
                # BASELINE configs[4] on the same tensors: SmoothQuant's fold composed with MXFP4 g32, ONE launch in place
                # (4 B/element; the two-pass form of rounds 2-5 moved 8)
                try:
                    gs = torch.Generator(device=dev).manual_seed(99)
                    svs = [torch.exp(0.5 * torch.randn(w.shape[1], generator=gs, device=dev, dtype=torch.float32)) for w in w70]
                    tmx = SegmentTable(w70, outputs=w70)
                    sides = [tmx.fold_side(svs, 32), tmx.fold_side([1.0 / v for v in svs], 32)]
                    flip = [0]

                    def fold_step():
                        tmx.fold_mx_fused(block_size=32, fmt="E2M1", side=sides[flip[0] & 1])
                        flip[0] += 1

                    ms = timed(fold_step, reps=4)
                    extra["llama3_70b_mxfp4_sq"] = {
                        "ms": round(ms, 3), "weights_GB": round(n70 * 2 / 1e9, 2), "weights_GBs": round(n70 * 2 / ms / 1e6, 1),
                        "hbm_GBs": round(n70 * 4.0 / ms / 1e6, 1), "frac_of_8TBs": round(n70 * 4.0 / ms / 1e6 / HBM_PEAK_GBS, 4),
                        "launches_per_step": 1, "kernel": "mt_fold_mx_kernel<bf16, 4, E2M1>"}
                    del tmx, sides, svs
                except Exception as e:
                    extra["llama3_70b_mxfp4_sq"] = {"failed": f"{type(e).__name__}: {e}"}
                del t70, w70
                torch.cuda.empty_cache()
        except Exception as e:  # a reported extra, never a reason to lose the main result
            extra["llama3_70b_int4g128_inplace"] = {"failed": f"{type(e).__name__}: {e}"}

    def cpu_baseline_inproc():
        try:
            return cpu_baseline(wl)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU result
            return {"value": None, "unit": "GB/s", "cores": os.cpu_count(), "kind": "port",
                    "sample": f"failed: {type(e).__name__}: {e}"}

    def cpu_baseline_subprocess():
        """The CPU baseline in a process of its own (`--cpu-baseline-only`): the oracle's OpenMP runtime (the system
        libgomp) and torch's bundled one never meet inside a process that times a flow, and nothing of it is resident
        while the GPU extras run."""
        import subprocess

        env = {k: v for k, v in os.environ.items()
               if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK",
                            "LOCAL_WORLD_SIZE", "OMP_NUM_THREADS", "TORCHELASTIC_RUN_ID")}
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", wl],
                               capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError(f"rc {r.returncode}: {r.stderr.strip()[-300:]}")
            return json.loads(lines[-1])
        except Exception as e:
            return {"value": None, "unit": "GB/s", "cores": os.cpu_count(), "kind": "port",
                    "sample": f"failed: {type(e).__name__}: {e}"}

    if rank == 0:
        out["cpu_baseline"] = None
        if args.cpu_baseline_inproc_first and not args.no_cpu_baseline:
            out["cpu_baseline"] = dict(cpu_baseline_inproc(), order="in process, before the AWQ extras (diagnostic)")
        elif world > 1 and not args.no_cpu_baseline:
            # N > 1: BEFORE the AWQ extra (still in a process of its own), so that the line the watchdog prints if that
            # flow's first multi-rank collectives never return carries it; the other ranks wait at the flow's first collective
            out["cpu_baseline"] = cpu_baseline_subprocess()

    watchdog = None
    if world > 1 and not args.no_extra and args.awq_layers > 0:
        # The AWQ flow below is the first time its collectives (object gathers, GiB-sized Gram reduces) meet more than
        # one rank on hardware.  A collective that never returns must not take the measured headline with it: if the flow
        # has not finished in time, rank 0 prints the line without it and every rank leaves.
        import threading

        def give_up(limit=args.awq_watchdog_s):
            if rank == 0:
                o = dict(out)
                o["extra"] = dict(extra, awq_wallclock_s=None,
                                  awq={"failed": f"watchdog: the {world}-rank AWQ flow had not finished after {limit} s"})
                print(json.dumps(o), flush=True)
            os._exit(0)

        watchdog = threading.Timer(args.awq_watchdog_s, give_up)
        watchdog.daemon = True
        watchdog.start()
    if not args.no_extra and args.awq_layers > 0:
        # the second half of BASELINE.json's metric: INT4-AWQ PTQ wall-clock (awq_lite g128, alpha_step 0.1, default
        # search) of the synthetic Llama-3-8B linear stack (configs[2]); every rank holds the linears, the calibration
        # batches (4096 tokens each) are dealt over the ranks, statistics travel in bucketed all-reduces
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import awq_bench

            line = awq_bench.run(moa, "llama3-8b", args.awq_layers, args.awq_batches, 4096, "auto", dev, rank, world)
            line.pop("best_alphas", None)
            extra["awq_wallclock_s"] = line["value"]
            extra["awq"] = {k: line[k] for k in ("config", "search", "rescored_linears", "rescored_candidates",
                                                 "search_gemm_TFLOPs_equiv", "best_alpha_hist", "passes", "stages_s",
                                                 "quantize_stages_s", "unstaged_s", "awq_unstaged_s",
                                                 "tie_check", "replayed_passes", "forward_loop_calls", "warm_forward_s", "rehearsal_s", "allocator_reserve",
                                                 "stored_input_bytes") if k in line}
        except Exception as e:
            extra["awq_wallclock_s"] = None
            extra["awq"] = {"failed": f"{type(e).__name__}: {e}"}
        if watchdog is not None:
            watchdog.cancel()
    if not args.no_extra and args.awq_layers > 0 and world == 1 and not args.no_hf:
        # the other AWQ case: a random-init Hugging Face Llama-3-8B (real decoder topology: attention, norms, the
        # inputs of q/k/v and gate/up shared), whose activations have no outlier channels -- all 11 candidates of a linear
        # score within a fraction of a percent, the worst case for the exact re-scoring (tools/hf_flow_check.py)
        try:
            # (no empty_cache here: memory handed back to the driver is wiped in the background at ~25 GB/s, and an allocation
            # that needs those pages waits for it -- seconds, inside whatever is being timed; the flow below is served from
            # torch's cache, tools/awq_bench.py has the measurement)
            if os.path.join(ROOT, "tools") not in sys.path:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
            import hf_flow_check

            hf = hf_flow_check.run(hf_flow_check.parse_args(["--layers", str(args.awq_layers), "--batches",
                                                             str(args.awq_batches), "--qformat", "int4_awq"]), moa, dev)
            extra["awq_hf_random_init"] = {"quantize_s": hf["quantize_s"], "plain_forward_loop_s": hf["plain_forward_loop_s"],
                                           "rescored_linears": hf.get("awq_rescored_linears"),
                                           "rescored_candidates": hf.get("awq_rescored_candidates"),
                                           "quantize_stages_s": hf.get("quantize_stages_s"),
                                           "stats": hf.get("awq_stats")}
        except Exception as e:
            extra["awq_hf_random_init"] = {"failed": f"{type(e).__name__}: {e}"}
    if (not args.no_extra and args.awq_layers > 0 and world == 1 and isinstance(extra.get("awq"), dict)
            and "failed" not in extra["awq"]):
        # BASELINE's metric is "INT4-AWQ PTQ wall-clock": `awq_wallclock_s` above is the flow with its first-use costs paid
        # ahead of the clock (a plain forward, a one-layer rehearsal, the flow's memory taken from the driver -- all printed).
        # This is the OTHER figure: a process of its own that imports, builds the same stack and calls quantize() ONCE, with
        # none of that -- what a user's first call pays (code-object loads, first touches, the allocator).  This process
        # hands its memory back first and waits until the driver serves a large allocation promptly again (freed VRAM is
        # wiped in the background at ~25 GB/s, tools/alloc_wipe_probe.py): the harness's own 230 GB must not be in the figure.
        try:
            extra["awq"]["cold_process"] = awq_cold_subprocess(args, dev)
        except Exception as e:
            extra["awq"]["cold_process"] = {"failed": f"{type(e).__name__}: {e}"}
    if rank == 0 and not args.no_cpu_baseline and out.get("cpu_baseline") is None:
        # N = 1: LAST, and in a process of its own
        out["cpu_baseline"] = cpu_baseline_subprocess()
    if extra:
        out["extra"] = extra

    if use_dist:
        # tear the communicator down with fd 1 on stderr as well (anything RCCL still has to say), THEN print: the result
        # line is the last thing on rank 0's stdout
        import ctypes

        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.destroy_process_group()
            ctypes.CDLL(None).fflush(None)
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
