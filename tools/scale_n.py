"""The driver's multi-GPU scaling run, reproduced: `bench.py` at N = 1, 2, 4, 8 back to back, each launched EXACTLY as the
driver launches it (N = 1: `python bench.py --gpus 1 --steps K --warmup W`; N > 1: `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`), one JSON line per N,
collected into one SCALE-shaped object: {"runs": [{"n_gpus", "value", "unit", "ms_per_step", "scaling", "wall_s", "rc",
"collective", "line"}], ...}.  Scaling efficiency is the DRIVER's to compute from the per-N values; this tool reports none.

    python tools/scale_n.py --gpus 1,2,4,8 --steps 20 --warmup 5 --out gpurun_out/SCALE_local.json
    python tools/scale_n.py --dry-run                      # print the command lines only (CPU tier: tests/test_bench_cpu.py)
    MOQ_BENCH_DEBUG_ONE_GPU=1 python tools/scale_n.py --gpus 1,2 --extra-args "--layers 2 --no-extra --no-cpu-baseline"
        # control flow of N > 1 on ONE GPU over gloo: never a measurement, and the lines say so (collective.multi_gpu_valid false)
"""

import argparse
import json
import os
import shlex
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline")


def command(n: int, steps: int, warmup: int, port: int, extra: list) -> list:
    tail = [os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(steps), "--warmup", str(warmup)] + extra
    if n == 1:
        return [sys.executable] + tail
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port)] + tail


def check_line(line: dict, n: int) -> list:
    """What a SCALE record needs of a line; returns the list of problems (empty: fine)."""
    bad = [f"missing field {k}" for k in REQUIRED if k not in line]
    if line.get("n_gpus") != n:
        bad.append(f"n_gpus {line.get('n_gpus')} != {n}")
    if n > 1:
        col = line.get("collective") or {}
        if not col:
            bad.append("no `collective` object on an N > 1 line")
        elif not col.get("multi_gpu_valid"):
            bad.append(f"not a multi-GPU measurement: backend {col.get('backend')}, ranks seen {col.get('rccl_ranks_seen')}, "
                       f"distinct devices {col.get('distinct_devices')}")
    return bad


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--port", type=int, default=29621)
    ap.add_argument("--extra-args", default="", help="appended to every bench.py command line")
    ap.add_argument("--timeout", type=int, default=1800)
    ap.add_argument("--dry-run", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args(argv)
    extra = shlex.split(args.extra_args)
    runs = []
    for i, n in enumerate(int(x) for x in args.gpus.split(",")):
        cmd = command(n, args.steps, args.warmup, args.port + i, extra)
        if args.dry_run:
            print(" ".join(shlex.quote(c) for c in cmd))
            continue
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.timeout, cwd=ROOT)
        wall = round(time.time() - t0, 1)
        rec = {"n_gpus": n, "rc": r.returncode, "wall_s": wall, "cmd": " ".join(cmd[1:])}
        try:
            line = json.loads(r.stdout.strip().splitlines()[-1])
            rec.update({k: line.get(k) for k in ("value", "unit", "ms_per_step", "scaling", "metric")})
            rec["collective"] = line.get("collective")
            rec["problems"] = check_line(line, n)
            rec["line"] = line
        except Exception as e:  # noqa: BLE001
            rec["problems"] = [f"no JSON line on stdout: {type(e).__name__}: {e}"]
            rec["stderr_tail"] = r.stderr[-600:]
        runs.append(rec)
        print(json.dumps({k: v for k, v in rec.items() if k != "line"}), flush=True)
    if args.dry_run:
        return 0
    out = {"runs": runs, "note": "scaling efficiency is computed by the driver from the per-N values"}
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(out, f)
    return 0 if all(r["rc"] == 0 for r in runs) else 1


if __name__ == "__main__":
    sys.exit(main())
