"""Kernel-only durations of the |x| histogram (moq_input_quant's histogram stage / moq_hist_abs) by size and data.

The launches take 10-30 us at the sizes a calibration flow presents, so a Python event pair measures the host's launch rate;
the durations come from a rocprofv3 kernel trace of THIS script, whose dispatch order is fixed (cases in order, `WARM + REPS`
launches each):

    rocprofv3 --kernel-trace -f csv -d <dir> -o hist -- python tools/hist_bench.py run [--mode stats]
    python tools/hist_bench.py parse <dir> [--mode stats]      -> markdown table (avg of the last REPS launches per case)

Experiment library knobs (MOQ_LIB_PATH = libmoquant_exp.so): MOQ_TUNE_HIST_PAT=0 selects the round-2 table kernel instead of
round 6's pattern counters."""

import csv
import glob
import os
import sys

WARM, REPS = 2, 12
SIZES_MB = [float(v) for v in os.environ.get("HIST_BENCH_MB", "8.4,33.5,67.1,268.4,2147.5").split(",")]
DATA = ["outliers", "relu"]
MODES = {"hist": (False,), "stats": (True,), "both": (False, True)}


def cases(mode):
    for with_amax in MODES[mode]:
        for data in DATA:
            for mb in SIZES_MB:
                yield with_amax, data, mb


def run(mode):
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import _moa_import

    ops = _moa_import.load().ops
    dev = "cuda:0"
    torch.manual_seed(7)
    cols = 8192
    chan = torch.exp(torch.randn(cols, device=dev))
    chan[:4] *= 50.0  # a few massive channels set the range: the bulk of |x| sits in the lowest bins
    made = {}
    for with_amax, data, mb in cases(mode):
        rows = int(mb * 1e6 / 2 / cols)
        key = (data, rows)
        if key not in made:
            x = torch.randn(rows, cols, device=dev) * chan
            if data == "relu":
                x = torch.relu(x)  # half the elements are exact zeros
            xb = x.to(torch.bfloat16)
            made[key] = (xb, float(xb.float().abs().max()))
        x, edge = made[key]
        counts = torch.zeros(2048, dtype=torch.int64, device=dev)
        amax = torch.zeros(1, dtype=torch.float32, device=dev) if with_amax else None
        for _ in range(WARM + REPS):
            ops.input_quant(x, None, amax_running=amax, hist_counts=counts, hist_max_edge=edge)
        torch.cuda.synchronize()
        assert os.environ.get("MOQ_TUNE_IQ_DBG") or int(counts.sum()) == (WARM + REPS) * x.numel()
        if mb > 1000:
            made.pop(key)
            del x
            torch.cuda.empty_cache()


def parse(d, mode):
    hits = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not hits:
        print(f"no kernel_trace.csv under {d}")
        return
    with open(hits[0]) as f:
        rows = [r for r in csv.DictReader(f) if "input_quant_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = WARM + REPS
    print("| stage | data | MB (bf16) | avg us | min us | of 8 TB/s (2 B/elem) |")
    print("|---|---|---|---|---|---|")
    for k, (with_amax, data, mb) in enumerate(cases(mode)):
        grp = rows[k * per + WARM:(k + 1) * per]
        if len(grp) < REPS:
            print(f"| (trace ends at case {k}) |")
            break
        us = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in grp]
        avg = sum(us) / len(us)
        print(f"| {'abs-max + histogram' if with_amax else 'histogram'} | {data} | {mb} | {avg:.2f} | {min(us):.2f} | "
              f"{mb * 1e6 / (avg * 1e-6) / 8e12:.3f} |")


if __name__ == "__main__":
    mode = sys.argv[sys.argv.index("--mode") + 1] if "--mode" in sys.argv else "both"
    if sys.argv[1] == "run":
        run(mode)
    else:
        parse(sys.argv[2], mode)
