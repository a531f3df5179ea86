"""Summarise the three rocprofv3 passes of tools/profile_bench.sh for one tag.

Reads   <dir>/<tag>_trace/**/<tag>_kernel_stats.csv         (per-kernel durations)
        <dir>/<tag>_fetch/**/<tag>_counter_collection.csv   (FETCH_SIZE per dispatch)
        <dir>/<tag>_write/**/<tag>_counter_collection.csv   (WRITE_SIZE per dispatch)
Prints a markdown summary and writes <dir>/<tag>_pmc.json with, per kernel: calls, average duration,
average raw FETCH_SIZE / WRITE_SIZE and the corrected HBM byte counts.

Units and corrections (MI355X_MICROARCH.md, "HBM"): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB-like
units of 1024 B; on gfx950 FETCH_SIZE counts 128-B requests of a wide coalesced streaming read as 64 B, i.e.
it reports exactly half of the bytes -> doubled here.  WRITE_SIZE is uncalibrated in the guide; it is reported
raw and as a ratio to the kernel's algorithmic write bytes where the caller supplies them.
"""

import csv
import glob
import json
import os
import sys
from collections import defaultdict


def _find(d, pat):
    hits = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return hits[0] if hits else None


def kernel_stats(path):
    out = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r.get("Name") or r.get("Kernel_Name") or ""
            out[name] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": float(r["TotalDurationNs"]),
                         "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"]), "pct": float(r["Percentage"])}
    return out


def counters(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc


def short(n):
    return n if len(n) <= 100 else n[:97] + "..."


def main(d, tag):
    res = {}
    ks = _find(os.path.join(d, f"{tag}_trace"), "*kernel_stats.csv")
    stats = kernel_stats(ks) if ks else {}
    fe = _find(os.path.join(d, f"{tag}_fetch"), "*counter_collection.csv")
    wr = _find(os.path.join(d, f"{tag}_write"), "*counter_collection.csv")
    fetch = counters(fe, "FETCH_SIZE") if fe else {}
    write = counters(wr, "WRITE_SIZE") if wr else {}
    print(f"# rocprofv3 summary `{tag}`\n")
    print("## kernel-trace --stats\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for n, s in sorted(stats.items(), key=lambda kv: -kv[1]["total_ns"])[:14]:
        print(f"| `{short(n)}` | {s['calls']} | {s['total_ns'] / 1e6:.3f} | {s['avg_ns'] / 1e3:.2f} | "
              f"{s['min_ns'] / 1e3:.2f} | {s['max_ns'] / 1e3:.2f} | {s['pct']:.1f} |")
    print("\n## PMC (separate passes; FETCH_SIZE x 1024 B x 2 [gfx950 half-count correction]; WRITE_SIZE x 1024 B raw)\n")
    print("| kernel | dispatches | avg FETCH_SIZE raw | HBM read bytes (corrected) | avg WRITE_SIZE raw | HBM write bytes (raw) |")
    print("|---|---|---|---|---|---|")
    for n in sorted(set(fetch) | set(write)):
        f_ = fetch.get(n, [])
        w_ = write.get(n, [])
        fa = sum(f_) / len(f_) if f_ else None
        wa = sum(w_) / len(w_) if w_ else None
        rb = fa * 1024 * 2 if fa is not None else None
        wb = wa * 1024 if wa is not None else None
        res[n] = {"dispatches": max(len(f_), len(w_)), "fetch_size_raw_avg": fa, "write_size_raw_avg": wa,
                  "hbm_read_bytes": rb, "hbm_write_bytes_raw": wb,
                  "avg_ns": stats.get(n, {}).get("avg_ns")}
        print(f"| `{short(n)}` | {max(len(f_), len(w_))} | {fa if fa is None else round(fa, 1)} | "
              f"{rb if rb is None else f'{rb:.4g}'} | {wa if wa is None else round(wa, 1)} | {wb if wb is None else f'{wb:.4g}'} |")
    with open(os.path.join(d, f"{tag}_pmc.json"), "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
