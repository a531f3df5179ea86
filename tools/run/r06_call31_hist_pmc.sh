#!/bin/bash
# (gpurun call 31 of round 6) HBM read traffic of the pattern-counter histogram (FETCH_SIZE, its own pass with --kernel-trace only)
set -u
O=gpurun_out/${1:-r06c31}; mkdir -p $O
export TMPDIR=/tmp
ROOT=$(pwd)
cd /tmp
HIST_BENCH_MB=67.1,2147.5 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "input_quant" -f csv -d $ROOT/$O/fetch -o hist -- python3 $ROOT/tools/hist_bench.py run --mode hist > $ROOT/$O/fetch.log 2>&1; echo "fetch rc=$?"
cd $ROOT
python3 - $O <<'P'
import csv, glob, sys, os
f = glob.glob(os.path.join(sys.argv[1], "fetch", "**", "*counter_collection.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r.get("Counter_Name") == "FETCH_SIZE" and "input_quant" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
per = 14
cases = [(d, mb) for d in ("outliers", "relu") for mb in (67.1, 2147.5)]
print("| data | MB | FETCH_SIZE avg (raw, KiB) | x 2 (gfx950 wide-read correction) bytes | / algorithmic |")
print("|---|---|---|---|---|")
for k, (d, mb) in enumerate(cases):
    g = rows[k * per + 2:(k + 1) * per]
    if not g: break
    raw = sum(float(r["Counter_Value"]) for r in g) / len(g)
    b = raw * 1024 * 2
    print(f"| {d} | {mb} | {raw:.0f} | {b:.4g} | {b / (mb * 1e6):.4f} |")
P
find $O -type f -name '*.csv' -size +2M -delete 2>/dev/null
