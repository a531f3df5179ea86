#!/bin/bash
# HBM traffic (PMC, separate passes) of the kernels matching a regex over tools/kbench.py.
# Usage (GPU box, repo root): tools/exp/pmc_kbench.sh <tag> <kernel-regex>   -> gpurun_out/prof/<tag>_pmc.txt
set -uo pipefail
TAG=$1; RE=$2
R=$(pwd)
OUT=$R/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace --kernel-include-regex "$RE" -f csv -d "$OUT/${TAG}_$C" -o k -- python "$R/tools/kbench.py" > "$OUT/${TAG}_$C.log" 2>&1
  echo "[pmc] $C rc=$?"
done
cd "$R"
python - "$OUT" "$TAG" > "$OUT/${TAG}_pmc.txt" <<'PY'
import csv, glob, collections, sys
out, tag = sys.argv[1], sys.argv[2]
print("# average raw counter per dispatch (FETCH_SIZE / WRITE_SIZE in units of 1 KiB; gfx950: streaming reads are counted at half their bytes -> x2)")
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for p in glob.glob(f"{out}/{tag}_{C}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            if r["Counter_Name"] == C:
                acc[(r["Kernel_Name"][:100], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        avg = sum(v) / len(v)
        mb = avg * 1024 * (2 if C == "FETCH_SIZE" else 1) / 1e6
        print(f"{C} | {k[0]} | grid {k[1]} | {len(v)} dispatches | raw {avg:.1f} | {mb:.1f} MB")
PY
find "$OUT" -type f ! -name '*.txt' ! -name '*.log' ! -name '*.md' ! -name '*.json' -path "*${TAG}*" -delete 2>/dev/null
cat "$OUT/${TAG}_pmc.txt"
