#!/bin/bash
# PMC pass over the error GEMM (tools/exp/gemm_pmc.py: 4096 x 14336 x 4096 bf16) for one or more loop structures.
# Usage (GPU box, repo root): tools/exp/gemm_pmc.sh <out-tag> [GEO ...]      -> gpurun_out/prof/<tag>_geo<G>_pmc.md
set -uo pipefail
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for G in "$@"; do
  export MOQ_TUNE_GEMM_GEO=$G
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum \
    --kernel-trace --kernel-include-regex "err_gemm" -f csv -d "$OUT/${TAG}_geo${G}" -o "$TAG" -- python "$ROOT/tools/exp/gemm_pmc.py" > "$OUT/${TAG}_geo${G}.log" 2>&1
  echo "[pmc] geo $G rc=$?"
  python - "$OUT/${TAG}_geo${G}" "$G" > "$OUT/${TAG}_geo${G}_pmc.md" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d, g = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
acc = defaultdict(list)
for path in f:
    for r in csv.DictReader(open(path)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"## GEO {g}: averages per dispatch over {max(len(v) for v in acc.values()) if acc else 0} dispatches\n")
avg = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(avg):
    print(f"* {k}: {avg[k]:.4g}")
if "SQ_WAVE_CYCLES" in avg:
    wc = avg["SQ_WAVE_CYCLES"]
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
        if k in avg:
            print(f"* {k} / SQ_WAVE_CYCLES = {avg[k] / wc:.3f}")
if "GRBM_GUI_ACTIVE" in avg and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
    print(f"* MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) = {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (avg['GRBM_GUI_ACTIVE'] * 1024):.3f}")
    if "SQ_LDS_IDX_ACTIVE" in avg:
        print(f"* LDS array busy = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE x 256 CUs) = {avg['SQ_LDS_IDX_ACTIVE'] / (avg['GRBM_GUI_ACTIVE'] * 256):.3f}")
if "TCC_HIT_sum" in avg:
    print(f"* L2 hit rate = {avg['TCC_HIT_sum'] / (avg['TCC_HIT_sum'] + avg['TCC_MISS_sum']):.3f}")
PY
  cat "$OUT/${TAG}_geo${G}_pmc.md"
done
find "$OUT" -type f ! -name '*.csv' ! -name '*.md' ! -name '*.log' ! -name '*.json' -delete 2>/dev/null
find "$OUT" -name '*agent_info*' -delete 2>/dev/null
exit 0
