#!/bin/bash
# PMC of the library's GEMM kernel and of ours on the same shape.  Usage (GPU box, repo root): tools/exp/blaslt_pmc.sh <tag>
set -uo pipefail
TAG=$1
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof; mkdir -p "$OUT"
export TMPDIR=/tmp; cd /tmp
CNT="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
summ() {
python - "$1" "$2" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d, name = sys.argv[1], sys.argv[2]
acc = defaultdict(list)
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
avg = {k: sum(v) / len(v) for k, v in acc.items()}
print(f"## {name}: averages per dispatch over {max(len(v) for v in acc.values()) if acc else 0} dispatches\n")
for k in sorted(avg): print(f"* {k}: {avg[k]:.4g}")
if "SQ_WAVE_CYCLES" in avg:
    wc = avg["SQ_WAVE_CYCLES"]
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
        if k in avg: print(f"* {k} / SQ_WAVE_CYCLES = {avg[k] / wc:.3f}")
if "GRBM_GUI_ACTIVE" in avg and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
    print(f"* MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs) = {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (avg['GRBM_GUI_ACTIVE'] * 1024):.3f}")
    if "SQ_LDS_IDX_ACTIVE" in avg: print(f"* LDS array busy = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE x 256 CUs) = {avg['SQ_LDS_IDX_ACTIVE'] / (avg['GRBM_GUI_ACTIVE'] * 256):.3f}")
if "TCC_HIT_sum" in avg: print(f"* L2 hit rate = {avg['TCC_HIT_sum'] / (avg['TCC_HIT_sum'] + avg['TCC_MISS_sum']):.3f}; requests per dispatch {avg['TCC_HIT_sum'] + avg['TCC_MISS_sum']:.4g}")
PY
}
rocprofv3 --pmc $CNT --kernel-trace --kernel-include-regex "Cijk" -f csv -d "$OUT/${TAG}_blaslt" -o "$TAG" -- python "$ROOT/tools/exp/blaslt_pmc.py" > "$OUT/${TAG}_blaslt.log" 2>&1
echo "[pmc] blaslt rc=$?"
summ "$OUT/${TAG}_blaslt" "hipBLASLt MT256x256x64 (F.linear 4096 x 14336 x 4096)" > "$OUT/${TAG}_blaslt_pmc.md"
rocprofv3 --pmc $CNT --kernel-trace --kernel-include-regex "err_gemm" -f csv -d "$OUT/${TAG}_ours" -o "$TAG" -- python "$ROOT/tools/exp/gemm_pmc.py" > "$OUT/${TAG}_ours.log" 2>&1
echo "[pmc] ours rc=$?"
summ "$OUT/${TAG}_ours" "err_gemm_kernel GEO 10 (same shape)" > "$OUT/${TAG}_ours_pmc.md"
cat "$OUT/${TAG}_blaslt_pmc.md" "$OUT/${TAG}_ours_pmc.md"
find "$OUT" -type f ! -name '*.csv' ! -name '*.md' ! -name '*.log' ! -name '*.json' -delete 2>/dev/null
find "$OUT" -name '*agent_info*' -delete 2>/dev/null
exit 0
