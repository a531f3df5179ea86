#!/bin/bash
# Profile one bench.py workload on the GPU box (run from the repo root):
#   1. rocprofv3 --kernel-trace --stats   -> per-kernel durations   (gpurun_out/prof/<tag>_trace)
#   2. rocprofv3 --pmc FETCH_SIZE         -> HBM read  bytes        (gpurun_out/prof/<tag>_fetch)
#   3. rocprofv3 --pmc WRITE_SIZE         -> HBM write bytes        (gpurun_out/prof/<tag>_write)
# PMC passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md: FETCH_SIZE takes 3 of the 4
# TCC slots, WRITE_SIZE 2; gpurun refuses --pmc together with sys/hip/hsa tracing).
# Usage: tools/profile_bench.sh <tag> <bench.py args...>
set -uo pipefail
TAG=$1; shift
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
COMMON="--no-cpu-baseline --no-extra --no-node-probe"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/${TAG}_trace" -o "$TAG" -- python "$ROOT/bench.py" --steps 10 --warmup 2 $COMMON "$@" > "$OUT/${TAG}_trace.log" 2>&1
echo "[profile] trace rc=$?"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "moq" -f csv -d "$OUT/${TAG}_fetch" -o "$TAG" -- python "$ROOT/bench.py" --steps 3 --warmup 1 $COMMON "$@" > "$OUT/${TAG}_fetch.log" 2>&1
echo "[profile] fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "moq" -f csv -d "$OUT/${TAG}_write" -o "$TAG" -- python "$ROOT/bench.py" --steps 3 --warmup 1 $COMMON "$@" > "$OUT/${TAG}_write.log" 2>&1
echo "[profile] write rc=$?"
cd "$ROOT"
python tools/pmc_summary.py "$OUT" "$TAG" > "$OUT/${TAG}_summary.md" 2> "$OUT/${TAG}_summary.err" || true
# keep the merge-back small: drop everything except csv/md/log/json
find "$OUT" -type f ! -name '*.csv' ! -name '*.md' ! -name '*.log' ! -name '*.json' ! -name '*.err' -delete 2>/dev/null
find "$OUT" -name '*agent_info*' -delete 2>/dev/null
exit 0
