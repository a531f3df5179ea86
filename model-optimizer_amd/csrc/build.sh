#!/bin/bash
# Build libmoquant.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU present.
#   -ffp-contract=off : the reference results depend on separately rounded mul / rint / div
#   -fhip-fp32-correctly-rounded-divide-sqrt : IEEE fp32 division (default, stated explicitly)
set -euo pipefail
cd "$(dirname "$0")"
OUT=${1:-libmoquant.so}
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -Wall -Wno-unused-function ${MOQ_EXTRA_FLAGS:-}"
objs=()
pids=()
# MOQ_EXPERIMENTS=1: the same sources compiled with -DMOQ_EXPERIMENTS -- the MOQ_TUNE_* knobs of moq_common.h become
# live environment reads (a library for A/B studies from tools/, never the one that ships; own object directory)
SRCS=(moq_*.hip)
OBJDIR=build
if [ "${MOQ_EXPERIMENTS:-0}" = "1" ]; then
  OBJDIR=build/exp
  FLAGS="$FLAGS -DMOQ_EXPERIMENTS -I."
  OUT=${1:-libmoquant_exp.so}
  echo "[moquant] EXPERIMENT build -> $OUT"
fi
for src in "${SRCS[@]}"; do
  base=$(basename "$src")
  obj="$OBJDIR/${base%.hip}.o"
  mkdir -p "$OBJDIR"
  if [ ! -f "$obj" ] || [ "$src" -nt "$obj" ] || [ moq_common.h -nt "$obj" ] || [ moq_chunk.h -nt "$obj" ] || [ moq_mx.h -nt "$obj" ] || [ ../../include/moquant.h -nt "$obj" ]; then
    echo "[moquant] hipcc $src"
    $HIPCC $FLAGS -c "$src" -o "$obj" &
    pids+=($!)
  fi
  objs+=("$obj")
done
for pid in "${pids[@]:-}"; do
  if [ -n "$pid" ]; then wait "$pid" || { echo "[moquant] compile failed"; exit 1; }; fi
done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT" "${objs[@]}"
echo "[moquant] built $(pwd)/$OUT"
