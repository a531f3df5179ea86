#!/bin/bash
# (gpurun call 22 of round 6) where the pattern-counter histogram's fixed ~8 us go: experiment library, MOQ_TUNE_IQ_DBG
# bit 0 = no pattern -> bin pass, bit 1 = no global flush, bit 3 = no LDS atomics in the sweep (timing only, counts wrong)
set -u
O=gpurun_out/${1:-r06c22}; mkdir -p $O
export TMPDIR=/tmp
ROOT=$(pwd)
export MOQ_LIB_PATH=$ROOT/model-optimizer_amd/csrc/libmoquant_exp.so HIST_BENCH_MB=8.4,67.1
cd /tmp
for dbg in 0 1 2 3 8 11; do
  MOQ_TUNE_IQ_DBG=$dbg rocprofv3 --kernel-trace -f csv -d $ROOT/$O/d$dbg -o hist -- python3 $ROOT/tools/hist_bench.py run --mode hist > $ROOT/$O/d$dbg.log 2>&1
  echo "## dbg=$dbg rc=$?"; HIST_BENCH_MB=$HIST_BENCH_MB python3 $ROOT/tools/hist_bench.py parse $ROOT/$O/d$dbg --mode hist | tail -4
done
cd $ROOT
find $O -type f ! -name '*.md' ! -name '*.log' -delete 2>/dev/null
