#!/bin/bash
# (gpurun call 23 of round 6) pattern-counter histogram with the batched / wave-summed flush: parity, then the fixed-cost
# breakdown again (experiment library; MOQ_TUNE_IQ_DBG as in call 22) and the full size table with the release library
set -u
O=gpurun_out/${1:-r06c23}; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_input_quant.py tests/test_gpu_host.py tests/test_gpu_fuzz.py -m gpu -q --tb=short -x -k "hist or input_quant or calib or fused_pass" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
ROOT=$(pwd)
cd /tmp
for dbg in 0 1 2; do
  MOQ_LIB_PATH=$ROOT/model-optimizer_amd/csrc/libmoquant_exp.so HIST_BENCH_MB=8.4,67.1 MOQ_TUNE_IQ_DBG=$dbg rocprofv3 --kernel-trace -f csv -d $ROOT/$O/d$dbg -o hist -- python3 $ROOT/tools/hist_bench.py run --mode hist > $ROOT/$O/d$dbg.log 2>&1
  echo "## dbg=$dbg rc=$?"; HIST_BENCH_MB=8.4,67.1 python3 $ROOT/tools/hist_bench.py parse $ROOT/$O/d$dbg --mode hist | tail -4
done
rocprofv3 --kernel-trace -f csv -d $ROOT/$O/pat -o hist -- python3 $ROOT/tools/hist_bench.py run > $ROOT/$O/pat.log 2>&1; echo "pat rc=$?"
cd $ROOT
echo "## pattern counters (release library)"; python3 tools/hist_bench.py parse $O/pat | tee $O/pat.md
find $O -type f ! -name '*.md' ! -name '*.log' -delete 2>/dev/null
