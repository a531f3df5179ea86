#!/bin/bash
# (gpurun call 21 of round 6) the histogram's pattern counters: parity tests, then kernel-only durations by size and data
# (tools/hist_bench.py under a rocprofv3 kernel trace) -- release library (pattern counters), and the experiment library
# with MOQ_TUNE_HIST_PAT=0 (round 2's pattern table) on the same box
set -u
O=gpurun_out/${1:-r06c21}; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_input_quant.py tests/test_gpu_host.py tests/test_gpu_fuzz.py -m gpu -q --tb=short -x -k "hist or input_quant or calib or fused_pass" > $O/tests.log 2>&1
echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-300
ROOT=$(pwd)
cd /tmp
rocprofv3 --kernel-trace -f csv -d $ROOT/$O/pat -o hist -- python3 $ROOT/tools/hist_bench.py run > $ROOT/$O/pat.log 2>&1; echo "pat rc=$?"
MOQ_LIB_PATH=$ROOT/model-optimizer_amd/csrc/libmoquant_exp.so MOQ_TUNE_HIST_PAT=0 rocprofv3 --kernel-trace -f csv -d $ROOT/$O/lut -o hist -- python3 $ROOT/tools/hist_bench.py run > $ROOT/$O/lut.log 2>&1; echo "lut rc=$?"
cd $ROOT
echo "## pattern counters (release library)"; python3 tools/hist_bench.py parse $O/pat | tee $O/pat.md
echo "## pattern table (MOQ_TUNE_HIST_PAT=0)"; python3 tools/hist_bench.py parse $O/lut | tee $O/lut.md
tail -3 $O/pat.log $O/lut.log | cut -c1-300
find $O -name '*.csv' -size +4M -delete 2>/dev/null
find $O -type f ! -name '*.csv' ! -name '*.md' ! -name '*.log' -delete 2>/dev/null
