#!/bin/bash
# (gpurun call of round 4) packers / unpackers / awq weight scale after the shared-division rewrites: the whole GPU suite,
# then tools/kbench.py with the release library and with the experiment library at three occupancy caps of the
# single-tensor read + write launches (MOQ_TUNE_COPY_LDS_1T, moq_chunk.h)
set -u
O=gpurun_out/r04t; mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q -n 2 > $O/gpu_suite.log 2>&1
echo "suite rc=$?"; tail -6 $O/gpu_suite.log
timeout 600 python3 tools/kbench.py > $O/kbench_release.md 2> $O/kbench_release.err
echo "kbench release rc=$?"
for L in 0 24576 32768; do
  MOQ_LIB_PATH=$(pwd)/model-optimizer_amd/csrc/libmoquant_exp.so MOQ_TUNE_COPY_LDS_1T=$L timeout 600 python3 tools/kbench.py > $O/kbench_exp_lds$L.md 2> $O/kbench_exp_lds$L.err
  echo "kbench exp lds=$L rc=$?"
done
python3 - <<'P'
import re
O="gpurun_out/r04t"
def rows(p):
    out={}
    for ln in open(p):
        m=re.match(r"\| (.+?) \| ([\d.]+) \| ([\d.]+) \| ([\d.]+) \|", ln)
        if m: out[m.group(1)[:60]]=float(m.group(4))
    return out
a=rows(f"{O}/kbench_release.md"); b={l:rows(f"{O}/kbench_exp_lds{l}.md") for l in (0,24576,32768)}
for k in a:
    print(f"{k:62s} rel {a[k]:.3f} | exp0 {b[0].get(k,0):.3f} | 24K {b[24576].get(k,0):.3f} | 32K {b[32768].get(k,0):.3f}")
P
