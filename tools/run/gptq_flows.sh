#!/bin/bash
# (gpurun call of round 4) GPTQ at real layer shapes: HF Llama-3-8B layers, INT4 g128 and FP8, whole-model and layer by layer;
# the GPU tests of the algorithm; a kernel trace of one run
set -u
O=gpurun_out/r04x; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_gptq.py tests/test_gpu_sparsegpt.py -x -q 2>&1 | tail -3
for F in int4_gptq int4_gptq_layerwise fp8_gptq int4_mse; do
  timeout 900 python3 tools/hf_flow_check.py --layers 4 --batches 16 --qformat $F > $O/flow_$F.json 2> $O/flow_$F.err
  echo "$F rc=$?"; tail -1 $O/flow_$F.json | cut -c1-900
done
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $GRAFT_REPO_ROOT/$O/trace -o gptq -- python3 $GRAFT_REPO_ROOT/tools/hf_flow_check.py --layers 2 --batches 16 --qformat int4_gptq > $GRAFT_REPO_ROOT/$O/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'P'
import csv, glob
f = glob.glob("gpurun_out/r04x/trace/**/*kernel_stats.csv", recursive=True)
if f:
    rows = list(csv.DictReader(open(f[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("| kernel | calls | total ms | share |\n|---|---|---|---|")
    for r in rows[:14]:
        print(f"| {r['Name'][:90]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['TotalDurationNs'])/tot:.3f} |")
P
find gpurun_out/r04x/trace -type f ! -name '*kernel_stats.csv' -delete 2>/dev/null
