#!/bin/bash
# (gpurun call 24 of round 5) counters behind moq_mse_sweep's VALU roofline: executed vector instructions per launch (against the
# ISA census' slots per element and candidate) and the shader clock while it runs
set -u
O=gpurun_out/r05c24; mkdir -p $O
ROOT=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --kernel-trace --kernel-include-regex "mse_" -f csv -d $ROOT/$O/pmc -o mse -- python3 $ROOT/tools/kbench.py mse_sweep > $ROOT/$O/pmc.log 2>&1
rocprofv3 --kernel-trace --stats --kernel-include-regex "mse_" -f csv -d $ROOT/$O/trace -o mse -- python3 $ROOT/tools/kbench.py mse_sweep > $ROOT/$O/trace.log 2>&1
cd $ROOT
python3 - $O <<'P'
import csv, glob, sys, collections
o = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(o + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void moq::", "")
        per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for f in glob.glob(o + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"].split("(")[0].replace("void moq::", "")] = float(r["AverageNs"])
print("| kernel | dispatches | SQ_INSTS_VALU per launch | SQ_ACTIVE_INST_VALU | GRBM_GUI_ACTIVE (sum of 8 XCDs) | avg us (trace pass) | shader clock |")
print("|---|---|---|---|---|---|---|")
for k, c in sorted(per.items()):
    n = len(c["SQ_INSTS_VALU"])
    a = lambda name: sum(c[name]) / max(len(c[name]), 1)
    us = dur.get(k, 0) / 1e3
    clk = a("GRBM_GUI_ACTIVE") / 8 / us if us else 0
    print(f"| `{k}` | {n} | {a('SQ_INSTS_VALU'):.4g} | {a('SQ_ACTIVE_INST_VALU'):.4g} | {a('GRBM_GUI_ACTIVE'):.4g} | {us:.1f} | {clk:.0f} MHz |")
P
find $O -name "*.csv" -size +1M -delete; find $O -name "*.db" -delete
