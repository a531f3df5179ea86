#!/bin/bash
set -u
O=gpurun_out/r03p; mkdir -p $O
export TMPDIR=/tmp
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
timeout 200 python tools/exp/sgpt_trailing_probe.py > $O/probe.md 2>&1
R=$(pwd); cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "sgpt_trailing" -f csv -d $R/$O/pmc -o tr -- python $R/tools/exp/sgpt_trailing_probe.py > $R/$O/pmc.log 2>&1
cd $R
python - <<'PY' > $O/pmc_summary.txt
import csv, glob
from collections import defaultdict
acc=defaultdict(list)
for f in glob.glob('gpurun_out/r03p/pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)): acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(acc.items()): print(k, sum(v)/len(v), len(v))
PY
find $O/pmc -type f ! -name '*.csv' -delete 2>/dev/null
cat $O/probe.md $O/pmc_summary.txt
