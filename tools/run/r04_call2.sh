#!/bin/bash
# round 4, call 2: the 14 GB pool's placement -- timing of differently placed copies of the same weights in ONE process, then
# rocprofv3 PMC passes over the same process layout (one FP8 QDQ dispatch pair per set, in set order)
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
python3 tools/pool_placement.py --out $O/timing.json > $O/timing.log 2> $O/timing.err
echo "timing rc=$?"; cat $O/timing.log
python3 tools/pool_placement.py --out $O/timing2.json > $O/timing2.log 2> $O/timing2.err
echo "timing2 rc=$?"; cat $O/timing2.log
cd /tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1
python3 - "$O" <<'P' > $O/passes.txt
import re, sys
avail = open(sys.argv[1] + "/counters_avail.txt").read()
names = set(re.findall(r"\b([A-Z][A-Za-z0-9_]{3,})\b", avail))
wish = [
 ["TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_RDREQ_sum"],
 ["TCC_EA0_RDREQ_32B_sum", "TCC_TAG_STALL_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_TOO_MANY_EA_WRREQS_STALL_sum"],
 ["TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_REQUEST_sum", "TCP_PENDING_STALL_CYCLES_sum"],
 ["TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_DRAM_sum", "TCC_HIT_sum", "TCC_MISS_sum"],
 ["TCC_BUBBLE_sum", "TCC_EA0_WR_UNCACHED_32B_sum", "TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"],
 ["GRBM_GUI_ACTIVE", "SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_INSTS_VMEM_WR", "SQ_INSTS_VMEM_RD"],
 ["TCP_TCC_WRITE_REQ_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_NC_WRITE_REQ_sum", "TCP_TCC_NC_READ_REQ_sum"],
 ["TCC_REQ_sum", "TCC_STREAMING_REQ_sum", "TCC_NC_REQ_sum", "TCC_WRITEBACK_sum"],
]
for group in wish:
    ok = [c for c in group if c in names]
    if ok:
        print(" ".join(ok))
P
cat $O/passes.txt
i=0
while read -r line; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line --kernel-trace --kernel-include-regex "mt_map_kernel" -f csv -d "$O/pmc$i" -o p -- python3 "$ROOT/tools/pool_placement.py" --pmc > "$O/pmc$i.log" 2>&1
  echo "pmc pass $i rc=$? : $line"
done < $O/passes.txt
cd "$ROOT"
find "$O" -type f ! -name '*.csv' ! -name '*.md' ! -name '*.log' ! -name '*.json' ! -name '*.err' ! -name '*.txt' -delete 2>/dev/null
find "$O" -name '*agent_info*' -delete 2>/dev/null
du -sh $O
