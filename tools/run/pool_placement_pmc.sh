#!/bin/bash
# rocprofv3 PMC passes over tools/pool_placement.py --pmc (every differently placed copy of the weights gets two FP8 QDQ
# dispatches, in set order), each pass reduced ON THE BOX to a small JSON (tools/pmc_reduce.py: the raw CSVs of a pass are
# tens of MB; gpurun merges back at most 64 MiB).  profiles/r04_pool_pmc_box3.json came from this.
set -u
ROOT=$(pwd)
O=$ROOT/gpurun_out/pool_pmc; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
while read -r line; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $line --kernel-trace -f csv -d "$O/pmc$i" -o p -- python3 "$ROOT/tools/pool_placement.py" --pmc --sets 4 > "$O/pmc$i.log" 2>&1
  echo "pmc pass $i rc=$? : $line"
  python3 $ROOT/tools/pmc_reduce.py "$O/pmc$i" mt_map_kernel "$O/pmc$i.json"
  grep pmc_order "$O/pmc$i.log" > "$O/pmc${i}_order.json"
  rm -rf "$O/pmc$i"
done <<'P'
TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_sum
TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum
TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_HIT_sum TCC_MISS_sum
TCC_BUBBLE_sum TCC_EA0_WR_UNCACHED_32B_sum TCC_EA0_RDREQ_GMI_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD
P
cd "$ROOT"; du -sh $O
