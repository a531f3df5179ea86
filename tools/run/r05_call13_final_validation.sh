#!/bin/bash
# (gpurun call 13 of round 5) HEAD: default bench line as the FIRST command of a fresh lease, whole GPU suite, smoke, the N = 2
# control flow on one GPU (gloo debug mode, never a measurement), kernel breakdown of the HF-topology INT4-AWQ flow, kernel table
set -u
O=gpurun_out/${1:-r05c13}; mkdir -p $O
ROOT=$(pwd); export TMPDIR=/tmp
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench.err ) 2> $O/bench_time.txt
echo "bench rc=$? $(grep real $O/bench_time.txt)"
python3 - "$O" <<'P'
import json, sys
d=json.loads(open(sys.argv[1]+"/bench_default.json").read().strip().splitlines()[-1]); e=d["extra"]
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], d["roofline"].get("traffic_source"), "awq", e.get("awq_wallclock_s"), (e.get("awq") or {}).get("stages_s"), (e.get("awq") or {}).get("passes"))
h=e.get("awq_hf_random_init") or {}
print("hf", h.get("quantize_s"), (h.get("stats") or {}).get("stages_s"))
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
for k in ("per_tensor_amax","mask_2to4","mxfp4_g32_qdq","fp8_mask24_step","int4g128_fused_amax_qdq","llama3_70b_int4g128_inplace","scale_base_n1"):
    print(k, json.dumps(e.get(k))[:220])
P
timeout 1500 python3 -m pytest tests -m gpu -q -n 2 --tb=short > $O/gpu_suite.log 2>&1
echo "suite rc=$?"; grep "passed\|failed\|^E  \|^FAILED" $O/gpu_suite.log | tail -12 | cut -c1-250
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
export MOQ_BENCH_DEBUG_ONE_GPU=1
for wl in fp8 int4g128; do
  timeout 600 python3 bench.py --gpus 2 --steps 3 --warmup 1 --workload $wl --layers 2 --awq-layers 1 --awq-batches 2 > $O/n2_$wl.json 2> $O/n2_$wl.err
  echo "n2 $wl rc=$?"; tail -1 $O/n2_$wl.json | cut -c1-300
done
unset MOQ_BENCH_DEBUG_ONE_GPU
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$O/prof_awq_hf -o awq -- python3 $ROOT/tools/hf_flow_check.py --layers 32 --batches 64 --qformat int4_awq > $ROOT/$O/flow_awq_hf.json 2> $ROOT/$O/flow_awq_hf.err
cd $ROOT
tail -1 $O/flow_awq_hf.json | cut -c1-300
python3 tools/kstats_all_md.py $O/prof_awq_hf 16 > $O/awq_hf_kernels.md; head -22 $O/awq_hf_kernels.md | cut -c1-200
python3 tools/kbench.py 2>&1 | grep -v Warning > $O/kernel_table.md; wc -l $O/kernel_table.md
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $ROOT/$O/prof_small -o small -- python3 $ROOT/tools/kbench.py "67 MB,columns,col_abs,awq_weight_scale,row_hist,scale_cols_multi" > $ROOT/$O/prof_small.log 2>&1
cd $ROOT
python3 tools/kstats_md.py $O/prof_small | tee $O/small_kernels_kernel_only.md | grep "row_hist\|scale_cols_multi\|col_stats\|awq_wscale\|finalize\|accum"
find $O -name "*kernel_trace.csv" -delete; find $O -name "*.db" -delete; find $O -name "*agent_info*" -delete
