#!/bin/bash
# round 3, GPU call 2: full suite at HEAD, bench line + rocprof/PMC of the same command in the same session,
# AWQ auto-vs-gemm on the synthetic stack, HF-topology flow with the self-checking margin, SparseGPT table
set -u
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 ) > $O/gpu_tests.txt
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err
timeout 400 bash tools/profile_bench.sh r03_fp8 --workload fp8 > $O/profile.log 2>&1
cp gpurun_out/prof/r03_fp8_summary.md gpurun_out/prof/r03_fp8_pmc.json $O/ 2>/dev/null
timeout 300 python tools/sgpt_bench.py > $O/sgpt_table.md 2> $O/sgpt.err
timeout 300 python tools/awq_bench.py --layers 32 --batches 64 --search auto --compare gemm > $O/awq_auto_vs_gemm.json 2> $O/awq.err
timeout 300 python tools/hf_flow_check.py --layers 32 --batches 64 --qformat int4_awq --dump $O/hf_auto_tables.json --note "r03b self-checking margin" >> $O/hf.jsonl 2> $O/hf.err
rm -rf gpurun_out/prof
ls -la $O
