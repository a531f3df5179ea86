#!/bin/bash
# round 3, GPU call 1: suite + bench at HEAD, HF-topology INT4-AWQ breakdown, score tables of both engines
set -u
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/gpu_tests.txt
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err
timeout 200 python tools/gemm_bench.py > $O/gemm_bench.txt 2>&1
timeout 300 python tools/hf_flow_check.py --layers 32 --batches 64 --qformat int4_awq --note "r03a HEAD default" >> $O/hf.jsonl 2> $O/hf1.err
timeout 400 python tools/hf_flow_check.py --layers 32 --batches 64 --qformat int4_awq --search auto --tie-margin inf --dump $O/hf_tables_p1.json --note "all candidates, both engines" >> $O/hf.jsonl 2> $O/hf2.err
MOQ_TUNE_GRAM_PLANES=3 timeout 300 python tools/hf_flow_check.py --layers 32 --batches 64 --qformat int4_awq --search gram --dump $O/hf_tables_p3.json --note "gram, 3 planes" >> $O/hf.jsonl 2> $O/hf3.err
R=$(pwd); cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof_hf -o hf -- python $R/tools/hf_flow_check.py --layers 8 --batches 64 --qformat int4_awq --note "8 layers under rocprof" >> $R/$O/hf.jsonl 2> $R/$O/hf4.err
cd $R
find $O/prof_hf -type f ! -name '*stats*.csv' -delete 2>/dev/null
ls -la $O
