#!/bin/bash
# rocprofv3 kernel shares of the two INT4-AWQ flows (8 layers each, 64 batches) and of the FP8 max-calibration flow
set -u
O=$PWD/gpurun_out/r03zd; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
ROOT=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $O/awq_syn -o t -- python $ROOT/tools/awq_bench.py --layers 8 --batches 64 --search auto > $O/awq_syn.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $O/awq_hf -o t -- python $ROOT/tools/hf_flow_check.py --layers 8 --batches 64 --qformat int4_awq > $O/awq_hf.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -f csv -d $O/fp8_hf -o t -- python $ROOT/tools/hf_flow_check.py --layers 8 --batches 64 --qformat fp8 > $O/fp8_hf.log 2>&1
cd $ROOT
for t in awq_syn awq_hf fp8_hf; do python tools/kstats_all_md.py $O/$t 18 > $O/$t.md 2>&1; done
find $O -type f ! -name '*.md' ! -name '*.log' ! -name '*.txt' -delete 2>/dev/null
head -24 $O/awq_syn.md; head -24 $O/awq_hf.md; head -16 $O/fp8_hf.md; grep -h "^{" $O/awq_syn.log $O/awq_hf.log $O/fp8_hf.log | cut -c1-400
