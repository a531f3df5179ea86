#!/bin/bash
# how much of the error GEMM is its epilogue's request traffic?  GEO 10 / 20 with and without the loss epilogue (timing only)
set -u
O=gpurun_out/r03zf; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
EXP=$PWD/model-optimizer_amd/csrc/libmoquant_exp.so
for g in 10 20; do for ne in 0 1; do
MOQ_LIB_PATH=$EXP MOQ_TUNE_GEMM_GEO=$g MOQ_TUNE_GEMM_NO_EPILOGUE=$ne timeout 300 python tools/gemm_bench.py > $O/gemm_geo${g}_ne$ne.md 2> $O/gemm_geo${g}_ne$ne.err
echo "GEO $g no_epilogue=$ne"; grep "^| [0-9a-z]" $O/gemm_geo${g}_ne$ne.md | cut -d'|' -f2,3,4,6,7,10
done; done
