#!/bin/bash
# GEO 21 (GEO 20 + whole K-tile of fragments resident, stage recycled a quarter into its tile): parity + table 10 / 20 / 21
set -u
O=gpurun_out/r03zh; mkdir -p $O
timeout 60 python -c "import torch; x=torch.randn(1<<26,device='cuda'); print('box ok', x.sum().item())" > $O/box.txt 2>&1 || { cat $O/box.txt; exit 0; }
EXP=$PWD/model-optimizer_amd/csrc/libmoquant_exp.so
( MOQ_LIB_PATH=$EXP MOQ_TUNE_GEMM_GEO=22 timeout 900 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_awq_search.py tests/test_gpu_sparsegpt.py -m gpu -q -k "not f32 and not dtype2" 2>&1 | grep "FAILED\|passed\|failed" | cut -c1-200 ) > $O/gpu_tests_geo22.txt
for g in 10 21 22; do
MOQ_LIB_PATH=$EXP MOQ_TUNE_GEMM_GEO=$g timeout 300 python tools/gemm_bench.py > $O/gemm_geo$g.md 2> $O/gemm_geo$g.err
done
cat $O/gpu_tests_geo22.txt; for g in 10 21 22; do echo GEO $g; grep "^| [0-9a-z]" $O/gemm_geo$g.md | cut -d'|' -f2,3,4,6,7,9,10; done
