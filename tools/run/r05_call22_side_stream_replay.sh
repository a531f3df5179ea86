#!/bin/bash
# (gpurun call 22 of round 5) the replayed exact pass with the candidates' scaled inputs of batch b + 1 written on a second stream
# under batch b's error GEMM: tests that walk the replay, then both AWQ flows
set -u
O=gpurun_out/r05c22; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_awq_search.py tests/test_gpu_export.py tests/test_gpu_gemm.py tests/test_gpu_layerwise.py -m gpu -q -n 2 --tb=short 2>&1 | grep -v "Warning\|warnings.warn\|^  " | tail -6 | cut -c1-300
python3 tools/awq_bench.py --layers 32 --batches 64 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('synthetic', d['value'], d['passes'], d['stages_s'], d['allocator_reserve'])"
python3 tools/hf_flow_check.py --layers 32 --batches 64 --qformat int4_awq 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('hf', d['quantize_s'], d['plain_forward_loop_s'], d.get('awq_stats',{}).get('stages_s') or d.get('quantize_stages_s'))"
