#!/bin/bash
# (gpurun call 43 of round 5: the last 80 GPU-seconds) the export / MoE / KV-cache GPU tests at HEAD, after the host-side export changes
O=gpurun_out/r05q; mkdir -p $O
( time timeout 70 python3 -m pytest tests/test_gpu_export.py tests/test_gpu_moe.py tests/test_gpu_kv_cache.py -m gpu -q -x --tb=short > $O/subset.log 2>&1 ) 2> $O/time.txt
echo "rc=$? $(grep real $O/time.txt)"; tail -4 $O/subset.log | cut -c1-300
