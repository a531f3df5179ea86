#!/bin/bash
# (gpurun call 44 of round 5: the last GPU-seconds) this package vs the staged reference on the device, whole flows, at HEAD
O=gpurun_out/r05r; mkdir -p $O
( time timeout 38 python3 -m pytest tests/test_gpu_reference_live.py -m gpu -q -x --tb=short -k "this_package_on_the_device or second_format" > $O/subset.log 2>&1 ) 2> $O/time.txt
echo "rc=$? $(grep real $O/time.txt)"; tail -3 $O/subset.log | cut -c1-300
