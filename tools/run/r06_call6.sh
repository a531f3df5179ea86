#!/bin/bash
# (gpurun call 6 of round 6) after the device-numerics activation mean + headroom + probation: the live-reference file, the host /
# AWQ-search / dist files, the drop-in timing again (INT4-AWQ through S7 should now be 56 / 56), cold AWQ through bench's extra
set -u
O=gpurun_out/${1:-r06c6}; mkdir -p $O
export TMPDIR=/tmp
timeout 2400 python3 -m pytest tests/test_gpu_reference_live.py tests/test_gpu_host.py tests/test_gpu_awq_search.py tests/test_gpu_dist_nccl.py tests/test_gpu_parity.py -m gpu -q --tb=short -n 2 > $O/tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|^E  \|^FAILED\|\[note\] INT4-AWQ\|\[note\] reference INT4\|\[note\] reference W4A8" $O/tests.log | tail -24 | cut -c1-700
timeout 2400 python3 tools/dropin_bench.py --layers 4 --batches 16 --rows 8 --seq 512 --formats int4_awq --out $O/dropin_awq.json > $O/dropin.log 2> $O/dropin.err
echo "dropin rc=$?"; grep "^{\"int" $O/dropin.log | cut -c1-420
timeout 1200 python3 bench.py --gpus 1 --steps 5 --warmup 2 --no-hf --no-cpu-baseline > $O/bench_short.json 2> $O/bench_short.err
python3 - "$O" <<'P'
import json, sys
d=json.loads(open(sys.argv[1]+"/bench_short.json").read().strip().splitlines()[-1]); e=d["extra"]
print(d["value"], "awq", e.get("awq_wallclock_s"), (e.get("awq") or {}).get("passes"), (e.get("awq") or {}).get("tie_check"))
print("cold", json.dumps((e.get("awq") or {}).get("cold_process"))[:1200])
P
