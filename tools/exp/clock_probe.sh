#!/bin/bash
# Does a rocprofv3 session change the state the NEXT plain run sees?  (same box, same binary)
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["config"]["format"], d["ms_per_step"], r["avg_launch_ms"], r["frac"])'
run() { python bench.py --steps 20 --no-extra --no-cpu-baseline 2>&1 | tail -1 | python -c "$P"; }
echo "== plain x3"; run; run; run
echo "== tiny rocprofv3 session"; (cd /tmp; TMPDIR=/tmp rocprofv3 --kernel-trace -d /tmp/rp0 -o x -- python -c "import torch; print(torch.zeros(4, device='cuda').sum().item())" > /dev/null 2>&1; echo rc=$?)
echo "== plain x2"; run; run
echo "== bench under rocprofv3"; (cd /tmp; TMPDIR=/tmp rocprofv3 --kernel-trace -d /tmp/rp1 -o x -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --no-extra --no-cpu-baseline 2>/dev/null | grep '^{' | tail -1 | python -c "$P")
echo "== plain x2"; run; run
