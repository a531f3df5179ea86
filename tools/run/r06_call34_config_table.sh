#!/bin/bash
# (gpurun call 34 of round 6) one bench line per BASELINE configuration on ONE GPU at HEAD (the N = 1 base of every multi-GPU default)
set -u
O=gpurun_out/${1:-r06c34}; mkdir -p $O; rm -f $O/config_lines.jsonl
export TMPDIR=/tmp
run() { python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-extra --no-cpu-baseline "$@" 2>> $O/err.log | tail -1 >> $O/config_lines.jsonl; echo "rc=$? $*"; }
run --workload fp8 --model llama3-8b
run --workload int4g128 --model llama3-8b
run --workload fp8 --model mixtral-8x7b
run --workload mask24 --model mixtral-8x7b
run --workload fp8-mask24 --model mixtral-8x7b
run --workload int4g128 --model llama3-70b
run --workload mxfp4 --model llama3-70b
run --workload mxfp4-sq --model llama3-70b
python3 - $O <<'P'
import json, sys
print("| config | workload | model | GB/s of weights | ms / step | dominant kernel | frac of 8 TB/s |"); print("|---|---|---|---|---|---|---|")
for l in open(sys.argv[1] + "/config_lines.jsonl"):
    d=json.loads(l); c=d["config"]; r=d["roofline"]
    print(f"| {c.get('baseline_config')} | {c['format']} | {c['model']} | {d['value']} | {d['ms_per_step']} | `{r['kernel']}` {r['avg_launch_ms']} ms | {r['frac']} |")
P
