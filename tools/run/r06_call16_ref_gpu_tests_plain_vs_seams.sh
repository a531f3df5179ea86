#!/bin/bash
# (gpurun call 16 of round 6) EVERY file of the reference's tests/gpu/torch/{quantization,export} three ways: plain (its own
# eager / extension-less path), + kernel seams, + algorithm seam; what passes plain must pass with the seams
set -u
O=gpurun_out/${1:-r06c16}; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python3 - "$O" <<'P'
import os, sys, json
sys.path.insert(0, "tests"); sys.path.insert(0, "tests/golden")
import _moa_import; _moa_import.load()
import test_gpu_reference_live as L
import ref_shim
root = ref_shim.reference_root()
skip = ("gpt_oss", "fsdp", "deepspeed", "onnx", "tp.py", "diffusers", "vllm", "torch_export", "unified_hf_export_and_check")
files = []
for d in ("quantization", "quantization/plugins", "export"):
    for f in sorted(os.listdir(os.path.join(root, "tests/gpu/torch", d))):
        if f.startswith("test_") and f.endswith(".py") and not any(s in f for s in skip):
            files.append(f"{d}/{f}")
print(len(files), "files")
res = {}
for mode, kw in (("plain", dict(seams=False)), ("seams", dict(seams=True)), ("s7", dict(seams=True, algorithms=True))):
    counts, outcomes, out = L.run_reference_tests(files, timeout=2400, **kw)
    res[mode] = outcomes
    open(os.path.join(sys.argv[1], f"ref_gpu_{mode}.txt"), "w").write(out)
    print(mode, counts)
json.dump(res, open(os.path.join(sys.argv[1], "ref_gpu_outcomes.json"), "w"))
for mode in ("seams", "s7"):
    reg = sorted(t for t, v in res["plain"].items() if v == "PASSED" and res[mode].get(t) != "PASSED")
    fixed = sorted(t for t, v in res[mode].items() if v == "PASSED" and res["plain"].get(t) not in ("PASSED",))
    print(mode, "regressed vs plain:", len(reg), reg[:30])
    print(mode, "pass only with seams:", len(fixed))
P
