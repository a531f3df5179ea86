#!/bin/bash
# Stage the reference's Python package for the GPU box -- TEST INFRASTRUCTURE ONLY.
#
# /root/reference exists in the build container and nowhere else.  The reference's native sources for this path are CUDA
# (.cu needing nvcc / cub / c10) and cannot be compiled, but its Python / eager path is what its own GPU tests call
# normative (tests/gpu/torch/quantization/test_tensor_quant_cuda.py:55-119, atol = 0 against eager).  This recipe packs
# that Python package, as it lies, into ONE gitignored archive under oracle/_ref/ (listed in .gitignore, not in
# .gpurunignore: it travels with `gpurun` like the built .so files and never enters the history):
#
#     oracle/_ref/reference_modelopt.tgz   <-  /root/reference/{modelopt, modelopt_recipes} and, so that the reference's OWN
#                                              GPU tests of this path can run unmodified on top of our library,
#                                              tests/{conftest.py, _test_utils, gpu/conftest.py, gpu/torch/quantization, gpu/torch/export}
#
# Nothing in model-optimizer_amd/, include/ or the timed region of bench.py reads it.  Readers:
#   * tests/golden/ref_shim.py   -- unpacks it into a temp dir when /root/reference is absent (the GPU box), so that
#                                   tests/test_gpu_reference_live.py can run modelopt_plugin.install() + the reference's own
#                                   mtq.quantize(model.cuda(), ...) on the MI355X;
#   * bench.py --cpu-baseline-only -- times the reference's eager CPU path on the GPU box's host cores
#                                   (cpu_baseline.kind = "reference").
# No source file of the reference is copied into the tree; `rm -rf oracle/_ref` undoes this script.
set -eu
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF="${REFERENCE_ROOT:-/root/reference}"
OUT="$ROOT/oracle/_ref"
if [ ! -d "$REF/modelopt" ]; then
  echo "stage_reference: $REF/modelopt not present (only the build container holds the reference); nothing staged" >&2
  exit 0
fi
mkdir -p "$OUT"
# deterministic archive: sorted names, fixed mtime / owner, no compiled files
tar --sort=name --mtime='2020-01-01 00:00:00' --owner=0 --group=0 --numeric-owner \
    --exclude='__pycache__' --exclude='*.pyc' --exclude='*.cu' --exclude='*.cuh' \
    -C "$REF" -czf "$OUT/reference_modelopt.tgz.tmp" modelopt modelopt_recipes \
    tests/conftest.py tests/_test_utils tests/gpu/conftest.py tests/gpu/torch/quantization tests/gpu/torch/export
mv "$OUT/reference_modelopt.tgz.tmp" "$OUT/reference_modelopt.tgz"
( cd "$REF" && git rev-parse HEAD 2>/dev/null || echo unknown ) > "$OUT/reference_revision.txt"
ls -l "$OUT/reference_modelopt.tgz" | awk '{print "staged", $NF, $5, "bytes"}'
