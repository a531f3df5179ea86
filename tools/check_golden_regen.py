"""Regenerate tests/golden/*.npz from the reference in a TEMP copy and list the arrays that come out differently from the
committed ones (build container only: needs /root/reference).

    python tools/check_golden_regen.py            # all fixtures (~4 min on 8 cores)

The committed fixtures regenerate bit for bit on the host class that generated them (rounds 1-4; the round-4 review re-ran
the generator and got 36 files / 2348 arrays identical).  They do NOT on every host: torch's CPU kernels are not the same
functions everywhere -- on the machine the build container moved to in round 5 (AVX-512 without bf16 / AMX units)
  * `torch.randn`-drawn INPUTS of some cases differ (amax c18 / c19, fp8_fq c0.., mask24 c6): stored in the fixture next to
    the reference's outputs for them, so every test stays self-consistent;
  * arrays downstream of a bf16 CPU FORWARD of a model differ (oneDNN's bf16 GEMM): `fp8_max_y` of model_flows in one element
    of 3072 by one bf16 ulp, and from there statistics / weights of later layers of the tiny Llama (export_llama*, gptq_llama,
    sq_mxfp4, awq_clip, w4a8).  Every f32 flow regenerates identically.
Tests that run such a forward themselves (tests/test_host_flows_cpu.py, tests/test_gptq_cpu.py: four of them) go through
tests/conftest.py::pinned_or_live -- committed bytes wherever the host meets them, otherwise the reference's LIVE output of
the same generator on this host -- and say so in the suite's [note] lines.  The others feed the fixture's recorded
activations to the code under test (the replay tests) and do not depend on the host."""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = os.path.join(ROOT, "tests", "golden")
    tmp = tempfile.mkdtemp(prefix="golden_regen_")
    dst = os.path.join(tmp, "golden")
    shutil.copytree(src, dst)
    subprocess.run([sys.executable, os.path.join(dst, "gen_golden.py")], check=True, cwd=dst, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    total, moved = 0, {}
    for f in sorted(glob.glob(os.path.join(src, "*.npz"))):
        a, b = np.load(f, allow_pickle=True), np.load(os.path.join(dst, os.path.basename(f)), allow_pickle=True)
        d = [k for k in a.files if k != "cases" and (k not in b.files or not np.array_equal(a[k], b[k]))]
        total += len(a.files)
        if d:
            moved[os.path.basename(f)] = d
    print(f"{total} arrays in {len(glob.glob(os.path.join(src, '*.npz')))} files; files that regenerate differently on this host: {len(moved)}")
    for f, d in moved.items():
        print(f"  {f}: {len(d)} arrays, e.g. {d[:3]}")
    shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
