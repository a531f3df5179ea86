"""Make the read-only reference importable: from /root/reference in the BUILD container, or -- on the GPU box, where
that checkout does not exist -- from the gitignored archive `tools/stage_reference.sh` packs under oracle/_ref/
(unpacked into a temp dir at first use; it travels with `gpurun` like the built .so files).

Test/fixture infrastructure only: never imported by the product path (model-optimizer_amd/, include/) and never
inside bench.py's timed region.  Nothing of the reference is in the tree: we only create three tiny stand-ins for
packages that are not installed here:

* a dist-info so ``importlib.metadata.version("nvidia-modelopt")`` resolves
  (reference: modelopt/__init__.py:20),
* ``omegaconf`` (imported at modelopt/torch/utils/robust_json.py:31),
* ``pulp`` (imported at modelopt/torch/opt/searcher.py:32, annotations only).
"""

import os
import sys
import tempfile

REFERENCE_ROOT = "/root/reference"
_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
STAGED_ARCHIVE = os.path.join(_REPO, "oracle", "_ref", "reference_modelopt.tgz")


def reference_source() -> str | None:
    """"checkout" (the build container), "staged" (the archive of tools/stage_reference.sh) or None."""
    if os.path.isdir(os.path.join(REFERENCE_ROOT, "modelopt")) and not os.environ.get("MOQ_REF_FORCE_STAGED"):
        return "checkout"
    if os.path.isfile(STAGED_ARCHIVE):
        return "staged"
    return None


def reference_available() -> bool:
    return reference_source() is not None


def reference_root() -> str:
    """Directory holding `modelopt/` (and, for the checkout, `tests/`).  The staged archive is unpacked once per
    archive content into <tmp>/moq_ref_stage_<size>_<mtime>/."""
    src = reference_source()
    if src == "checkout":
        return REFERENCE_ROOT
    if src is None:
        raise RuntimeError("no reference: neither /root/reference nor oracle/_ref/reference_modelopt.tgz "
                           "(run tools/stage_reference.sh in the build container)")
    import tarfile

    st = os.stat(STAGED_ARCHIVE)
    dest = os.path.join(tempfile.gettempdir(), f"moq_ref_stage_{st.st_size}_{int(st.st_mtime)}")
    if not os.path.isdir(os.path.join(dest, "modelopt")):
        part = f"{dest}.{os.getpid()}.part"
        with tarfile.open(STAGED_ARCHIVE) as tf:
            tf.extractall(part)
        try:
            os.rename(part, dest)
        except OSError:  # another process (xdist worker) finished first
            import shutil

            shutil.rmtree(part, ignore_errors=True)
    return dest


def install(shim_dir: str | None = None) -> str:
    """Create the shim (idempotent) and put shim + reference on sys.path. Returns the shim dir."""
    root = reference_root()
    shim = shim_dir or os.path.join(tempfile.gettempdir(), "moq_ref_shim")
    os.makedirs(os.path.join(shim, "nvidia_modelopt-0.0.0.dist-info"), exist_ok=True)
    with open(os.path.join(shim, "nvidia_modelopt-0.0.0.dist-info", "METADATA"), "w") as f:
        f.write("Metadata-Version: 2.1\nName: nvidia-modelopt\nVersion: 0.0.0\n")
    os.makedirs(os.path.join(shim, "omegaconf"), exist_ok=True)
    with open(os.path.join(shim, "omegaconf", "__init__.py"), "w") as f:
        f.write(
            "class DictConfig(dict):\n    pass\n\n"
            "class ListConfig(list):\n    pass\n\n"
            "class OmegaConf:\n"
            "    @staticmethod\n    def to_container(x, **k):\n        return x\n"
            "    @staticmethod\n    def create(x=None, **k):\n        return x\n"
        )
    os.makedirs(os.path.join(shim, "pulp"), exist_ok=True)
    with open(os.path.join(shim, "pulp", "__init__.py"), "w") as f:
        f.write(
            "def __getattr__(n):\n"
            "    if n.startswith('__'):\n        raise AttributeError(n)\n"
            "    return type(n, (), {})\n"
        )
    for p in (os.path.join(root, "tests"), root, shim):
        if p not in sys.path and (os.path.isdir(p) or p == shim):
            sys.path.insert(0, p)
    return shim
