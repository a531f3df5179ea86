"""Make the read-only reference (/root/reference) importable in the BUILD container.

Test/fixture infrastructure only (never imported by the product path, never used on the GPU box,
where /root/reference does not exist).  Nothing is copied from the reference: we only create three
tiny stand-ins for packages that are not installed here:

* a dist-info so ``importlib.metadata.version("nvidia-modelopt")`` resolves
  (reference: modelopt/__init__.py:20),
* ``omegaconf`` (imported at modelopt/torch/utils/robust_json.py:31),
* ``pulp`` (imported at modelopt/torch/opt/searcher.py:32, annotations only).
"""

import os
import sys
import tempfile

REFERENCE_ROOT = "/root/reference"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "modelopt"))


def install(shim_dir: str | None = None) -> str:
    """Create the shim (idempotent) and put shim + reference on sys.path. Returns the shim dir."""
    if not reference_available():
        raise RuntimeError("reference tree not present (expected only in the build container)")
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
    for p in (os.path.join(REFERENCE_ROOT, "tests"), REFERENCE_ROOT, shim):
        if p not in sys.path:
            sys.path.insert(0, p)
    return shim
