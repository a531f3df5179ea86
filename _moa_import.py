"""Import helper: the package directory is named `model-optimizer_amd` (hyphen, as the repo contract
asks), which is not a legal Python identifier -- load it under the module name `model_optimizer_amd`."""

import importlib.util
import os
import sys

_NAME = "model_optimizer_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "model-optimizer_amd")
    spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(root, "__init__.py"), submodule_search_locations=[root])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
