"""Import the package in ./diffusion-model_amd under the importable alias ``dm_amd``."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "diffusion-model_amd")


def load():
    if "dm_amd" in sys.modules:
        return sys.modules["dm_amd"]
    spec = importlib.util.spec_from_file_location("dm_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["dm_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


load()
