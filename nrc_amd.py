"""Registers `neural-radiance-caching_amd/` (not a valid identifier) as module `nrc_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "neural-radiance-caching_amd")
_spec = importlib.util.spec_from_file_location(
    "nrc_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nrc_amd"] = _mod
_spec.loader.exec_module(_mod)
