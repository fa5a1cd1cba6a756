"""Import shim: the product package lives in the directory ``spatial-clip_amd/`` (a hyphen is not a valid
Python identifier), this module makes it importable as ``spatial_clip_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "spatial-clip_amd")
_spec = importlib.util.spec_from_file_location(
    "spatial_clip_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["spatial_clip_amd"] = _mod
_spec.loader.exec_module(_mod)
