"""Imports the package directory `icra20-hand-object-pose_amd/` (not a valid Python identifier) under the
module name `hop_amd`.  Used by tests/, bench.py and __graft_entry__.py."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "icra20-hand-object-pose_amd")


def load():
    if "hop_amd" in sys.modules:
        return sys.modules["hop_amd"]
    spec = importlib.util.spec_from_file_location(
        "hop_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["hop_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
