"""Import shim: ``import refnerf_pl_amd`` loads the package that lives in the
``refnerf-pl_amd/`` directory (a hyphen is not importable as a module name)."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
_pkg_dir = _os.path.join(_here, "refnerf-pl_amd")
_spec = _ilu.spec_from_file_location(
    "refnerf_pl_amd", _os.path.join(_pkg_dir, "__init__.py"),
    submodule_search_locations=[_pkg_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules["refnerf_pl_amd"] = _mod
_spec.loader.exec_module(_mod)
