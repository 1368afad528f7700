"""Import shim: exposes the package directory ``dftk.jl_amd/`` as module ``dftk_jl_amd``."""
import importlib.util
import os
import sys

_here = os.path.dirname(os.path.abspath(__file__))
_pkg = os.path.join(_here, "dftk.jl_amd")
_spec = importlib.util.spec_from_file_location("dftk_jl_amd", os.path.join(_pkg, "__init__.py"),
                                               submodule_search_locations=[_pkg])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dftk_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
