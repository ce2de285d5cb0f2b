"""Import shim: the package directory is `nonlinearsolve.jl_amd/` (a dot is not importable), so this module
loads it under the name `nonlinearsolve_jl_amd`."""
import importlib.util
import os
import sys

_d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nonlinearsolve.jl_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_d, "__init__.py"),
                                               submodule_search_locations=[_d])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
