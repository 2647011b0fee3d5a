"""Import shim: the package directory is named `tulip.jl_amd` (not a valid Python identifier),
so `import tulip_jl_amd` loads it from that directory."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tulip.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "tulip_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tulip_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
