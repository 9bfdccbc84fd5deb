"""Import shim: the package directory is `low-precision-raytracer_amd/` (hyphens are not importable),
so `import lprt_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "low-precision-raytracer_amd")
_spec = importlib.util.spec_from_file_location("lprt_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lprt_amd"] = _mod
_spec.loader.exec_module(_mod)
