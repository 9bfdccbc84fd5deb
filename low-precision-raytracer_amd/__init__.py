"""MI355X-native low-precision path tracer: host-side harness over the C-ABI in include/lprt.h.

Only the hot path of ioissss/Low-precision-raytracer (rt/trace + rt/rtrt) lives here; see DESIGN.md.
The HIP library (csrc/liblprt_hip.so) is the product; importing this package never falls back to a CPU
implementation — `lprt_amd.Renderer` raises if the library or a GPU is missing.
"""
from . import abi, scenes  # noqa: F401

__all__ = ["abi", "scenes"]
__version__ = "0.1.0"


def __getattr__(name):  # lazy: keeps `import lprt_amd` usable for CPU-only host logic (scenes, abi)
    import importlib
    if name in ("Renderer", "load_library", "LprtError", "library_path"):
        return getattr(importlib.import_module(__name__ + ".renderer"), name)
    if name in ("dist", "renderer", "gltf"):
        return importlib.import_module(__name__ + "." + name)
    raise AttributeError(name)
