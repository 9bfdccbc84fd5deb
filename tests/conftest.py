import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the suites need the product library (symbol / layout checks on CPU, everything on GPU): build it when a fresh
    # checkout has not been through __graft_entry__.build() yet (hipcc cross-compiles without a GPU)
    lib = os.path.join(ROOT, "low-precision-raytracer_amd", "csrc", "liblprt_hip.so")
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(["make", "-C", os.path.dirname(lib), "-j8"], check=True, capture_output=True)
    _install_guarded_out_arrays()


def _install_guarded_out_arrays():
    """Every out-array the bindings hand to native code (product library and oracle) ends at a PROT_NONE page."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import guarded
    import lprt_amd  # noqa: F401
    from lprt_amd import renderer
    from oracle import binding
    renderer.alloc_out = guarded.empty
    binding.alloc_out = guarded.empty


@pytest.fixture(autouse=True)
def _deterministic_native_lifetimes():
    """No context outlives the test that made it: lprt_destroy / orc_destroy run HERE, on the main thread, between tests,
    with the HIP runtime, PyTorch and every stream still alive -- never from a garbage-collection pass at some
    arbitrary allocation, and never at interpreter shutdown."""
    yield
    mods = sys.modules
    if "lprt_amd.renderer" in mods:
        mods["lprt_amd.renderer"].close_all()
    if "oracle.binding" in mods:
        mods["oracle.binding"].close_all()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def cornell():
    import lprt_amd  # noqa: F401
    from lprt_amd import scenes
    return scenes.cornell_box()


@pytest.fixture(scope="session")
def small_scene():
    import lprt_amd  # noqa: F401
    from lprt_amd import scenes
    return scenes.small_test_scene()


@pytest.fixture(scope="session")
def sponza():
    import lprt_amd  # noqa: F401
    from lprt_amd import scenes
    return scenes.sponza_like()


def rel_l2(a, b):
    import numpy as np
    a = np.nan_to_num(np.asarray(a, dtype=np.float64), nan=0.0, posinf=0.0, neginf=0.0)
    b = np.nan_to_num(np.asarray(b, dtype=np.float64), nan=0.0, posinf=0.0, neginf=0.0)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / den) if den > 0 else float(np.linalg.norm(a - b))
