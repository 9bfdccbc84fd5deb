"""Run N BLAS builds of the atrium (for rocprofv3 --kernel-trace: per-launch durations of one build)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lprt_amd
from lprt_amd import abi, scenes
sc = scenes.SCENES["sponza_like"]()
with lprt_amd.Renderer(64, 64, abi.F16) as r:
    r.load_scene(sc)
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
        r.build_blas()
    r.synchronize()
