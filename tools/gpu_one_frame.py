"""One 1080p frame (spp from argv) per precision of a scene: a small target for rocprofv3 --pmc runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lprt_amd
from lprt_amd import abi, scenes
name = sys.argv[1] if len(sys.argv) > 1 else "cornell"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sc = scenes.SCENES[name]()
for prec in (abi.F32, abi.F16):
    r = lprt_amd.Renderer(1920, 1080, prec)
    r.load_scene(sc)
    for _ in range(2):
        r.set_scene(sc)
        r.render(abi.default_settings(spp=spp))
    r.synchronize()
    r.close()
