"""N frames of the atrium with every BLAS rebuilt inside each frame (configs[3] at a small size), for a rocprofv3 --hip-trace
run: the API summary shows which HIP calls the frames make -- in particular that no hipStreamSynchronize /
hipDeviceSynchronize sits between lprt_set_scene and the end of lprt_render_frame (the count of synchronisations does not
grow with the number of frames).  usage: gpu_rebuild_frames.py <frames>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lprt_amd
from lprt_amd import abi, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sc = scenes.SCENES["sponza_like"]()
with lprt_amd.Renderer(640, 360, abi.F16) as r:
    r.load_scene(sc)
    prepared = r.prepare_scene(sc)
    fs = abi.default_settings(spp=2)
    fs.rebuild_blas = 1
    r.set_scene(prepared=prepared)
    r.render(fs)          # first frame: allocations, layout, plan
    r.synchronize()
    print("MARK frames begin", flush=True)
    for _ in range(n):
        r.set_scene(prepared=prepared)
        r.render(fs)
    r.synchronize()
    print("MARK frames end", flush=True)
