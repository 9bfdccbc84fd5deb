import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
sc = scenes.cornell_box()
r = lprt_amd.Renderer(64, 64, abi.F32); r.load_scene(sc)
rays = np.zeros(1000, dtype=abi.RAY_DTYPE); rays["source"] = (0, 0, 3.5); rays["direction"] = (0, 0, -1); rays["max_distance"] = 1e5; rays["skip_vao"] = 0xffffffff
print("trace ok", r.trace(rays)["hit"].sum(), flush=True)
fs = abi.default_settings(spp=1)
try:
    r.render(fs); r.synchronize()
    print("render ok", flush=True)
    print(r.read("G_OBJID")[::8, ::8, 0])
except Exception as e:
    print("ERR", e, flush=True)
