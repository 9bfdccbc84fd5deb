"""Create / use / destroy textured contexts repeatedly (host-side lifetime check)."""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
from oracle.binding import Oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
sc = scenes.textured_scene()
sc2 = scenes.textured_scene(); sc2.skybox = None
for it in range(n):
    o = Oracle(96, 54, abi.F32); o.load_scene(sc); o.render_frame(abi.default_settings(spp=1, svgf=0))
    r = lprt_amd.Renderer(96, 54, abi.F32); r.load_scene(sc); r.render(abi.default_settings(spp=1, svgf=0))
    a = r.read("G_COLOR"); b = r.read("INTENSITY0"); c = o.read("INTENSITY0")
    r2 = lprt_amd.Renderer(96, 54, abi.F32); r2.load_scene(sc2); r2.render(abi.default_settings(spp=1, svgf=0)); d = r2.read("INTENSITY0")
    r3 = lprt_amd.Renderer(8, 8, abi.F32)
    try:
        bad = scenes.textured_scene(); bad.textures = bad.textures[:1]
        r3.set_rdresource(bad)
    except lprt_amd.LprtError:
        pass
    del r, r2, r3, o
    gc.collect()
    print("iteration", it, "ok", flush=True)
