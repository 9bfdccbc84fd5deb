import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
from oracle.binding import Oracle
import replay
from test_gpu_trace import random_rays
for name in ("cornell", "small_test"):
    sc = scenes.SCENES[name]()
    rays = random_rays(sc, 60000, 12, 0.1)
    o = Oracle(16, 16, abi.F16); o.load_scene(sc)
    ho, hoa = o.trace(rays, False), o.trace(rays, True)
    for slab in (0, 1):
        with lprt_amd.Renderer(16, 16, abi.F16) as r:
            r.set_option("reference_slab", slab)
            r.load_scene(sc)
            hg, hga = r.trace(rays, False), r.trace(rays, True)
        d = replay.closest_differs(hg, ho)
        j = replay.adjudicate(o, rays[d], hg[d], ho[d])
        gn, on = j["gpu_nearest"], j["oracle_nearest"]
        print(name, "slab", slab, "hitflag", int((hg["hit"] != ho["hit"]).sum()), "disputed", int(d.sum()), "both", int((gn & on).sum()), "only_gpu", int((gn & ~on).sum()),
              "only_oracle", int((~gn & on).sum()), "neither", int((~gn & ~on).sum()), "anyhit differ", int((hga["hit"] != hoa["hit"]).sum()), flush=True)
