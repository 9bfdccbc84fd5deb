"""Odd frame sizes (tile edges of the LDS kernels), several flag combinations, two frames each, GPU vs oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
from oracle.binding import Oracle
from conftest import rel_l2
worst = 0.0
for name in ("cornell", "textured"):
    sc = scenes.SCENES[name]()
    for (W, H) in ((161, 97), (67, 131), (33, 9), (300, 5)):
        for prec in (abi.F32, abi.F16):
            for kw in (dict(spp=1), dict(spp=3, gi_on=False), dict(spp=2, svgf=False)):
                o = Oracle(W, H, prec); o.load_scene(sc)
                r = lprt_amd.Renderer(W, H, prec); r.load_scene(sc)
                for _ in range(2):
                    fs = abi.default_settings(**kw)
                    o.set_scene(sc); o.render_frame(fs); r.set_scene(sc); r.render(fs)
                e = rel_l2(r.read("OUTPUT"), o.read("OUTPUT"))
                tol = 1e-3 if prec == abi.F32 else 5e-2
                flag = "" if e < tol else "  <-- FAIL"
                worst = max(worst, e / tol)
                print(f"{name:9s} {W}x{H} prec={prec} {kw} rel_l2={e:.2e}{flag}", flush=True)
                r.close(); o.close()
print("worst error / tolerance:", worst)
sys.exit(0 if worst < 1 else 1)
