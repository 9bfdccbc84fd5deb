import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
from oracle.binding import Oracle
sys.path.insert(0, "tests")
from conftest import rel_l2
sc = scenes.textured_scene()
if len(sys.argv) > 1 and sys.argv[1] == "nosky":
    sc.skybox = None
W, H = 192, 108
for prec in (abi.F32, abi.F16):
    o = Oracle(W, H, prec); o.load_scene(sc)
    r = lprt_amd.Renderer(W, H, prec); r.load_scene(sc)
    fs = abi.default_settings(spp=1)
    o.render_frame(fs); r.render(fs)
    print("prec", prec)
    for name in abi.BUF:
        try:
            a, b = r.read(name), o.read(name)
        except Exception as e:
            continue
        print(f"  {name:18s} {rel_l2(a, b):.3e}")
    a, b = r.read("INTENSITY1"), o.read("INTENSITY1")
    d = np.abs(a - b).sum(axis=-1)
    idx = np.argsort(d.reshape(-1))[::-1][:8]
    objid = r.read("G_OBJID")[..., 0]
    for i in idx:
        y, x = divmod(int(i), W)
        print("   ", x, y, "obj", objid[y, x], "gpu", a[y, x], "orc", b[y, x])
