"""rel-L2 of every read-back buffer, GPU vs oracle: python tools/gpu_buffer_diff.py scene W H f32|f16 [frames]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
from oracle.binding import Oracle
from conftest import rel_l2
name, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
prec = abi.F16 if sys.argv[4] == "f16" else abi.F32
frames = int(sys.argv[5]) if len(sys.argv) > 5 else 1
sc = scenes.SCENES[name]()
o = Oracle(W, H, prec); o.load_scene(sc)
r = lprt_amd.Renderer(W, H, prec); r.load_scene(sc)
for _ in range(frames):
    fs = abi.default_settings(spp=1)
    o.set_scene(sc); o.render_frame(fs); r.set_scene(sc); r.render(fs)
for n in abi.BUF:
    if n == "GI_DIRECTION":
        continue
    a, b = r.read(n), o.read(n)
    mism = (np.abs(np.nan_to_num(a) - np.nan_to_num(b)).sum(axis=-1) > 1e-3 * (1 + np.abs(np.nan_to_num(b)).sum(axis=-1))).mean()
    print(f"{n:18s} rel_l2={rel_l2(a, b):.3e}  pixels off: {mism:.4f}")
