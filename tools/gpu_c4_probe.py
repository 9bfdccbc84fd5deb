"""Exploration: atrium (BASELINE configs[3] scene) frames against the oracle with ray replay (tests/replay.py).
Prints, per precision and frame: the number of pixels where some ray found something else, and every radiance plane's
relative L2 / worst per-pixel error over all pixels and over the pixels no such ray touches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
from oracle.binding import Oracle
import replay
from conftest import rel_l2

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 144)
spp, frames = 2, 2
scene = sys.argv[3] if len(sys.argv) > 3 else "sponza_like"
sc = scenes.SCENES[scene]()
PRE = ["INTENSITY0", "INTENSITY1", "RAW_COLOR", "RAW_GI_COLORED", "RAW_GI_WHITE"]
POST = ["MUL_GI_COLORED", "MUL_GI_WHITE", "COLOR_INPROGRESS", "OUTPUT"]
for prec in (abi.F32, abi.F16):
    o = Oracle(W, H, prec); o.load_scene(sc)
    r = lprt_amd.Renderer(W, H, prec); r.load_scene(sc)
    probe = lprt_amd.Renderer(W, H, prec); probe.load_scene(sc)
    oprobe = Oracle(W, H, prec); oprobe.load_scene(sc)
    ever = np.zeros((H, W), dtype=bool)
    for f in range(frames):
        fs = abi.default_settings(spp=spp); fs.rebuild_blas = 1
        o.set_scene(sc); o.render_frame(fs); r.set_scene(sc); r.render(fs)
        c = replay.classify(r, o, sc, prec, probe, oprobe, f, spp, rebuild_blas=1)
        ever |= c["any"]
        n = lambda k: [int(g.sum()) for g in c[k]]
        print(f"prec={prec} frame={f}: primary {int(c['primary'].sum())}  shadow0 {int(c['shadow0'].sum())}  gi {n('gi')} (tie {n('gi_tie')}, other ray {n('gi_other_ray')}, "
              f"cull {n('gi_cull')})  shadow1 {n('shadow1')}  any {int(c['any'].sum())} of {W*H} px; {c['n_rays']} rays replayed", flush=True)
        floor = 1e-3
        for name in PRE + POST:
            a, b = r.read(name), o.read(name)
            excl = c["last_sample"] if name in ("INTENSITY0", "INTENSITY1") else c["any"]
            if name in POST:
                excl = replay.dilate(ever, 67)
            keep = ~excl
            e = replay.pixel_error(a, b, floor)
            print(f"   {name:18s} rel_l2 all {rel_l2(a, b):.3e}  kept({int(keep.sum())}) {rel_l2(a[keep], b[keep]):.3e}  worst px err all {e.max():.3e} kept {e[keep].max() if keep.any() else 0:.3e}"
                  f"  px>1e-2 kept {int((e[keep] > 1e-2).sum())}", flush=True)
    r.close(); o.close(); probe.close(); oprobe.close()
