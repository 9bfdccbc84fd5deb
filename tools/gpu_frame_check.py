"""GPU bring-up: full-frame parity of every read-back buffer against the oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
from oracle.binding import Oracle


def rel_l2(a, b):
    a = np.nan_to_num(a.astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
    b = np.nan_to_num(b.astype(np.float64), nan=0.0, posinf=0.0, neginf=0.0)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / den) if den > 0 else float(np.linalg.norm(a - b))


def check(scene_name, W, H, prec, spp=1, frames=1, svgf=True):
    sc = scenes.SCENES[scene_name]()
    o = Oracle(W, H, prec); o.load_scene(sc)
    r = lprt_amd.Renderer(W, H, prec); r.load_scene(sc)
    for f in range(frames):
        fs = abi.default_settings(spp=spp, svgf=svgf)
        t0 = time.time(); o.set_scene(sc); o.render_frame(fs); t_o = time.time() - t0
        t0 = time.time(); r.set_scene(sc); r.render(fs); r.synchronize(); t_g = time.time() - t0
        print(f"[{scene_name} {W}x{H} prec={prec} spp={spp} frame={f}] oracle {t_o:.2f}s gpu {t_g*1e3:.1f}ms")
        for name in ["G_OBJID", "G_OFFSET", "G_COLOR", "G_POSITION", "G_NORMAL_DEPTH", "G_TANGENT", "ALBEDO", "INTENSITY0",
                     "INTENSITY1", "RAW_COLOR", "RAW_GI_COLORED", "RAW_GI_WHITE", "MUL_GI_COLORED", "MUL_GI_WHITE",
                     "COLOR_INPROGRESS", "OUTPUT"]:
            a, b = r.read(name), o.read(name)
            nbad = int((np.abs(a - b) > 1e-3 * (1 + np.abs(b))).any(axis=-1).sum())
            print(f"    {name:18s} rel_l2={rel_l2(a, b):.3e}  max|d|={float(np.nanmax(np.abs(np.nan_to_num(a)-np.nan_to_num(b)))):.3e}  px_off={nbad}"
                  f"  nonfinite gpu/orc={int((~np.isfinite(a)).sum())}/{int((~np.isfinite(b)).sum())}")
    return r, o


if __name__ == "__main__":
    which = sys.argv[1:] or ["cornell32", "cornell16", "small32", "multi"]
    if "cornell32" in which:
        check("cornell", 256, 256, abi.F32)
    if "cornell16" in which:
        check("cornell", 256, 256, abi.F16)
    if "small32" in which:
        check("small_test", 200, 120, abi.F32)
        check("small_test", 200, 120, abi.F16)
    if "multi" in which:
        check("cornell", 128, 128, abi.F32, spp=2, frames=3)
        check("cornell", 128, 128, abi.F16, spp=2, frames=3)
