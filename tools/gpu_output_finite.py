"""Exploration: where the atrium's OUTPUT plane is non-finite or has alpha != 1, on the HIP path and in the oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
from oracle.binding import Oracle

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (480, 270)
with_oracle = (len(sys.argv) <= 3) or sys.argv[3] != "gpu"
sc = scenes.SCENES["sponza_like"]()
for prec in ((abi.F16,) if os.environ.get('ONLY_F16') else (abi.F32, abi.F16)):
    r = lprt_amd.Renderer(W, H, prec); r.load_scene(sc)
    o = None
    if with_oracle:
        o = Oracle(W, H, prec); o.load_scene(sc)
    for f in range(int(os.environ.get('FRAMES', '4'))):
        fs = abi.default_settings(spp=8); fs.rebuild_blas = 1
        r.set_scene(sc); r.render(fs)
        a = r.read("OUTPUT")
        line = f"prec={prec} frame={f}: gpu nonfinite px {int((~np.isfinite(a)).any(axis=-1).sum())} alpha!=1 px {int((a[..., 3] != 1).sum())}"
        for name in ("RAW_COLOR", "RAW_GI_COLORED", "MUL_GI_COLORED", "COLOR_INPROGRESS", "G_COLOR"):
            x = r.read(name); line += f" | {name} nf {int((~np.isfinite(x)).any(axis=-1).sum())}"
        if o is not None:
            o.set_scene(sc); o.render_frame(fs)
            b = o.read("OUTPUT")
            nfa, nfb = ~np.isfinite(a).all(axis=-1), ~np.isfinite(b).all(axis=-1)
            line += f" || oracle nonfinite px {int(nfb.sum())} alpha!=1 px {int((b[..., 3] != 1).sum())} same set {bool((nfa == nfb).all())}"
            ci_a, ci_b = r.read("COLOR_INPROGRESS"), o.read("COLOR_INPROGRESS")
            line += f" | negative COLOR_INPROGRESS px gpu {int((ci_a[..., :3] < 0).any(axis=-1).sum())} oracle {int((ci_b[..., :3] < 0).any(axis=-1).sum())} min {ci_a[..., :3].min():.3e} / {ci_b[..., :3].min():.3e}"
        print(line, flush=True)
    r.close()
    if o is not None:
        o.close()
