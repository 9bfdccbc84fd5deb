import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lprt_amd
from lprt_amd import abi, scenes, dist
from oracle.binding import Oracle
sc = scenes.cornell_box()
W, H = 160, 300
o = Oracle(W, H, abi.F32); o.load_scene(sc)
full = lprt_amd.Renderer(W, H, abi.F32); full.load_scene(sc)
alts = {}
for we in (168, 200, 169, 128):
    r = lprt_amd.Renderer(W, H, abi.F32, row_begin=0, row_end=we); r.load_scene(sc); alts[we] = r
for k in range(2):
    fs = abi.default_settings(spp=1)
    o.set_scene(sc); o.render_frame(fs)
    full.set_scene(sc); full.render(fs)
    for r in alts.values():
        r.set_scene(sc); r.render(fs)
want = o.read("OUTPUT")
f = full.read("OUTPUT")
d = np.abs(f - want).reshape(H, -1).max(axis=1)
print("full vs oracle: rows with err>1e-4:", np.where(d > 1e-4)[0][:20], d.max())
for we, r in alts.items():
    g = r.read("OUTPUT")
    bad = np.where((g[:60] != f[:60]).reshape(60, -1).any(axis=1))[0]
    bad_all = np.where((g[:we-68] != f[:we-68]).reshape(we-68, -1).any(axis=1))[0]
    d = np.abs(g - want[:we]).reshape(we, -1).max(axis=1)
    print(f"window 0..{we}: rows != full (first {we-68} rows): {bad_all[:10]}; err vs oracle at those rows {d[bad_all[:5]]}, full err there {np.abs(f - want).reshape(H,-1).max(axis=1)[bad_all[:5]]}")
    for row in bad_all[:2]:
        cols = np.where((g[row] != f[row]).any(axis=1))[0]
        print("   row", row, "cols", cols[:10], "...", len(cols), "g", g[row, cols[0]], "f", f[row, cols[0]], "o", want[row, cols[0]])
