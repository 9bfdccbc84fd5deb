"""Find the GI rays of one seed that walk (nearly) the whole BLAS: reads back the GI commands, re-traces them in chunks
through lprt_trace_rays with counters and prints the worst chunks / rays."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
name, k = sys.argv[1], int(sys.argv[2])
sc = scenes.SCENES[name]()
W, H = 1920, 1080
r = lprt_amd.Renderer(W, H, abi.F32)
r.load_scene(sc)
a = 3 * k + 1
r.render(abi.default_settings(spp=1, seeds=[a, a + 1, a + 2]))
gd = r.read("GI_DIRECTION").reshape(-1, 4)
pos = r.read("G_POSITION").reshape(-1, 4)
valid = gd[:, 3] == 1
print("gi rays", valid.sum())
d = gd[valid, :3]; o = pos[valid, :3]
zeros = (d == 0).sum(axis=1)
small = (np.abs(d) < 1e-30).sum(axis=1)
print("rays with exact-zero components:", np.bincount(zeros, minlength=4), "components < 1e-30:", np.bincount(small, minlength=4))
print("non-finite dirs:", (~np.isfinite(d)).any(axis=1).sum(), "non-unit:", (np.abs(np.linalg.norm(d, axis=1) - 1) > 1e-3).sum())
rays = np.zeros(len(d), dtype=abi.RAY_DTYPE)
rays["source"], rays["direction"] = o, d
rays["min_distance"], rays["max_distance"] = 1e-4, 1e5
rays["skip_vao"] = 0xFFFFFFFF
r.enable_counters(True)
chunk = 1 << 14
worst = []
for i in range(0, len(rays), chunk):
    r.reset_counters()
    r.trace(rays[i:i + chunk])
    c = r.counters()
    worst.append((c["tri_tests"], i))
worst.sort(reverse=True)
print("worst chunks (tri tests, first ray):", worst[:5])
t, i0 = worst[0]
sub = rays[i0:i0 + chunk]
best = (0, None)
for j in range(0, chunk, 256):
    r.reset_counters(); r.trace(sub[j:j + 256]); c = r.counters()
    if c["tri_tests"] > best[0]:
        best = (c["tri_tests"], j)
print("worst 256-group:", best)
grp = sub[best[1]:best[1] + 256]
for j in range(len(grp)):
    r.reset_counters(); r.trace(grp[j:j + 1]); c = r.counters()
    if c["tri_tests"] > 1000:
        print("ray", i0 + best[1] + j, "o", grp[j]["source"], "d", grp[j]["direction"], {k_: v for k_, v in c.items()})
