"""GPU-side sanity run: BVH build + ray batch parity against the oracle (used during bring-up)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import lprt_amd
from lprt_amd import abi, scenes
from oracle.binding import Oracle


def random_rays(sc, n, seed, rng_box=2.5):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=abi.RAY_DTYPE)
    lo = sc.aabbs[:, :3].min(axis=0) - 0.5
    hi = sc.aabbs[:, 3:].max(axis=0) + 0.5
    rays["source"] = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays["direction"] = d.astype(np.float32)
    rays["min_distance"] = 1e-4
    rays["max_distance"] = 1e5
    rays["skip_vao"] = 0xFFFFFFFF
    return rays


def compare(name, sc, prec, n=200000):
    W = H = 64
    o = Oracle(W, H, prec)
    o.load_scene(sc)
    r = lprt_amd.Renderer(W, H, prec)
    t0 = time.time()
    r.load_scene(sc)
    r.synchronize()
    print(f"[{name} prec={prec}] upload+build {time.time()-t0:.3f}s  tris={sc.n_tris}")
    rays = random_rays(sc, n, 123)
    if prec == abi.F16:
        rays["min_distance"] = 0.1
    for any_hit in (False, True):
        ho = o.trace(rays, any_hit)
        r.enable_counters(True); r.reset_counters()
        t0 = time.time()
        hg = r.trace(rays, any_hit)
        dt = time.time() - t0
        hit_mismatch = int((ho["hit"] != hg["hit"]).sum())
        both = (ho["hit"] == 1) & (hg["hit"] == 1)
        id_mismatch = int(((ho["inscene_objid"] != hg["inscene_objid"]) | (ho["inobject_offset"] != hg["inobject_offset"]))[both].sum()) if not any_hit else 0
        same = both & (ho["inscene_objid"] == hg["inscene_objid"]) & (ho["inobject_offset"] == hg["inobject_offset"])
        dt_max = float(np.abs(ho["t"][same] - hg["t"][same]).max()) if same.any() and not any_hit else 0.0
        db_max = float(np.abs(ho["barycentric"][same] - hg["barycentric"][same]).max()) if same.any() and not any_hit else 0.0
        print(f"   any={any_hit}: hits oracle={int(ho['hit'].sum())} gpu={int(hg['hit'].sum())} hit_mismatch={hit_mismatch} "
              f"id_mismatch={id_mismatch} max|dt|={dt_max:.3g} max|dbary|={db_max:.3g}  ({dt*1e3:.1f} ms incl. copies)")
        print("   gpu counters", r.counters())
    o.close(); r.close()


if __name__ == "__main__":
    print(lprt_amd.load_library().lprt_version().decode())
    for prec in (abi.F32, abi.F16):
        compare("cornell", scenes.cornell_box(), prec)
        compare("small", scenes.small_test_scene(), prec)
    sp = scenes.sponza_like()
    for prec in (abi.F32, abi.F16):
        compare("sponza_like", sp, prec, n=100000)
