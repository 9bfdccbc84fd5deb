"""GPU parity (through the C-ABI): acceleration structures and cast_ray hit records against the CPU oracle and the
committed golden ray batch.  Integer results (hit flag, object id, triangle offset) must be identical; the only
admitted differences are exact-t ties between coplanar triangles, which depend on BVH traversal order
 and are counted and bounded."""
import os

import numpy as np
import pytest

import lprt_amd
from lprt_amd import abi, scenes
from oracle.binding import Oracle
import replay

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def random_rays(sc, n, seed, tmin):
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=abi.RAY_DTYPE)
    lo = sc.aabbs[:, :3].min(axis=0) - 0.5
    hi = sc.aabbs[:, 3:].max(axis=0) + 0.5
    rays["source"] = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3))
    rays["direction"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["min_distance"] = tmin
    rays["max_distance"] = 1e5
    rays["skip_vao"] = 0xFFFFFFFF
    # axis-parallel rays and rays starting exactly on geometry planes (slab tests with non-finite axes, Q3)
    rays["direction"][:64] = np.tile(np.eye(3, dtype=np.float32), (22, 1))[:64]
    rays["source"][64:96, 1] = -1.0
    return rays


def compare_hits(ho, hg, any_hit, max_tie_frac):
    assert np.array_equal(ho["hit"], hg["hit"]), f"{int((ho['hit'] != hg['hit']).sum())} hit flags differ"
    if any_hit:
        return 0
    both = ho["hit"] == 1
    same = both & (ho["inscene_objid"] == hg["inscene_objid"]) & (ho["inobject_offset"] == hg["inobject_offset"])
    ties = both & ~same
    # a differing triangle is only legal at a tie in t (coplanar triangles of two objects: t agrees to rounding)
    assert np.allclose(ho["t"][ties], hg["t"][ties], rtol=1e-5, atol=1e-6), "closest hits differ at different t"
    assert ties.sum() <= max_tie_frac * max(1, both.sum())
    assert np.array_equal(ho["t"][same], hg["t"][same])
    assert np.array_equal(ho["barycentric"][same], hg["barycentric"][same])
    return int(ties.sum())


@pytest.mark.parametrize("scene_name,n,builder", [("cornell", 60000, -1), ("small_test", 60000, -1), ("small_test", 60000, 2), ("sponza_like", 40000, 0),
                                                   ("sponza_like", 40000, 2)])
def test_fp32_hit_records_match_oracle(scene_name, n, builder):
    """builder: LPRT_OPT_BLAS_BUILDER (-1 the default choice, 0 median split, 2 surface-area heuristic): the hierarchy is a
    matter of speed, the hit records are the oracle's either way (exact-t ties excepted)."""
    sc = scenes.SCENES[scene_name]()
    o = Oracle(16, 16, abi.F32)
    o.load_scene(sc)
    r = lprt_amd.Renderer(16, 16, abi.F32)
    r.set_option("blas_builder", builder)
    r.load_scene(sc)
    rays = random_rays(sc, n, 11, 1e-4)
    for any_hit in (False, True):  # any-hit batches take the walk the frame kernels take: two levels per step on the atrium
        compare_hits(o.trace(rays, any_hit), r.trace(rays, any_hit), any_hit, 0.01)


@pytest.mark.parametrize("slab", [0, 1], ids=["f32-slab", "reference-slab"])
@pytest.mark.parametrize("scene_name", ["cornell", "small_test"])
def test_fp16_hit_records_match_oracle(scene_name, slab):
    """Half-precision hit records on the small scenes, 60 k rays, both slab modes, held to what was measured in round 3
    (tools/gpu_fp16_small_probe.py) instead of percentages:
      * random soup (`small_test`): identical to the oracle -- flags, triangles, t, barycentrics, visibility -- in both modes;
      * Cornell box, reference slab arithmetic: identical flags and visibility; ~50 rays report another triangle, every one an
        exact-t tie between coplanar triangles of two instances (both answers are nearest hits by exhaustive search);
      * Cornell box, default (f32) slab parameters: the same ties, plus at most a handful of rays where ONLY the GPU reports the
        nearest hit (the reference's half slab test culled the box) and never the reverse."""
    sc = scenes.SCENES[scene_name]()
    o = Oracle(16, 16, abi.F16)
    o.load_scene(sc)
    r = lprt_amd.Renderer(16, 16, abi.F16)
    r.set_option("reference_slab", slab)
    r.load_scene(sc)
    rays = random_rays(sc, 60000, 12, 0.1)
    ho, hg = o.trace(rays, False), r.trace(rays, False)
    hoa, hga = o.trace(rays, True), r.trace(rays, True)
    d = replay.closest_differs(hg, ho)
    same = ~d & (ho["hit"] == 1)
    assert np.array_equal(ho["t"][same], hg["t"][same]) and np.array_equal(ho["barycentric"][same], hg["barycentric"][same])
    if scene_name == "small_test":
        assert not d.any() and np.array_equal(hoa["hit"], hga["hit"])
        return
    j = replay.adjudicate(o, rays[d], hg[d], ho[d])
    only_gpu, only_oracle = j["gpu_nearest"] & ~j["oracle_nearest"], ~j["gpu_nearest"] & j["oracle_nearest"]
    assert int(d.sum()) <= 80 and not only_oracle.any() and (j["gpu_nearest"] | j["oracle_nearest"]).all()
    flags, vis = int((ho["hit"] != hg["hit"]).sum()), int((hoa["hit"] != hga["hit"]).sum())
    if slab:
        assert flags == 0 and vis == 0 and not only_gpu.any()
    else:
        assert flags <= 3 and vis <= 3 and int(only_gpu.sum()) <= 6


@pytest.mark.parametrize("slab", [0, 1], ids=["f32-slab", "reference-slab"])
def test_fp16_closest_hits_on_the_atrium_against_oracle_and_exhaustive_search(sponza, slab):
    """Half-precision closest hits on the 262 k-triangle atrium, id and t, against the oracle -- in both slab modes
    (LPRT_OPT_REFERENCE_SLAB: 0 = slab parameters in f32, the default; 1 = the reference's float16 arithmetic).

    In half precision the answer depends on the hierarchy: the reference's slab test culls boxes the ray does pass, and t has
    11 bits, so several triangles can sit at the "same" t.  The reference's own hierarchy is only defined up to the order
    std::nth_element leaves equal keys in (object_bvh.hpp:122-125), so the yardstick has two parts:
      * how often two LEGAL reference topologies disagree with each other (the oracle under tie orders 1 and 2);
      * the exhaustive search (the reference's triangle test against every triangle, no boxes): a traversal can only lose
        candidates, so the answer whose t equals the exhaustive t is a nearest hit.
    HIP vs oracle: disputed rays <= 1.25x the oracle-vs-oracle count; t / barycentrics bit-equal wherever the triangle
    agrees; on the disputed rays the GPU loses the nearest hit no more often than one reference topology loses it against
    another; undisputed rays (a sample) are nearest hits on both sides."""
    prec = abi.F16
    rays = random_rays(sponza, 40000, 13, 0.1)
    orc, ans = {}, {}
    for tie in (0, 1, 2):
        orc[tie] = Oracle(16, 16, prec)
        orc[tie].set_tie_order(tie)
        orc[tie].load_scene(sponza)
        ans[tie] = orc[tie].trace(rays, False)
    with lprt_amd.Renderer(16, 16, prec) as r:
        r.set_option("reference_slab", slab)
        r.load_scene(sponza)
        hg = r.trace(rays, False)
    ho = ans[0]
    baseline = max(int(replay.closest_differs(ans[a], ans[b]).sum()) for a, b in ((0, 1), (0, 2), (1, 2)))
    d = replay.closest_differs(hg, ho)
    assert int((hg["hit"] != ho["hit"]).sum()) <= 12
    assert int(d.sum()) <= 1.25 * baseline, (int(d.sum()), baseline)
    same = ~d & (ho["hit"] == 1)
    assert np.array_equal(hg["t"][same], ho["t"][same]) and np.array_equal(hg["barycentric"][same], ho["barycentric"][same])
    j = replay.adjudicate(orc[0], rays[d], hg[d], ho[d])
    gpu_lost = int((~j["gpu_nearest"] & j["oracle_nearest"]).sum())
    neither = int((~j["gpu_nearest"] & ~j["oracle_nearest"]).sum())
    lost_between_topologies = 0
    for a, b in ((1, 0), (2, 0), (0, 1), (0, 2)):
        dd = replay.closest_differs(ans[a], ans[b])
        jj = replay.adjudicate(orc[0], rays[dd], ans[a][dd], ans[b][dd])
        lost_between_topologies = max(lost_between_topologies, int((~jj["gpu_nearest"] & jj["oracle_nearest"]).sum()))
    assert gpu_lost <= 1.25 * lost_between_topologies and neither <= 2, (gpu_lost, lost_between_topologies, neither)
    idx = np.flatnonzero(~d)[:400]
    j = replay.adjudicate(orc[0], rays[idx], hg[idx], ho[idx])
    assert j["gpu_nearest"].all() and j["oracle_nearest"].all()


def test_fp32_closest_hits_on_the_atrium_are_the_exhaustive_search_s(sponza):
    """Single precision: nothing is lost to a box test -- a sample of closest hits equals the exhaustive search in t."""
    rays = random_rays(sponza, 400, 17, 1e-4)
    with Oracle(16, 16, abi.F32) as o, lprt_amd.Renderer(16, 16, abi.F32) as r:
        o.load_scene(sponza)
        r.load_scene(sponza)
        hg = r.trace(rays, False)
        j = replay.adjudicate(o, rays, hg, o.trace(rays, False))
        assert j["gpu_nearest"].all() and j["oracle_nearest"].all()


@pytest.mark.parametrize("prec", [abi.F32, abi.F16])
def test_two_level_walk_reports_the_hits_of_the_one_level_walk(sponza, prec):
    """The walk the shadow rays of deep BLAS trees take by default (trace.hpp, WIDE: grandchild boxes tested straight from
    the two adjacent child records), at hit-record level on the 262 k-triangle atrium: LPRT_OPT_WIDE_WALK = 1 sends
    lprt_trace_rays through it for any-hit AND closest-hit batches.  Against the one-level walk (HIP vs HIP): identical
    visibility, identical closest hits.  Against the oracle: what the one-level walk is held to."""
    rays = random_rays(sponza, 40000, 13, 1e-4 if prec == abi.F32 else 0.1)
    o = Oracle(16, 16, prec)
    o.load_scene(sponza)
    hits = {}
    for wide in (0, 1):
        with lprt_amd.Renderer(16, 16, prec) as r:
            r.set_option("blas_builder", 0)  # the two-level walk exists for the perfect trees of the median builder
            r.set_option("wide_walk", wide)
            r.load_scene(sponza)
            r.enable_counters(True)
            hits[wide] = (r.trace(rays, False), r.trace(rays, True))
            assert r.counters()["rays"] == 2 * len(rays)
    for any_hit in (0, 1):
        a, b = hits[0][any_hit], hits[1][any_hit]
        assert np.array_equal(a["hit"], b["hit"])
        if not any_hit:
            same = (a["inscene_objid"] == b["inscene_objid"]) & (a["inobject_offset"] == b["inobject_offset"])
            assert (~same).sum() <= 0.002 * len(rays)           # traversal order differs: ties only
            if prec == abi.F32:
                assert np.array_equal(a["t"][~same], b["t"][~same])
            else:
                # Half precision: the running closest t is kept in half ("the same t" = within a few half ulps), and the
                # half slab test loses some boxes a ray does pass -- the two-level walk never tests the
                # child boxes, so now and then it reaches a closer triangle the one-level walk culled.  Bounded, never
                # farther than a tie's worth.
                assert (b["t"][~same] <= a["t"][~same] * (1 + 4e-3)).all()
            assert np.array_equal(a["t"][same], b["t"][same]) and np.array_equal(a["barycentric"][same], b["barycentric"][same])
        ho = o.trace(rays, bool(any_hit))
        if prec == abi.F32:
            compare_hits(ho, b, bool(any_hit), 0.01)
        else:  # (the half-precision statement proper: test_fp16_closest_hits_on_the_atrium_against_oracle_and_exhaustive_search)
            assert (ho["hit"] != b["hit"]).mean() < 2e-3


def test_golden_ray_batch(small_scene):
    g = np.load(os.path.join(GOLD, "rays_small.npz"))
    for prec, tag, tol in ((abi.F32, "f32", 0.0), (abi.F16, "f16", 1e-3)):
        r = lprt_amd.Renderer(16, 16, prec)
        r.load_scene(small_scene)
        rays = g["rays"].copy()
        if prec == abi.F16:
            rays["min_distance"] = 0.1
        h, w = r.trace(rays, False), g[f"closest_{tag}"]
        assert (h["hit"] != w["hit"]).mean() <= tol
        ok = (h["hit"] == 1) & (w["hit"] == 1) & (h["inobject_offset"] == w["inobject_offset"]) & (h["inscene_objid"] == w["inscene_objid"])
        assert ok.sum() >= (1 - 10 * tol) * (w["hit"] == 1).sum() - 2
        assert np.array_equal(h["t"][ok], w["t"][ok])
        assert (r.trace(rays, True)["hit"].astype(np.uint8) != g[f"any_{tag}"]).mean() <= tol


def test_skip_geometry_and_distance_window(cornell):
    """SkipGeometry and (min, max) window semantics of cast_ray (cuda.hpp:446, 263)."""
    r = lprt_amd.Renderer(16, 16, abi.F32)
    r.load_scene(cornell)
    o = Oracle(16, 16, abi.F32)
    o.load_scene(cornell)
    rays = np.zeros(4, dtype=abi.RAY_DTYPE)
    rays["source"] = (0.31, 0.17, 3.5)  # off the wall's diagonal: the shared edge itself is a strict-test miss (Q14)
    rays["direction"] = (0.0, 0.0, -1.0)
    rays["max_distance"] = [1e5, 1e5, 4.0, 1e5]
    rays["min_distance"] = [0.0, 0.0, 0.0, 4.6]
    rays["skip_vao"] = 0xFFFFFFFF
    h0 = r.trace(rays[:1])[0]
    assert h0["hit"] == 1 and abs(h0["t"] - 4.5) < 1e-5  # back wall at z = -1
    rays["skip_vao"][1] = cornell.objects[h0["inscene_objid"]]["objectid"]
    rays["skip_inobject_offset"][1] = h0["inobject_offset"]
    hg, ho = r.trace(rays), o.trace(rays)
    assert np.array_equal(hg["hit"], ho["hit"]) and np.array_equal(hg["t"], ho["t"])
    assert hg["hit"][1] == 0 or hg["inobject_offset"][1] != h0["inobject_offset"]
    assert hg["hit"][2] == 0 and hg["hit"][3] == 0


@pytest.mark.parametrize("builder", ["median", "sah"])
def test_blas_is_a_valid_hierarchy(sponza, builder):
    """GPU build (B1), both builders: the leaf order is a permutation; walking the node records from the root (children of
    slot i at 2i, 2i + 1; a leaf word per child) every triangle lies in exactly one leaf, every leaf holds 1..2 triangles,
    every child box contains its triangles (leaf) or both of its own child boxes (inner node), and no node lies deeper than
    the depth the BLAS reports.  Median builder: depth = floor(log2 n) and every leaf at that depth; surface-area builder:
    depth bound ceil(log2(ceil(n / 2))) + 1, leaves wherever a node has <= 2 triangles left."""
    INNER = 0xFFFFFFFF
    r = lprt_amd.Renderer(16, 16, abi.F32)
    r.set_option("blas_builder", {"median": 0, "sah": 2}[builder])
    r.set_rdresource(sponza)
    pos = sponza.vertices["position"]
    for vao in (0, 4, 8, 23):
        info = r.blas_info(vao)
        n, D = info["n_tris"], info["depth"]
        assert n == sponza.vaos[vao][3] // 3 and info["n_nodes"] == (1 << D) - 1
        assert D == (int(np.floor(np.log2(n))) if builder == "median" else int(np.ceil(np.log2(np.ceil(n / 2)))) + 1)
        order = r.blas_order(vao)
        assert np.array_equal(np.sort(order), np.arange(n))
        root, boxes, words = r.blas_nodes(vao)
        tri = pos[sponza.indices[sponza.vaos[vao][2]:sponza.vaos[vao][2] + 3 * n].reshape(n, 3)]  # [n,3,3]
        tlo, thi = tri.min(axis=1)[order], tri.max(axis=1)[order]
        seen = np.zeros(n, dtype=np.int32)
        leaf_depths = []
        stack = [(1, 0, root)]
        while stack:
            i, d, box = stack.pop()
            assert i <= (1 << D) - 1 and d < D
            for k in (0, 1):
                cb, w = boxes[i - 1, k], int(words[i - 1, k])
                assert (box[:3] <= cb[:3]).all() and (box[3:] >= cb[3:]).all(), (vao, i, k)
                if w == INNER:
                    stack.append((2 * i + k, d + 1, cb))
                else:
                    first, cnt = w >> 3, w & 7
                    assert 1 <= cnt <= 2
                    seen[first:first + cnt] += 1
                    leaf_depths.append(d + 1)
                    assert (cb[:3] <= tlo[first:first + cnt].min(axis=0)).all() and (cb[3:] >= thi[first:first + cnt].max(axis=0)).all()
        assert (seen == 1).all()
        assert max(leaf_depths) <= D and (builder == "sah" or min(leaf_depths) == D)


def pathological_scene(seed=5):
    """Meshes that take the corners of the surface-area builder (build.hip::k_sah_level): every triangle the same (coincident
    centroids: no admissible plane, the object median by position); triangles strung along one axis (two of three axes
    without extent); sizes at the seams between the code paths -- 4 and 5 (the smallest trees with records), 7, 64 / 65 (one
    wave's worth), 256 / 257 (wave-per-node vs workgroup-per-node), 1025, 4100 (one-axis binning); a mesh with degenerate
    (zero-area) and duplicated triangles; a huge triangle among small ones (the capacity rule excludes lopsided planes)."""
    from lprt_amd.scenes import Mesh, assemble, translate
    rng = np.random.default_rng(seed)

    def mesh_from(tri):
        tri = np.asarray(tri, dtype=np.float64)
        nt = len(tri)
        nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        ln = np.linalg.norm(nrm, axis=1, keepdims=True)
        nrm = np.where(ln > 0, nrm / np.maximum(ln, 1e-30), np.array([[0.0, 1.0, 0.0]]))
        return Mesh(tri.reshape(-1, 3), np.repeat(nrm, 3, axis=0), np.arange(3 * nt, dtype=np.uint32))

    def soup(nt, spread=1.0, size=0.2):
        return rng.uniform(-spread, spread, size=(nt, 1, 3)) + rng.normal(scale=size, size=(nt, 3, 3))

    one = np.array([[0.0, 0.0, 0.0], [0.5, 0.0, 0.1], [0.1, 0.6, 0.0]])
    meshes = [mesh_from(np.repeat(one[None], 37, axis=0)),                                            # 0: 37 identical triangles
              mesh_from(np.stack([one * 0.05 + np.array([0.13 * k, 0.0, 0.0]) for k in range(70)])),   # 1: a string along x
              mesh_from(soup(4)), mesh_from(soup(5)), mesh_from(soup(7)), mesh_from(soup(64)), mesh_from(soup(65)),
              mesh_from(soup(256)), mesh_from(soup(257)), mesh_from(soup(1025, 1.5, 0.08)), mesh_from(soup(4100, 2.0, 0.05))]
    weird = soup(90)
    weird[10:20, 2] = weird[10:20, 1]          # zero-area triangles
    weird[30:40] = weird[50:60]                 # duplicates
    weird[0] = np.array([[-3.0, -3.0, 0.3], [3.0, -3.0, 0.3], [0.0, 3.0, 0.3]])  # one huge triangle
    meshes.append(mesh_from(weird))
    materials = [dict(color=(0.7, 0.7, 0.7), metallic=0.0, roughness=1.0, double_sided=True)]
    instances = [(i, 0, translate((4.5 * (i % 4) - 6.0, 4.5 * (i // 4) - 4.0, 0.0))) for i in range(len(meshes))]
    lights = [dict(type=abi.LIGHT_POINT, position=(0.0, 0.0, 9.0), intensity=(50.0, 50.0, 50.0))]
    return assemble("pathological", meshes, materials, instances, lights, cam_eye=(0.0, 0.0, 16.0), cam_target=(0, 0, 0), fovy=0.9)


@pytest.mark.parametrize("prec", [abi.F32, abi.F16])
def test_surface_area_builder_on_pathological_meshes(prec):
    """Every corner of k_sah_level (pathological_scene) at hit-record level: the trees are valid hierarchies over all the
    triangles and cast_ray reports what the oracle reports (single precision: exactly, ties excepted; half precision:
    disputed hits adjudicated by exhaustive search), closest-hit and any-hit, one- and two-level walk."""
    INNER = 0xFFFFFFFF
    sc = pathological_scene()
    rays = random_rays(sc, 30000, 21, 1e-4 if prec == abi.F32 else 0.1)
    with Oracle(16, 16, prec) as o:
        o.load_scene(sc)
        ho, hoa = o.trace(rays, False), o.trace(rays, True)
        for wide in (0, 1):
            with lprt_amd.Renderer(16, 16, prec) as r:
                r.set_option("blas_builder", 2)
                r.set_option("wide_walk", wide)
                r.load_scene(sc)
                if not wide:
                    for vao in range(len(sc.vaos)):
                        info = r.blas_info(vao)
                        n, D = info["n_tris"], info["depth"]
                        assert D == int(np.ceil(np.log2(np.ceil(n / 2)))) + 1
                        root, boxes, words = r.blas_nodes(vao)
                        seen = np.zeros(n, dtype=np.int32)
                        stack = [(1, 0)]
                        while stack:
                            i, d = stack.pop()
                            assert d < D
                            for k in (0, 1):
                                w = int(words[i - 1, k])
                                if w == INNER:
                                    stack.append((2 * i + k, d + 1))
                                else:
                                    assert 1 <= (w & 7) <= 2
                                    seen[w >> 3:(w >> 3) + (w & 7)] += 1
                        assert (seen == 1).all(), vao
                hg, hga = r.trace(rays, False), r.trace(rays, True)
            if prec == abi.F32:
                compare_hits(ho, hg, False, 0.05)   # (37 coincident triangles: every hit of that mesh is a tie)
                compare_hits(hoa, hga, True, 0.0)
            else:
                d = replay.closest_differs(hg, ho)
                assert d.mean() < 0.02
                j = replay.adjudicate(o, rays[d], hg[d], ho[d])
                assert int((~j["gpu_nearest"] & j["oracle_nearest"]).sum()) <= max(3, int((j["gpu_nearest"] & ~j["oracle_nearest"]).sum()))
                assert (hga["hit"] != hoa["hit"]).mean() < 5e-3


def test_ragged_and_tiny_primitives():
    """1-, 2- and 3-triangle primitives (depth-0/1 trees) and an instance placed twice."""
    sc = scenes.small_test_scene(seed=99, n_objects=6)
    assert sorted(int(v[3]) // 3 for v in sc.vaos)[:3] == [1, 2, 3]
    o = Oracle(16, 16, abi.F32)
    o.load_scene(sc)
    r = lprt_amd.Renderer(16, 16, abi.F32)
    r.load_scene(sc)
    rays = random_rays(sc, 30000, 5, 1e-4)
    compare_hits(o.trace(rays, False), r.trace(rays, False), False, 0.01)


def test_errors_are_reported_not_thrown(cornell):
    r = lprt_amd.Renderer(16, 16, abi.F32)
    with pytest.raises(lprt_amd.LprtError, match="BLAS not built|no scene"):
        r.trace(np.zeros(1, dtype=abi.RAY_DTYPE))
    r.set_rdresource(cornell, build=False)
    with pytest.raises(lprt_amd.LprtError):
        r.render()
    bad = scenes.cornell_box()
    bad.indices = bad.indices.copy()
    bad.indices[0] = 10**6
    with pytest.raises(lprt_amd.LprtError, match="index out of range"):
        r.set_rdresource(bad)


def test_rng_states_match_curand_restatement():
    """k_rng_expand against the oracle's curand_init(seed, pixel, 0) for pixels across a 1920x1080 frame."""
    import ctypes as C
    from oracle import binding
    r = lprt_amd.Renderer(1920, 1080, abi.F32)
    L = binding.lib()
    for seed in (1, 2, 0x123456789abcdef):
        for first, count in ((0, 130), (4095, 70), (262143, 5), (1920 * 1080 - 64, 64), (1234567, 200)):
            got = r.rng_states(seed, first, count)
            for i in (0, 1, count // 2, count - 1):
                want = np.zeros(6, dtype=np.uint32)
                L.orc_xorwow_state(seed, first + i, want.ctypes.data_as(C.c_void_p))
                assert np.array_equal(got[i], want), (seed, first + i)


@pytest.mark.parametrize("prec", [abi.F32, abi.F16])
def test_stationary_and_near_stationary_axes_do_not_walk_the_whole_tree(sponza, prec):
    """Rays that do not move along one or two axes (Box-Muller hands shade() an exactly axis-aligned GI direction every few
    frames) and, in half precision, rays whose slab parameters overflow 65504: the reference ignores those axes, so such
    a ray passes every box in front of it.  The product rejects boxes the ray cannot reach (trace.hpp: stationary_reject,
    overflowed_axes_miss) — same hits, a bounded walk."""
    rng = np.random.default_rng(3)
    n = 512
    rays = np.zeros(n, dtype=abi.RAY_DTYPE)
    lo, hi = sponza.aabbs[:, :3].min(axis=0) + 0.5, sponza.aabbs[:, 3:].max(axis=0) - 0.5
    rays["source"] = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = np.zeros((n, 3), dtype=np.float32)
    axis = rng.integers(0, 3, size=n)
    d[np.arange(n), axis] = rng.choice([-1.0, 1.0], size=n)
    tiny = rng.normal(size=(n, 3)).astype(np.float32) * np.float32(1e-5)  # second half: almost axis-aligned
    d[n // 2:] += tiny[n // 2:]
    d[n // 2:] /= np.linalg.norm(d[n // 2:], axis=1, keepdims=True)
    d[:8] *= np.float32(-0.0) + 1  # keep a few signed zeros: (-0, -0, -1) is what shade() produced
    rays["direction"] = d
    rays["min_distance"] = 1e-4 if prec == abi.F32 else 0.1
    rays["max_distance"] = 1e5
    rays["skip_vao"] = 0xFFFFFFFF
    o = Oracle(16, 16, prec)
    o.load_scene(sponza)
    r = lprt_amd.Renderer(16, 16, prec)
    r.load_scene(sponza)
    r.enable_counters(True)
    for any_hit in (False, True):
        r.reset_counters()
        hg, ho = r.trace(rays, any_hit), o.trace(rays, any_hit)
        c = r.counters()
        assert c["blas_node_visits"] / n < 400, c   # ~85 for a random ray; 131 k for an axis-aligned one before
        if prec == abi.F32:
            compare_hits(ho, hg, any_hit, 0.02)
        else:
            assert (ho["hit"] != hg["hit"]).mean() < 0.02
            same = (ho["hit"] == 1) & (hg["hit"] == 1) & (ho["inobject_offset"] == hg["inobject_offset"])
            assert same.sum() > 0.9 * ((ho["hit"] == 1) & (hg["hit"] == 1)).sum() or any_hit


def test_tlas_tree_walk_with_many_instances():
    """More than 32 instances: cast_ray walks the TLAS tree (the flat scan is for small scenes)."""
    sc = scenes.instanced_field(150)
    assert len(sc.objects) > 32
    o = Oracle(16, 16, abi.F32)
    o.load_scene(sc)
    r = lprt_amd.Renderer(16, 16, abi.F32)
    r.load_scene(sc)
    rays = random_rays(sc, 50000, 21, 1e-4)
    for any_hit in (False, True):
        compare_hits(o.trace(rays, any_hit), r.trace(rays, any_hit), any_hit, 0.01)
    assert o.trace(rays, False)["hit"].mean() > 0.2


@pytest.mark.parametrize("scene_name,n", [("instanced_field", 50000), ("sponza_like", 40000), ("small_test", 30000)])
def test_ray_stream_hit_records_equal_the_per_pixel_walk_and_the_oracle(scene_name, n):
    """lprt_trace_rays through the persistent stream kernel (LPRT_OPT_RAY_STREAM = 1: TLAS tree walk on the 150-instance
    field, flat scan elsewhere, one- and two-level BLAS steps) against the one-ray-per-lane kernel: identical hit records,
    closest and any, both precisions; and against the oracle in single precision."""
    sc = scenes.SCENES[scene_name]() if scene_name != "instanced_field" else scenes.instanced_field(150)
    for prec, tmin in ((abi.F32, 1e-4), (abi.F16, 0.1)):
        rays = random_rays(sc, n, 31, tmin)
        hits = {}
        for stream in (0, 1):
            with lprt_amd.Renderer(16, 16, prec) as r:
                r.set_option("ray_stream", stream)
                r.load_scene(sc)
                hits[stream] = (r.trace(rays, False), r.trace(rays, True))
        for any_hit in (0, 1):
            a, b = hits[0][any_hit], hits[1][any_hit]
            for f in ("hit", "inscene_objid", "inobject_offset", "t", "barycentric"):
                assert np.array_equal(a[f], b[f]), (prec, any_hit, f)
        if prec == abi.F32:
            with Oracle(16, 16, prec) as o:
                o.load_scene(sc)
                for any_hit in (False, True):
                    compare_hits(o.trace(rays, any_hit), hits[1][int(any_hit)], any_hit, 0.01)


def test_local_subtree_ordering_equals_the_all_global_build(sponza, small_scene):
    """build.hip: below 1024 triangles per node one workgroup finishes the subtree's median ordering in LDS (bitonic sort
    of sub-node | key | position); the permutation must be the one the level-by-level global radix sorts produce."""
    for sc in (small_scene, sponza):
        orders = {}
        for mode in ("0", "1"):
            r = lprt_amd.Renderer(16, 16, abi.F32)
            r.set_option("blas_builder", 0)  # the median builder (the atrium's default is the surface-area builder)
            r.set_option("blas_local", int(mode))
            r.set_rdresource(sc)
            orders[mode] = [r.blas_order(v) for v in range(len(sc.vaos))]
            r.close()
        for a, b in zip(orders["0"], orders["1"]):
            assert np.array_equal(a, b)
            assert np.array_equal(np.sort(a), np.arange(len(a)))
