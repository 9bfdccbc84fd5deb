"""Ray replay for frame parity: tells "a ray found something else" apart from "the arithmetic differs".

A frame's radiance planes are functions of (G-buffer, RNG seeds, ray results).  Where the HIP frame and the oracle
frame differ, the cause is one of: (1) a primary / GI / shadow ray reported another triangle or another visibility --
legal only at exact-t ties between coplanar triangles and, in half precision, where box culling
depends on the tree (Q2: the reference's half slab test loses hits, and which ones depends on the hierarchy
std::nth_element happens to build); or (2) the shading / accumulation arithmetic differs -- never legal beyond a few
ulps.  This module reads the rays the HIP frame kernels actually cast back through the C-ABI (G-buffer, GI direction,
light commands, skip records), replays them through lprt_trace_rays AND the oracle's cast_ray, and returns the pixel
sets of kind (1), so that a test can bound their number and hold everything else to a tight per-pixel bound.

Test infrastructure (uses the oracle): imported by tests/ and tools/ only."""
import numpy as np

from lprt_amd import abi

MOVE_FORWARD = {abi.F32: 1e-4, abi.F16: 1e-1}  # RAY_MOVEFORWARD_T (rt/rtrt/cuda.hpp:934-942)


def sample_seeds(frame, spp, sample):
    """The seed triple of sample `sample` of frame `frame` when lprt_frame_settings.seeds is NULL (include/lprt.h)."""
    a = 3 * (frame * spp + sample) + 1
    return [a, a + 1, a + 2]


def _rays(n):
    rays = np.zeros(n, dtype=abi.RAY_DTYPE)
    rays["max_distance"] = 1e5
    rays["skip_vao"] = 0xFFFFFFFF
    return rays


def primary_mismatch(r, o):
    """Pixels whose primary hit differs (covered flag, instance, triangle)."""
    cov_r, cov_o = r.read("G_COLOR")[..., 3] > 0, o.read("G_COLOR")[..., 3] > 0
    d = (cov_r != cov_o) | (cov_r & ((r.read("G_OBJID")[..., 0] != o.read("G_OBJID")[..., 0]) | (r.read("G_OFFSET")[..., 0] != o.read("G_OFFSET")[..., 0])))
    return d


def gi_rays(r, sc, prec):
    """The GI rays the last sample of r's last frame cast (trace_gi, cuda.hpp:2083-2105): source = G-buffer position,
    direction = TraceGICommand direction, skip = the primary hit.  Returns (rays, pixel mask)."""
    gd = r.read("GI_DIRECTION")
    pos = r.read("G_POSITION")
    mask = gd[..., 3] == 1
    obj = r.read("G_OBJID")[..., 0].astype(np.int64)
    vao = np.array([ob["objectid"] for ob in sc.objects], dtype=np.uint32)
    rays = _rays(int(mask.sum()))
    rays["source"] = pos[mask][:, :3]
    rays["direction"] = gd[mask][:, :3]
    rays["min_distance"] = MOVE_FORWARD[prec]
    rays["skip_vao"] = vao[obj[mask]]
    rays["skip_inobject_offset"] = r.read("G_OFFSET")[..., 0][mask].astype(np.uint32)
    return rays, mask


def shadow_rays(r, prec, first_round):
    """The shadow rays of the LAST shade of r's last frame (trace_di_light, cuda.hpp:2058-2081): round 1 (source = GI
    hit) after a gi_on frame, round 0 (source = G-buffer position) after a gi_off frame.  Returns (rays, pixel index of
    each ray, flat)."""
    cmd = r.read("LIGHT_DIR_MAXT")                     # [4][rows][W][4]
    lmask = r.read("LIGHT_MASK")[..., 0].astype(np.uint32)
    src = (r.read("G_POSITION") if first_round else r.read("SHADE_POSITION"))[..., :3]
    skip_f = r.read("SKIP")  # exact integers as float32, except UINT32_MAX ("skip nothing"), which reads back as 2^32
    nothing = skip_f >= 4294967040.0
    skip = np.where(nothing, 0.0, skip_f).astype(np.uint32)
    skip[nothing] = 0xFFFFFFFF
    rows, W = lmask.shape
    out, pix = [], []
    for s in range(abi.MAX_DIRECT_LIGHT):
        m = (lmask >> s) & 1 == 1
        rays = _rays(int(m.sum()))
        rays["source"] = src[m]
        rays["direction"] = cmd[s][m][:, :3]
        rays["min_distance"] = MOVE_FORWARD[prec]
        rays["max_distance"] = cmd[s][m][:, 3]
        rays["skip_vao"] = skip[m][:, 0]
        rays["skip_inobject_offset"] = skip[m][:, 1]
        out.append(rays)
        pix.append(np.flatnonzero(m.reshape(-1)))
    return np.concatenate(out), np.concatenate(pix), (rows, W)


def closest_differs(hg, ho):
    return (hg["hit"] != ho["hit"]) | ((hg["hit"] == 1) & ((hg["inscene_objid"] != ho["inscene_objid"]) | (hg["inobject_offset"] != ho["inobject_offset"])))


def _ray_slots(pix):
    """rays are concatenated slot by slot; the slot of each ray = how often the pixel order has restarted before it"""
    slot = np.zeros(len(pix), dtype=np.uint32)
    if len(pix):
        starts = np.flatnonzero(np.diff(pix.astype(np.int64)) < 0) + 1
        slot[starts] = 1
        slot = np.cumsum(slot).astype(np.uint32)
    return slot


def _visibility_bits(x, rays, pix, npix, keep=None):
    """bit s of pixel p: the shadow ray of light slot s is blocked (own commands on own tracer).  keep: a list that receives
    (rays, pixel * 4 + slot, blocked) for adjudicate_shadow()."""
    vis = x.trace(rays, True)["hit"].astype(np.uint32)
    slot = _ray_slots(pix)
    if keep is not None:
        keep.append((rays, pix.astype(np.int64) * 4 + slot, vis))
    bits = np.zeros(npix, dtype=np.uint32)
    np.bitwise_or.at(bits, pix, vis << slot)
    return bits


def _side(x, sc, prec, frame, spp, rebuild_blas):
    """Everything the rays of one frame found on one implementation (x: a Renderer or an Oracle used as a PROBE: its
    history is reset).  Per-sample stages depend on nothing but the G-buffer and the seeds, so a 1-spp frame with sample
    s's seeds casts sample s's rays; the planes it leaves behind are then read back and its own rays replayed through
    its own cast_ray."""
    out = {"gi": [], "gi_rays": [], "gi_mask": [], "shadow1": [], "shadow1_mask": [], "n_rays": 0, "shadow_rays": []}

    def run(seeds, gi_on):
        fs = abi.default_settings(spp=1, gi_on=gi_on, svgf=False, seeds=seeds)
        fs.rebuild_blas = rebuild_blas
        x.reset_history()
        x.set_scene(sc)
        x.render_frame(fs)

    run(sample_seeds(frame, spp, 0), False)  # round 0: its commands depend on the G-buffer and the lights only
    rays, pix, (rows, W) = shadow_rays(x, prec, True)
    out["shadow0"] = _visibility_bits(x, rays, pix, rows * W, out["shadow_rays"]).reshape(rows, W)
    out["shadow0_mask"] = x.read("LIGHT_MASK")[..., 0].astype(np.uint32)
    out["n_rays"] += len(rays)
    for s in range(spp):
        run(sample_seeds(frame, spp, s), True)
        rays, mask = gi_rays(x, sc, prec)
        hits = x.trace(rays, False)
        rec = np.zeros((rows, W), dtype=abi.HIT_DTYPE)
        rec[mask] = hits
        full = np.zeros((rows, W), dtype=abi.RAY_DTYPE)
        full[mask] = rays
        out["gi"].append(rec)
        out["gi_rays"].append(full)
        out["gi_mask"].append(mask)
        rays1, pix, _ = shadow_rays(x, prec, False)
        out["shadow1"].append(_visibility_bits(x, rays1, pix, rows * W, out["shadow_rays"]).reshape(rows, W))
        out["shadow1_mask"].append(x.read("LIGHT_MASK")[..., 0].astype(np.uint32))
        out["n_rays"] += len(rays) + len(rays1)
    return out


def classify(r, o, sc, prec, probe_g, probe_o, frame, spp, rebuild_blas=0):
    """Pixel sets where some ray of frame `frame` (spp samples, default seeds) found something else on the GPU than in the
    oracle.  `r` / `o` have just rendered that frame (primary hits are compared on them); `probe_g` / `probe_o` are a second
    HIP renderer and a second oracle holding the same scene, which re-run the frame's samples one by one (_side).

    Returns boolean [rows][W] masks: primary, shadow0, gi[s], shadow1[s], any, last_sample; and for the closest-hit rays
    the split of the differing ones by cause:
      gi_tie[s]      both implementations cast the bit-identical ray and report different triangles at the same t
                     (coplanar geometry) -- legal in either precision;
      gi_other_ray[s] the two rays are not bit-identical (the shading that produced the direction differs in its last
                     bits: libm vs device transcendentals), so they are different rays;
      gi_cull[s]     identical ray, different result at a different t: box culling -- legal in half precision only."""
    g, c = _side(probe_g, sc, prec, frame, spp, rebuild_blas), _side(probe_o, sc, prec, frame, spp, rebuild_blas)
    res = {"primary": primary_mismatch(r, o), "gi": [], "shadow1": [], "gi_tie": [], "gi_other_ray": [], "gi_cull": []}
    res["shadow0"] = (g["shadow0"] != c["shadow0"]) | (g["shadow0_mask"] != c["shadow0_mask"])
    for s in range(spp):
        hg, ho = g["gi"][s], c["gi"][s]
        d = closest_differs(hg, ho) | (g["gi_mask"][s] != c["gi_mask"][s])
        same_ray = g["gi_mask"][s] & c["gi_mask"][s]
        for f in ("source", "direction"):
            same_ray &= (g["gi_rays"][s][f].view(np.uint32) == c["gi_rays"][s][f].view(np.uint32)).all(axis=-1)
        both = (hg["hit"] == 1) & (ho["hit"] == 1)
        same_t = both & np.isclose(hg["t"], ho["t"], rtol=1e-5, atol=1e-6)
        res["gi"].append(d)
        res["gi_tie"].append(d & same_ray & same_t)
        res["gi_other_ray"].append(d & ~same_ray)
        res["gi_cull"].append(d & same_ray & ~same_t)
        res["shadow1"].append((g["shadow1"][s] != c["shadow1"][s]) | (g["shadow1_mask"][s] != c["shadow1_mask"][s]))
    res["sides"] = (g, c)  # everything both sides' rays found (for adjudicate())
    res["n_rays"] = g["n_rays"]
    res["any"] = res["primary"] | res["shadow0"] | np.any(res["gi"], axis=0) | np.any(res["shadow1"], axis=0)
    res["last_sample"] = res["primary"] | res["shadow0"] | res["gi"][-1] | res["shadow1"][-1]
    return res


def adjudicate(o, rays, hg, ho):
    """Who is right where two closest-hit answers differ?  A traversal can only LOSE candidates -- a box test that rejects a box
    the ray does pass -- never invent one: every reported hit is a triangle the reference's triangle test accepts.  So the
    exhaustive search (Oracle.trace_exhaustive: that test against every triangle, no boxes) gives the smallest t any
    hierarchy could report, and an answer whose t equals it is a nearest hit; one with a larger t lost the nearest
    triangle to a cull.  rays / hg / ho: the disputed rays and the two answers (flat arrays).  Returns boolean arrays:
      gpu_nearest, oracle_nearest   the answer's t is the exhaustive t (or both say miss);
    (both set: an exact-t tie between two triangles; neither: both walks lost it)."""
    e = o.trace_exhaustive(rays)

    def nearest(h):
        return ((h["hit"] == 0) & (e["hit"] == 0)) | ((h["hit"] == 1) & (e["hit"] == 1) & (h["t"] == e["t"]))

    return {"gpu_nearest": nearest(hg), "oracle_nearest": nearest(ho), "exhaustive": e}


def adjudicate_gi(c, o, kinds=("gi_cull", "gi_tie")):
    """The disputed closest-hit rays of a classified frame (bit-identical ray on both sides, different answer), every
    sample, adjudicated by exhaustive search on oracle `o`.  Returns counts: disputed, gpu_nearest, oracle_nearest,
    gpu_lost (the oracle reports the nearest hit and the GPU a farther one), oracle_lost, neither."""
    g, cc = c["sides"]
    tot = dict(disputed=0, gpu_nearest=0, oracle_nearest=0, gpu_lost=0, oracle_lost=0, neither=0)
    for s in range(len(c["gi"])):
        m = np.zeros_like(c["gi"][s])
        for k in kinds:
            m |= c[k][s]
        if not m.any():
            continue
        j = adjudicate(o, g["gi_rays"][s][m], g["gi"][s][m], cc["gi"][s][m])
        gn, on = j["gpu_nearest"], j["oracle_nearest"]
        tot["disputed"] += int(m.sum())
        tot["gpu_nearest"] += int(gn.sum())
        tot["oracle_nearest"] += int(on.sum())
        tot["gpu_lost"] += int((~gn & on).sum())
        tot["oracle_lost"] += int((gn & ~on).sum())
        tot["neither"] += int((~gn & ~on).sum())
    return tot


def adjudicate_shadow(c, o):
    """The disputed SHADOW rays of a classified frame (round 0 and every sample's round 1): the same bit-identical ray on both
    sides, blocked on one and not on the other.  "Blocked" is always right -- it names a triangle the reference's triangle
    test accepts inside the ray's window -- so the side that reports the ray as free lost the blocker to a box cull; the
    exhaustive search says which rays are blocked at all.  Returns counts: disputed, gpu_lost (the GPU reports a blocked ray
    as free), oracle_lost, unexplained (the exhaustive search finds no blocker although one side reports one: impossible
    unless a walk invents a hit)."""
    g, cc = c["sides"]
    tot = dict(disputed=0, gpu_lost=0, oracle_lost=0, unexplained=0)
    for (rg, kg, vg), (ro, ko, vo) in zip(g["shadow_rays"], cc["shadow_rays"]):
        common, ig, io = np.intersect1d(kg, ko, return_indices=True)
        if not len(common):
            continue
        a, b = rg[ig], ro[io]
        same = np.ones(len(common), dtype=bool)
        for f in ("source", "direction", "max_distance"):
            x, y = a[f].view(np.uint32), b[f].view(np.uint32)
            same &= (x == y).reshape(len(common), -1).all(axis=1)
        d = same & (vg[ig] != vo[io])
        if not d.any():
            continue
        e = o.trace_exhaustive(a[d])["hit"]
        gv, ov = vg[ig][d], vo[io][d]
        tot["disputed"] += int(d.sum())
        tot["gpu_lost"] += int(((gv == 0) & (ov == 1) & (e == 1)).sum())
        tot["oracle_lost"] += int(((gv == 1) & (ov == 0) & (e == 1)).sum())
        tot["unexplained"] += int((e == 0).sum())
    return tot


def dilate(mask, radius):
    """Chebyshev dilation: every pixel within `radius` of a set pixel (the footprint a filter of that reach smears it over)."""
    out = mask.copy()
    ys, xs = np.nonzero(mask)
    H, W = mask.shape
    for y, x in zip(ys, xs):
        out[max(0, y - radius):min(H, y + radius + 1), max(0, x - radius):min(W, x + radius + 1)] = True
    return out


def pixel_error(a, b, floor):
    """max over channels of |a - b| / (|b| + floor), per pixel; non-finite values must coincide."""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    fin = np.isfinite(a) & np.isfinite(b)
    same_class = fin | (np.isnan(a) & np.isnan(b)) | ((a == b) & ~fin)
    e = np.where(fin, np.abs(np.where(fin, a, 0) - np.where(fin, b, 0)) / (np.abs(np.where(fin, b, 0)) + floor), np.where(same_class, 0.0, np.inf))
    return e.max(axis=-1)
