"""GPU frame parity through the C-ABI.  Tolerances are the north star's: radiance buffers within 1e-3 relative L2
(fp32) and 5e-2 (fp16) of the reference semantics (here: the CPU oracle and the committed golden vectors)."""
import os

import numpy as np
import pytest

import lprt_amd
from lprt_amd import abi, scenes, dist
from oracle.binding import Oracle
from conftest import rel_l2
import replay

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
TOL = {abi.F32: 1e-3, abi.F16: 5e-2}
RADIANCE = ["INTENSITY0", "INTENSITY1", "RAW_COLOR", "RAW_GI_COLORED", "RAW_GI_WHITE", "MUL_GI_COLORED", "MUL_GI_WHITE",
            "COLOR_INPROGRESS", "OUTPUT"]


# Half precision runs in both slab modes (LPRT_OPT_REFERENCE_SLAB, include/lprt.h): 0 = the product's default (slab parameters
# of half boxes in f32: the GPU never culls a box the ray reaches), 1 = ray_aabb_intersection_* in float16 operation for
# operation as the reference (its own lossy test).  Single precision has one mode: its arithmetic is the reference's.
PREC_SLAB = [(abi.F32, 0), (abi.F16, 0), (abi.F16, 1)]
PREC_SLAB_IDS = ["f32", "f16", "f16-refslab"]


def renderer(W, H, prec, slab=0, **kw):
    r = lprt_amd.Renderer(W, H, prec, **kw)
    if slab:
        r.set_option("reference_slab", slab)
    return r


def render_pair(sc, W, H, prec, frames=1, slab=0, **kw):
    o = Oracle(W, H, prec)
    o.load_scene(sc)
    r = renderer(W, H, prec, slab)
    r.load_scene(sc)
    for _ in range(frames):
        fs = abi.default_settings(**kw)
        o.set_scene(sc)
        o.render_frame(fs)
        r.set_scene(sc)
        r.render(fs)
    return r, o


@pytest.mark.parametrize("prec,slab", PREC_SLAB, ids=PREC_SLAB_IDS)
def test_c1_golden_cornell_256(cornell, prec, slab):
    """BASELINE configs[0] on the GPU against the committed oracle vectors (fp32) / live oracle (fp16)."""
    r = renderer(256, 256, prec, slab)
    r.load_scene(cornell)
    r.render(abi.default_settings(spp=1))
    if prec == abi.F32:
        g = np.load(os.path.join(GOLD, "cornell_c1_f32.npz"))
        assert np.array_equal(r.read("G_OBJID")[..., 0].astype(np.uint8), g["g_objid"])
        assert np.array_equal(r.read("G_OFFSET")[..., 0].astype(np.uint8), g["g_offset"])
        assert np.array_equal(r.read("G_COLOR")[..., 3] > 0, g["g_alpha"])
        assert rel_l2(r.read("OUTPUT")[..., :3], g["output"]) < TOL[prec]
    else:
        o = Oracle(256, 256, prec)
        o.load_scene(cornell)
        o.render_frame(abi.default_settings(spp=1))
        assert (r.read("G_OBJID") != o.read("G_OBJID")).mean() < 1e-3
        for name in RADIANCE:
            assert rel_l2(r.read(name), o.read(name)) < TOL[prec], name


PRE_DENOISE = ["INTENSITY0", "INTENSITY1", "RAW_COLOR", "RAW_GI_COLORED", "RAW_GI_WHITE"]
POST_DENOISE = ["MUL_GI_COLORED", "MUL_GI_WHITE", "COLOR_INPROGRESS", "OUTPUT"]


def assert_planes_ray_by_ray(r, o, c, prec, kept_l2, kept_px, max_outliers):
    """Frame parity with the pixels set aside where tests/replay.py found a ray that reported something else (c: its
    classification).  Pre-denoise planes: over the other pixels, relative L2 <= kept_l2 and per-pixel relative error
    <= kept_px except for at most max_outliers pixels (an `other ray` hit of the same triangle at a grazing angle).  With no
    such pixel at all, every plane, denoised ones included, is held to the north star's whole-buffer tolerance."""
    for name in PRE_DENOISE:
        a, b = r.read(name), o.read(name)
        keep = ~(c["last_sample"] if name in ("INTENSITY0", "INTENSITY1") else c["any"])
        assert rel_l2(a[keep], b[keep]) <= kept_l2, (name, rel_l2(a[keep], b[keep]))
        e = replay.pixel_error(a, b, 1e-3)[keep]
        assert int((e > kept_px).sum()) <= max_outliers, (name, float(e.max()), int((e > kept_px).sum()))
    if not c["any"].any():
        for name in RADIANCE:
            assert rel_l2(r.read(name), o.read(name)) < TOL[prec], name


@pytest.mark.parametrize("prec,slab", PREC_SLAB, ids=PREC_SLAB_IDS)
@pytest.mark.parametrize("scene_name,W,H", [("cornell", 160, 96), ("small_test", 200, 120), ("textured", 192, 108), ("instanced_field", 160, 90)])
def test_every_buffer_matches_oracle(scene_name, W, H, prec, slab):
    """One frame, every plane.  Single precision: integer planes bit-exact, every radiance plane within 1e-3 over ALL pixels
    (no ray of these scenes finds anything else than in the oracle).  Half precision: a few rays per frame do (primary
    rays at silhouettes, GI rays whose direction differs in its last bit, ties at half resolution of t: 0.1 % of the pixels
    of `instanced_field`); tests/replay.py names those pixels by replaying every ray of the frame on both sides, their
    number is bounded, and every other pixel of the pre-denoise planes agrees to 1e-4 -- where the rays agree, the
    arithmetic does."""
    sc = scenes.SCENES[scene_name]()
    r, o = render_pair(sc, W, H, prec, frames=1, slab=slab, spp=1)
    if prec == abi.F32:  # integer planes bit-exact
        for name in ("G_OBJID", "G_OFFSET", "G_COLOR"):
            assert np.array_equal(r.read(name), o.read(name)), name
        for name in ("G_POSITION", "G_NORMAL_DEPTH", "G_TANGENT", "G_UV0UV1", "ALBEDO"):
            assert rel_l2(r.read(name), o.read(name)) < 1e-6, name
        for name in RADIANCE:
            a, b = r.read(name), o.read(name)
            assert np.array_equal(np.isfinite(a), np.isfinite(b)), name
            assert rel_l2(a, b) < TOL[prec], name
        return
    probe_g, probe_o = renderer(W, H, prec, slab), Oracle(W, H, prec)
    probe_g.load_scene(sc)
    probe_o.load_scene(sc)
    c = replay.classify(r, o, sc, prec, probe_g, probe_o, frame=0, spp=1)
    assert c["any"].mean() < 2.5e-3, int(c["any"].sum())
    assert sum(int(m.sum()) for m in c["gi_cull"]) <= 2
    # since the transcendentals that shape rays are shared (include/lprt_fmath.h) two implementations that agree on the
    # primary hit cast the bit-identical GI ray: an `other ray` can only follow from a primary ray that found something else
    assert not (c["gi_other_ray"][0] & ~c["primary"]).any()
    assert_planes_ray_by_ray(r, o, c, prec, kept_l2=1e-4, kept_px=1e-2, max_outliers=0)
    assert rel_l2(r.read("OUTPUT"), o.read("OUTPUT")) < TOL[prec]  # what is displayed, over all pixels


# Whole-buffer ceilings of the atrium's half-precision frames (relative L2 over ALL pixels, HIP vs oracle), recorded from
# the round-3 run (gpurun_out/r3_probe1.log: measured maxima over both frames and both slab modes in brackets) with ~1.4x
# headroom, so that drift is caught.  They are NOT the north star's 5e-2: on this workload the oracle misses that bar
# against ITSELF when only the order of equal split keys changes (tests/golden/topology_baseline_sponza_f16.json,
# tools/oracle_topology_baseline.py: INTENSITY1 0.14, RAW_GI_COLORED 0.066, MUL_GI_WHITE 0.050) -- one sample of one-bounce
# radiance is a few bright pixels, and in half precision which triangle wins at the "same" t depends on the hierarchy.
ATRIUM_F16_CEILING = {"INTENSITY0": 0.025, "INTENSITY1": 0.25, "RAW_COLOR": 0.025, "RAW_GI_COLORED": 0.10, "RAW_GI_WHITE": 0.30,
                      "MUL_GI_COLORED": 0.085, "MUL_GI_WHITE": 0.12, "COLOR_INPROGRESS": 0.036, "OUTPUT": 0.005}
#                      [0.017, 0.175, 0.017, 0.071, 0.213, 0.060, 0.083, 0.025, 0.0035]


@pytest.mark.parametrize("prec,slab", PREC_SLAB, ids=PREC_SLAB_IDS)
def test_atrium_frames_with_per_frame_rebuild_match_oracle_ray_by_ray(sponza, prec, slab):
    """BASELINE configs[3] workload (262 k-triangle atrium, every BLAS rebuilt on the GPU inside each frame) against the
    oracle: two frames at 2 spp, reduced to 256x144 so the oracle finishes in seconds.  Every ray of every sample is
    replayed on both sides (tests/replay.py, ~320 k rays per frame).

    Single precision: every radiance plane within 1e-3 over ALL pixels, nothing set aside.  The transcendentals that
    shape rays are shared with the oracle (include/lprt_fmath.h), so both sides cast bit-identical rays: no `other ray`, no
    `cull`; what remains are exact-t ties between coplanar triangles, a handful of pixels.

    Half precision: which triangle a ray reports depends on the hierarchy (t has 11 bits at 5-30 m; the reference's half
    slab test culls boxes the ray does pass, and which ones depends on the tree std::nth_element happens to build).  The
    statement therefore has four parts, in both slab modes:
      1. per class, the pixels where HIP and oracle disagree are at most 1.25x (primary, GI, all) / 2x (culls, shadow rays:
         the GPU's tree differs from the oracle's by more than a tie order) what the ORACLE disagrees with ITSELF under two
         legal tie orders of the reference's builder (committed baseline, tests/golden/topology_baseline_sponza_f16.json);
      2. disputed closest hits adjudicated by exhaustive search (replay.adjudicate_gi): the GPU reports a farther hit than
         the oracle no more often than the reverse -- a lost pending bit or a missed box would show up here;
      3. every other pixel of the pre-denoise planes agrees to 1e-4: where the rays agree, the arithmetic does;
      4. every radiance plane over ALL pixels under a recorded ceiling; OUTPUT and COLOR_INPROGRESS under the north star's."""
    import json
    W, H, spp = 256, 144, 2
    o, probe_o = Oracle(W, H, prec), Oracle(W, H, prec)
    r, probe_g = renderer(W, H, prec, slab), renderer(W, H, prec, slab)
    for x in (o, probe_o, r, probe_g):
        x.load_scene(sponza)
    with open(os.path.join(GOLD, "topology_baseline_sponza_f16.json")) as fh:
        base = json.load(fh)
    assert base["size"] == [W, H] and base["spp"] == spp
    for f in range(2):
        fs = abi.default_settings(spp=spp)
        fs.rebuild_blas = 1
        o.set_scene(sponza)
        o.render_frame(fs)
        r.set_scene(sponza)
        r.render(fs)
        c = replay.classify(r, o, sponza, prec, probe_g, probe_o, f, spp, rebuild_blas=1)
        count = lambda k: max(int(m.sum()) for m in c[k])
        if prec == abi.F32:
            assert count("gi_cull") == 0, "identical rays at different t: a traversal bug, not a tie"
            assert count("gi_other_ray") == 0, "the two sides cast different rays"
            assert int(c["primary"].sum()) <= 2 and int(c["shadow0"].sum()) <= 2
            assert count("gi") <= 6 and count("shadow1") <= 4 and int(c["any"].sum()) <= 10
            for name in RADIANCE:  # over ALL pixels, nothing set aside
                a, b = r.read(name), o.read(name)
                assert np.array_equal(np.isfinite(a), np.isfinite(b)), (f, name)
                assert rel_l2(a, b) < TOL[prec], (f, name, rel_l2(a, b))
            continue
        b = base["frames"][f]["px"]
        got = {"primary": int(c["primary"].sum()), "gi": count("gi"), "any": int(c["any"].sum()), "gi_cull": count("gi_cull"),
               "shadow0": int(c["shadow0"].sum()), "shadow1": count("shadow1")}
        for k, factor in (("primary", 1.25), ("gi", 1.25), ("any", 1.25), ("gi_cull", 2.0), ("shadow0", 2.0), ("shadow1", 2.0)):
            ref = max(b[k]) if isinstance(b[k], list) else b[k]
            assert got[k] <= factor * ref, (f, k, got[k], ref)
        # an `other ray` follows from a primary ray that found something else (another source point), never from the shading
        for s_ in range(spp):
            assert not (c["gi_other_ray"][s_] & ~c["primary"]).any(), (f, s_)
        j = replay.adjudicate_gi(c, probe_o)
        # (both walks lose a nearest hit now and then -- the half triangle test places t within ~1e-3 relative, a box entry can
        # lie beyond the running closest t although a triangle inside computes a smaller one: `neither` -- but the GPU's f32
        # slab parameters on outward-rounded boxes lose fewer than the reference's half ones)
        assert j["neither"] <= 2 + 0.01 * j["disputed"] and j["gpu_lost"] <= max(8, j["oracle_lost"]), f"frame {f}: {sorted(j.items())}"
        js = replay.adjudicate_shadow(c, probe_o)  # the same for the shadow rays: a blocked ray reported free is a lost blocker
        assert js["unexplained"] == 0 and js["gpu_lost"] <= max(8, js["oracle_lost"]), f"frame {f}: {sorted(js.items())}"
        assert_planes_ray_by_ray(r, o, c, prec, kept_l2=1e-4, kept_px=1e-2, max_outliers=0)
        for name in RADIANCE:
            lim = min(ATRIUM_F16_CEILING[name], TOL[prec]) if name in ("COLOR_INPROGRESS", "OUTPUT") else ATRIUM_F16_CEILING[name]
            assert rel_l2(r.read(name), o.read(name)) <= lim, (f, name, rel_l2(r.read(name), o.read(name)))


@pytest.mark.parametrize("prec", [abi.F32, abi.F16])
@pytest.mark.parametrize("scene_name,W,H", [("sponza_like", 256, 144), ("instanced_field", 160, 90), ("cornell", 160, 96)])
def test_launch_structure_never_changes_a_frame(scene_name, W, H, prec):
    """How the rays are launched is a matter of speed only.  Ray streams (csrc/stream.hpp: persistent waves that refill
    finished lanes; flat instance scan on the atrium and Cornell, the TLAS tree walk on the 150-instance field) against one
    ray per lane in the fused kernels, one stream counter against eight XCD-local ranges, and 1 / 2 / 8 samples per launch
    (sample slots, k_spp_mean in sample order): three
    frames at 3 spp come out bit-identical -- display image, the last sample's radiance of both rounds, the denoiser's
    input -- for every combination, and so do the ray counts."""
    sc = scenes.SCENES[scene_name]()
    outs = {}
    for stream, batch, xcd, sort in ((0, 1, 1, 0), (0, 8, 1, 3), (1, 1, 1, 0), (1, 2, 0, 1), (1, 8, 1, 0), (1, 8, 0, 2), (1, 8, 1, 3), (-1, 8, 1, -1)):
        with lprt_amd.Renderer(W, H, prec) as r:
            r.set_option("ray_stream", stream)
            r.set_option("sample_batch", batch)
            r.set_option("stream_xcd", xcd)  # one work counter, or eight item ranges drained XCD group by XCD group
            r.set_option("ray_sort", sort)   # stream items in pixel-block order, or sorted by (origin cell, direction octant)
            r.load_scene(sc)
            r.enable_counters(True)
            for _ in range(3):
                r.set_scene(sc)
                r.render(abi.default_settings(spp=3))
            outs[(stream, batch, xcd, sort)] = [r.read(n) for n in ("OUTPUT", "INTENSITY0", "INTENSITY1", "RAW_GI_COLORED", "RAW_COLOR", "GI_DIRECTION")] + \
                                    [np.array([r.stage_counters(st)["rays"] for st in ("primary", "trace_di1", "trace_gi", "trace_di2")])]
    first = outs[(0, 1, 1, 0)]
    for key, got in outs.items():
        for a, b in zip(first, got):
            assert np.array_equal(a, b, equal_nan=True), key


@pytest.mark.parametrize("prec", [abi.F32, abi.F16])
def test_rebuilding_every_blas_inside_the_frame_changes_nothing(sponza, prec):
    """lprt_frame_settings.rebuild_blas (config C4): three frames with the BLAS set rebuilt on the GPU before each one are
    bit-identical to three frames over the BLAS built once -- the build is deterministic and nothing of the old hierarchy
    leaks into a frame."""
    outs = []
    for rebuild in (0, 1):
        with lprt_amd.Renderer(256, 144, prec) as r:
            r.load_scene(sponza)
            for _ in range(3):
                fs = abi.default_settings(spp=2)
                fs.rebuild_blas = rebuild
                r.set_scene(sponza)
                r.render(fs)
            outs.append([r.read(n) for n in ("OUTPUT", "G_OFFSET", "INTENSITY1", "MUL_GI_WHITE")])
            if rebuild:
                assert r.blas_info(0)["n_tris"] == sponza.vaos[0][3] // 3
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("prec,slab", PREC_SLAB, ids=PREC_SLAB_IDS)
def test_temporal_path_three_frames_spp2(cornell, prec, slab):
    """frames >= 1 exercise reprojection maps, colour/moment history, frame counters (+4 per frame, Q6), TAA history."""
    o = Oracle(128, 96, prec)
    o.load_scene(cornell)
    r = renderer(128, 96, prec, slab)
    r.load_scene(cornell)
    for f in range(3):
        fs = abi.default_settings(spp=2)
        fs.taa_mix_weight = 0.5 if f else 1.0  # make the TAA history observable
        o.set_scene(cornell)
        o.render_frame(fs)
        r.set_scene(cornell)
        r.render(fs)
        for name in ("MUL_GI_COLORED", "MUL_GI_WHITE", "COLOR_INPROGRESS", "OUTPUT"):
            assert rel_l2(r.read(name), o.read(name)) < TOL[prec], (f, name)


def test_flags_gi_off_svgf_off_demodulate(cornell):
    for kw, extra in ((dict(gi_on=False), {}), (dict(svgf=False), {}), (dict(), dict(demodulate=1)), (dict(), dict(add_direct_out=0))):
        o = Oracle(96, 64, abi.F32)
        o.load_scene(cornell)
        r = lprt_amd.Renderer(96, 64, abi.F32)
        r.load_scene(cornell)
        fs = abi.default_settings(spp=1, **kw)
        for k, v in extra.items():
            setattr(fs, k, v)
        o.render_frame(fs)
        r.render(fs)
        for name in ("INTENSITY0", "COLOR_INPROGRESS", "OUTPUT"):
            assert rel_l2(r.read(name), o.read(name)) < 1e-3, (kw, extra, name)


def test_explicit_seeds_and_spp_mean(cornell):
    seeds = [101, 202, 303, 404, 505, 606]
    r, o = render_pair(cornell, 96, 64, abi.F32, spp=2, seeds=seeds)
    assert rel_l2(r.read("RAW_GI_COLORED"), o.read("RAW_GI_COLORED")) < 1e-3
    # different seeds give a different estimate
    r2 = lprt_amd.Renderer(96, 64, abi.F32)
    r2.load_scene(cornell)
    r2.render(abi.default_settings(spp=2))
    assert rel_l2(r2.read("RAW_GI_COLORED"), r.read("RAW_GI_COLORED")) > 1e-2


def test_history_reset_and_determinism(cornell):
    r = lprt_amd.Renderer(128, 72, abi.F16)
    r.load_scene(cornell)
    r.render(abi.default_settings(spp=2))
    a = r.read("OUTPUT").copy()
    r.render(abi.default_settings(spp=2))
    r.reset_history()
    r.render(abi.default_settings(spp=2))
    assert np.array_equal(r.read("OUTPUT"), a)


def test_counters_match_oracle_ray_counts(cornell):
    """rays, object visits and triangle tests of the instrumented kernels; node visits differ by design (different tree)."""
    r, o = render_pair(cornell, 96, 64, abi.F32, spp=1)
    r.reset_history()
    r.enable_counters(True)
    r.reset_counters()
    r.set_scene(cornell)
    r.render(abi.default_settings(spp=1))
    g, c = r.counters(), o.counters()
    assert g["rays"] == c["rays"]
    assert g["rays"] == sum(r.stage_counters(s)["rays"] for s in ("primary", "trace_di1", "trace_gi", "trace_di2"))
    assert r.stage_counters("primary")["rays"] == 96 * 64


@pytest.mark.parametrize("world", [2, 8])
def test_row_strips_with_halo_equal_the_full_frame(cornell, world):
    """Sharded rendering (option A): a rank's strip rendered with the 68-row halo is bit-identical to the
    same rows of the single-GPU frame (frame 0), for every rank of a 2- and an 8-way split."""
    W, H = 192, 432
    full = lprt_amd.Renderer(W, H, abi.F32)
    full.load_scene(cornell)
    full.render(abi.default_settings(spp=2))
    want = full.read("OUTPUT")
    strips = []
    for rank in range(world):
        wb, we = dist.window_rows(H, world, rank)
        b, e = dist.strip_rows(H, world, rank)
        r = lprt_amd.Renderer(W, H, abi.F32, row_begin=wb, row_end=we)
        r.load_scene(cornell)
        r.render(abi.default_settings(spp=2))
        out = r.read("OUTPUT")
        strips.append(out[b - wb:e - wb])
        r.close()
    got = dist.assemble_strips(strips, H, world)
    assert np.array_equal(got, want)


def test_output_target_writes_only_owned_rows(cornell):
    import torch
    W, H, world, rank = 96, 200, 2, 1
    wb, we = dist.window_rows(H, world, rank)
    frame = dist.ShardedFrame(W, H, rank, world, torch.device("cuda", 0))
    r = lprt_amd.Renderer(W, H, abi.F32, row_begin=wb, row_end=we)
    r.set_stream(torch.cuda.current_stream().cuda_stream)
    ptr, first, n = frame.target()
    r.set_output_target(ptr, first, n)
    r.load_scene(cornell)
    r.render(abi.default_settings())
    torch.cuda.synchronize()
    own = r.read("OUTPUT")[frame.begin - wb:frame.end - wb]
    assert np.array_equal(frame.send.cpu().numpy(), own)


def test_full_size_1080p_properties(cornell):
    """BASELINE size (1920x1080, 4 spp): size-independent checks — finite output, deterministic, ray count within the
    analytic bounds N*(1+spp*(1..3)), fp16 within tolerance of fp32."""
    W, H, spp = 1920, 1080, 4
    outs = {}
    for prec in (abi.F32, abi.F16):
        r = lprt_amd.Renderer(W, H, prec)
        r.load_scene(cornell)
        r.enable_counters(True)
        r.render(abi.default_settings(spp=spp))
        a = r.read("OUTPUT")
        rays = r.counters()["rays"]
        r.enable_counters(False)
        r.reset_history()
        r.render(abi.default_settings(spp=spp))
        assert np.array_equal(a, r.read("OUTPUT"))
        assert np.isfinite(a).all() and a[..., 3].min() == 1.0
        assert W * H * (1 + spp) <= rays <= W * H * (1 + 3 * spp)
        outs[prec] = a
        r.close()
    assert rel_l2(outs[abi.F16], outs[abi.F32]) < 5e-2


def test_gltf_loaded_cornell_renders_identically(cornell, tmp_path):
    """§8(f) rank 1: the Cornell box written as .gltf + .bin and read back through the loader contract gives the
    same frame bit for bit (same arrays; material ids shifted by the loader's default material)."""
    from lprt_amd import gltf
    back = gltf.load_gltf(gltf.save_gltf(cornell, str(tmp_path / "cornell.gltf")))
    outs = []
    for sc in (cornell, back):
        r = lprt_amd.Renderer(160, 96, abi.F32)
        r.load_scene(sc)
        r.render(abi.default_settings(spp=1))
        outs.append((r.read("OUTPUT"), r.read("G_OBJID")))
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][0], outs[1][0])


def test_texture_unit_matches_oracle_bit_for_bit():
    """§8(f) rank 2: RGBA8 fetch with wrap addressing, bilinear 1.8 fixed-point weights and the sRGB table."""
    rng = np.random.default_rng(5)
    r = lprt_amd.Renderer(16, 16, abi.F32)
    o = Oracle(16, 16, abi.F32)
    uv = np.concatenate([rng.uniform(-3, 3, size=(4096, 2)), rng.uniform(0, 1, size=(4096, 2)),
                         np.array([[0, 0], [1, 1], [0.5, 0.5], [-0.0, 1e-9], [1e9, -1e9], [np.nan, 0.3], [np.inf, -np.inf]])]).astype(np.float32)
    for (w, h), srgb in (((16, 8), True), ((5, 7), False), ((1, 1), True), ((64, 3), False)):
        img = scenes.checker_texture(w, h, seed=w + h)
        img[..., 3] = rng.integers(0, 256, size=(h, w))
        hr, ho = r.upload_texture(img, srgb), o.upload_texture(img, srgb)
        a, b = r.sample_texture(hr, uv), o.sample_texture(ho, uv)
        assert np.array_equal(a, b), (w, h, srgb, np.abs(a - b).max())
        centres = np.array([[(i + 0.5) / w, (j + 0.5) / h] for j in range(h) for i in range(w)], dtype=np.float32)
        if not srgb and w & (w - 1) == 0 and h & (h - 1) == 0:  # texel centres are exact in binary only then
            assert np.array_equal(r.sample_texture(hr, centres), img.reshape(-1, 4).astype(np.float32) / np.float32(255))


def test_skybox_fills_empty_pixels(tmp_path):
    sc = scenes.textured_scene()
    W, H = 96, 54
    r, o = render_pair(sc, W, H, abi.F32, frames=1, spp=1, svgf=0)
    empty = r.read("G_COLOR")[..., 3] == 0
    assert 0.3 < empty.mean() < 0.9
    i0 = r.read("INTENSITY0")
    assert (i0[empty][:, :3] > 0).all() and rel_l2(i0, o.read("INTENSITY0")) < 1e-5
    # without the skybox the same pixels carry no radiance
    sc2 = scenes.textured_scene()
    sc2.skybox = None
    r2 = lprt_amd.Renderer(W, H, abi.F32)
    r2.load_scene(sc2)
    r2.render(abi.default_settings(spp=1, svgf=0))
    assert (r2.read("INTENSITY0")[empty] == 0).all()


def test_texture_errors():
    r = lprt_amd.Renderer(8, 8, abi.F32)
    sc = scenes.textured_scene()
    sc.textures = sc.textures[:1]  # materials name texture 1, the skybox texture 2: neither uploaded
    with pytest.raises(lprt_amd.LprtError, match="texture handle"):
        r.set_rdresource(sc)
    with pytest.raises(lprt_amd.LprtError):
        r.upload_texture(np.zeros((0, 4, 4), dtype=np.uint8))


@pytest.mark.parametrize("prec", [abi.F32, abi.F16])
def test_sharded_sequence_with_history_exchange_equals_single_gpu(cornell, prec):
    """§8(f) rank 4: three frames on a 3-way row split.  With the per-frame exchange of halo-row history (what
    dist.ShardedFrame.exchange_history does between ranks) every frame equals the single-GPU frame bit for bit;
    without it the strips drift from frame 1 on."""
    import torch
    W, H, world, frames = 160, 300, 3, 3
    full = lprt_amd.Renderer(W, H, prec)
    full.load_scene(cornell)
    wins = [dist.window_rows(H, world, r) for r in range(world)]
    strips = [dist.strip_rows(H, world, r) for r in range(world)]

    def make():
        rs = []
        for wb, we in wins:
            r = lprt_amd.Renderer(W, H, prec, row_begin=wb, row_end=we)
            r.load_scene(cornell)
            rs.append(r)
        return rs

    with_x, without_x = make(), make()
    row_bytes = with_x[0].history_row_bytes()
    assert row_bytes == W * (65 if prec == abi.F32 else 33)
    plan = dist.history_transfers(H, world)
    drift = False
    for k in range(frames):
        fs = abi.default_settings(spp=1)
        full.set_scene(cornell)
        full.render(fs)
        want = full.read("OUTPUT")
        for rs in (with_x, without_x):
            for r in rs:
                r.set_scene(cornell)
                r.render(fs)
        for src, dst, b, e in plan:  # device-to-device here; over RCCL p2p between processes in dist.py
            buf = torch.empty((e - b) * row_bytes, dtype=torch.uint8, device="cuda")
            with_x[src].export_history_rows(b, e - b, buf.data_ptr())
            with_x[src].synchronize()
            with_x[dst].import_history_rows(b, e - b, buf.data_ptr())
            with_x[dst].synchronize()
        got = dist.assemble_strips([r.read("OUTPUT")[b - wb:e - wb] for r, (wb, we), (b, e) in zip(with_x, wins, strips)], H, world)
        assert np.array_equal(got, want, equal_nan=True), f"frame {k}"  # one wall-edge pixel is NaN in the reference semantics too
        other = dist.assemble_strips([r.read("OUTPUT")[b - wb:e - wb] for r, (wb, we), (b, e) in zip(without_x, wins, strips)], H, world)
        drift = drift or not np.array_equal(other, want, equal_nan=True)
    if prec == abi.F32:  # (fp16 rounding can absorb the last-bit differences of the truncated stencils)
        assert drift, "the exchange is supposed to matter from frame 1 on"
    with pytest.raises(lprt_amd.LprtError, match="row window"):
        with_x[0].export_history_rows(wins[0][1], 1, 0xdead0000)


@pytest.mark.parametrize("prec", [abi.F32, abi.F16])
@pytest.mark.parametrize("spp,gi", [(2, True), (1, False)])
def test_strips_that_import_their_halo_samples_equal_single_gpu(cornell, prec, spp, gi):
    """Sharding without redundant rays: every rank runs the per-sample stages on the rows it owns only
    (lprt_set_sample_rows) and, between LPRT_PHASE_SAMPLE and LPRT_PHASE_RESOLVE, imports the sample means of its halo
    rows (LPRT_HIST_SAMPLES) and last frame's history from their owners.  Frames equal the single-GPU frames bit for bit,
    and the halo rows cast no rays."""
    import torch
    W, H, world, frames = 160, 300, 3, 3
    full = lprt_amd.Renderer(W, H, prec)
    full.load_scene(cornell)
    wins = [dist.window_rows(H, world, r) for r in range(world)]
    strips = [dist.strip_rows(H, world, r) for r in range(world)]
    ranks = []
    for (wb, we), (b, e) in zip(wins, strips):
        r = lprt_amd.Renderer(W, H, prec, row_begin=wb, row_end=we)
        r.load_scene(cornell)
        r.set_sample_rows(b, e - b)
        r.enable_counters(True)
        ranks.append(r)
    full.enable_counters(True)
    groups = abi.HIST_ALL | abi.HIST_SAMPLES
    row_bytes = ranks[0].history_row_bytes(groups)
    assert row_bytes == W * ((65 + 32) if prec == abi.F32 else (33 + 16))
    plan = dist.history_transfers(H, world)
    for k in range(frames):
        fs = abi.default_settings(spp=spp, gi_on=gi)
        full.set_scene(cornell)
        full.render(fs)
        want = full.read("OUTPUT")
        for r in ranks:
            r.set_scene(cornell)
            r.render_phase(fs, abi.PHASE_SAMPLE)
        for src, dst, b, e in plan:
            buf = torch.empty((e - b) * row_bytes, dtype=torch.uint8, device="cuda")
            ranks[src].export_history_rows(b, e - b, buf.data_ptr(), groups)
            ranks[src].synchronize()
            ranks[dst].import_history_rows(b, e - b, buf.data_ptr(), groups)
            ranks[dst].synchronize()
        for r in ranks:
            r.render_phase(fs, abi.PHASE_RESOLVE)
        got = dist.assemble_strips([r.read("OUTPUT")[b - wb:e - wb] for r, (wb, we), (b, e) in zip(ranks, wins, strips)], H, world)
        assert np.array_equal(got, want, equal_nan=True), f"frame {k}"
    # shadow + GI rays of all ranks together = those of the full frame; only the primary rays cover the halo as well
    for st in ("trace_di1", "trace_gi", "trace_di2"):
        assert sum(r.stage_counters(st)["rays"] for r in ranks) == full.stage_counters(st)["rays"], st
    assert sum(r.stage_counters("primary")["rays"] for r in ranks) == W * sum(we - wb for wb, we in wins) * frames


@pytest.mark.parametrize("prec", [abi.F32, abi.F16])
def test_sharded_frames_under_motion_equal_single_gpu(cornell, prec):
    """§8(f) rank 4, the case the halo and the history exchange exist for: objects move and the camera dollies and climbs,
    so the reprojected history taps (generate_temporal_map, cuda.hpp:2265-2416) of pixels next to a strip boundary land in
    the NEIGHBOUR's rows.  Three ranks, five frames, halo-untraced path (sample means + history imported from the owners
    between the two render phases, as dist.ShardedFrame.render does): every frame equals the single-context frame bit for
    bit.  The halo is 68 rows + the vertical motion per frame (dist.window_rows)."""
    import torch
    W, H, world, frames, halo = 160, 300, 3, 5, dist.HALO_ROWS + 12
    full = lprt_amd.Renderer(W, H, prec)
    full.load_scene(cornell)
    wins = [dist.window_rows(H, world, r, halo) for r in range(world)]
    strips = [dist.strip_rows(H, world, r) for r in range(world)]
    ranks = []
    for (wb, we), (b, e) in zip(wins, strips):
        r = lprt_amd.Renderer(W, H, prec, row_begin=wb, row_end=we)
        r.load_scene(cornell)
        r.set_sample_rows(b, e - b)
        ranks.append(r)
    groups = abi.HIST_ALL | abi.HIST_SAMPLES
    row_bytes = ranks[0].history_row_bytes(groups)
    plan = dist.history_transfers(H, world, halo)
    crossings, prev_w2c = 0, None
    for k in range(frames):
        sc = scenes.moving_cornell(k, cornell, climb=0.06)
        fs = abi.default_settings(spp=2)
        full.set_scene(sc)
        full.render(fs)
        want = full.read("OUTPUT")
        for r in ranks:
            r.set_scene(sc)
            r.render_phase(fs, abi.PHASE_SAMPLE)
        for src, dst, b, e in plan:
            buf = torch.empty((e - b) * row_bytes, dtype=torch.uint8, device="cuda")
            ranks[src].export_history_rows(b, e - b, buf.data_ptr(), groups)
            ranks[src].synchronize()
            ranks[dst].import_history_rows(b, e - b, buf.data_ptr(), groups)
            ranks[dst].synchronize()
        for r in ranks:
            r.render_phase(fs, abi.PHASE_RESOLVE)
        got = dist.assemble_strips([r.read("OUTPUT")[b - wb:e - wb] for r, (wb, we), (b, e) in zip(ranks, wins, strips)], H, world)
        assert np.array_equal(got, want, equal_nan=True), f"frame {k}"
        # the premise: static geometry (the walls) next to a strip boundary reprojects into the other strip's rows
        if prev_w2c is not None:
            pos = full.read("G_POSITION").astype(np.float64)
            pos[..., 3] = 1.0
            clip = pos @ prev_w2c.T
            prev_row = np.floor((1.0 + clip[..., 1] / clip[..., 3]) / 2.0 * H - 0.5)
            row = np.arange(H)[:, None] * np.ones((1, W))
            walls = (full.read("G_COLOR")[..., 3] > 0) & (full.read("G_OBJID")[..., 0] < 5)
            for b, _ in strips[1:]:
                crossings += int((walls & (row >= b) & (prev_row < b)).sum() + (walls & (row < b) & (prev_row >= b)).sum())
        prev_w2c = sc.transform_W2C(W, H).astype(np.float64)
    assert crossings > 200, crossings
    # and the motion matters: the same ranks with the default 68-row halo are NOT enough at this climb rate ... only if
    # a tap actually leaves the window; this sequence moves < 12 rows per frame, which the widened halo absorbs


def test_c5_window_of_a_4k_16spp_fp16_frame_equals_the_full_frame(sponza):
    """BASELINE configs[4] at full size (atrium, 3840x2160, 16 spp, packed fp16, 8-way row split): rank 3 of 8 renders its
    406-row window the way bench.py --gpus 8 does -- primary rays over the window, the 16 samples over its 270 owned rows
    only, the halo rows' sample means imported from the two neighbour contexts that own them -- and its owned rows of the
    final image equal rows 810..1079 of the single-context 4K frame bit for bit.  Size-independent property (no oracle at
    this size): a strip is not an approximation of the frame."""
    import torch
    W, H, world, spp, prec = 3840, 2160, 8, 16, abi.F16
    fs = abi.default_settings(spp=spp)
    with lprt_amd.Renderer(W, H, prec) as full:
        full.load_scene(sponza)
        full.enable_counters(True)
        full.render(fs)
        b3, e3 = dist.strip_rows(H, world, 3)
        want = full.read("OUTPUT")[b3:e3].copy()
        full_rays = {st: full.stage_counters(st)["rays"] for st in ("primary", "trace_gi")}
    assert (b3, e3) == (810, 1080) and np.isfinite(want).all() and want[..., 3].min() == 1.0
    ranks = {}
    for rank in (2, 3, 4):
        wb, we = dist.window_rows(H, world, rank)
        b, e = dist.strip_rows(H, world, rank)
        r = lprt_amd.Renderer(W, H, prec, row_begin=wb, row_end=we)
        r.load_scene(sponza)
        r.set_sample_rows(b, e - b)
        r.enable_counters(True)
        r.render_phase(fs, abi.PHASE_SAMPLE)
        ranks[rank] = r
    assert dist.window_rows(H, world, 3) == (742, 1148)
    row_bytes = ranks[3].history_row_bytes(abi.HIST_SAMPLES)
    assert row_bytes == W * 16
    for src, dst, b, e in dist.history_transfers(H, world):
        if dst != 3:
            continue
        assert src in (2, 4) and e - b == dist.HALO_ROWS
        buf = torch.empty((e - b) * row_bytes, dtype=torch.uint8, device="cuda")
        ranks[src].export_history_rows(b, e - b, buf.data_ptr(), abi.HIST_SAMPLES)
        ranks[src].synchronize()
        ranks[3].import_history_rows(b, e - b, buf.data_ptr(), abi.HIST_SAMPLES)
        ranks[3].synchronize()
    ranks[3].render_phase(fs, abi.PHASE_RESOLVE)
    got = ranks[3].read("OUTPUT")[b3 - 742:e3 - 742]
    assert np.array_equal(got, want)
    # the window traced its primary rays, the samples ran on the owned rows only
    assert ranks[3].stage_counters("primary")["rays"] == W * (1148 - 742)
    assert 0 < ranks[3].stage_counters("trace_gi")["rays"] < full_rays["trace_gi"] / 6


def test_render_phases_and_sample_rows_are_checked(cornell):
    r = lprt_amd.Renderer(64, 64, abi.F32, row_begin=8, row_end=40)
    r.load_scene(cornell)
    fs = abi.default_settings(spp=1)
    with pytest.raises(lprt_amd.LprtError, match="out of order"):
        r.render_phase(fs, abi.PHASE_RESOLVE)
    r.render_phase(fs, abi.PHASE_SAMPLE)
    with pytest.raises(lprt_amd.LprtError, match="out of order"):
        r.render_phase(fs, abi.PHASE_SAMPLE)
    r.render_phase(fs, abi.PHASE_RESOLVE)
    with pytest.raises(lprt_amd.LprtError, match="unknown render phase"):
        r.render_phase(fs, 3)
    with pytest.raises(lprt_amd.LprtError, match="row window"):
        r.set_sample_rows(0, 16)
    with pytest.raises(lprt_amd.LprtError, match="row window"):
        r.set_sample_rows(32, 16)
    # first_row + n_rows wraps around in 32 bits: rejected, not "inside the window"
    with pytest.raises(lprt_amd.LprtError, match="row window"):
        r.set_sample_rows(10, 0xFFFFFFF7)
    with pytest.raises(lprt_amd.LprtError, match="row window"):
        r.export_history_rows(10, 0xFFFFFFF7, 0xdead0000)
    with pytest.raises(lprt_amd.LprtError, match="row window"):
        r.set_output_target(0xdead0000, 10, 0xFFFFFFF7)
    r.set_sample_rows(16, 8)
    r.set_sample_rows(0, 0)  # back to the whole window
    r.render(fs)


def test_options_out_array_sizes_and_null_pointers_are_checked(cornell):
    import ctypes as C
    with lprt_amd.Renderer(32, 32, abi.F32) as r:
        assert [r.get_option(n) for n in abi.OPTIONS] == [-1, -1, 1, 0, -1, 1, 0, -1, 0, 12, 20, 0, 8, 0, 1, 0, -1]
        r.set_option("wide_walk", 1)
        assert r.get_option("wide_walk") == 1
        with pytest.raises(lprt_amd.LprtError, match="out of range"):
            r.set_option("axis_reject", -1)
        with pytest.raises(lprt_amd.LprtError, match="out of range"):
            r.set_option("wide_walk", 2)
        assert r.L.lprt_set_option(r.h, 99, 0) == -1
        # sized out-arrays: a wrong length is an error, nothing is written
        ms = (C.c_float * 4)(1, 2, 3, 4)
        assert r.L.lprt_get_stage_times(r.h, ms, None, 4) == -1 and list(ms) == [1, 2, 3, 4]
        out = np.full(12, 7, dtype=np.uint32)
        assert r.L.lprt_debug_rng_states(r.h, 1, 0, 4, out.ctypes.data_as(C.c_void_p), out.size) == -1 and (out == 7).all()
        r.set_rdresource(cornell)
        assert r.L.lprt_get_blas_info(r.h, 0, None, None, None) == -1
        assert b"null" in r.L.lprt_last_error(r.h)
    assert r.h is None  # closed by the context manager


# moving Cornell, half precision, DEFAULT slab mode: whole-buffer relative L2 (all pixels) recorded in round 3
# (gpurun_out/r3_probe1.log part C; measured maxima over the four frames in brackets).  The pre-denoise GI planes of frame 0
# exceed the north star's 5e-2 in this mode and ONLY in this mode: the f32 slab test admits rays to boxes the reference's
# half test culls (turned boxes: 6-10 primary hits per frame differ), and one sample of bounce radiance is a few bright
# pixels.  In reference-slab mode every plane of every frame is within 5e-2 (asserted below).
MOVING_F16_CEILING = {"INTENSITY0": 0.004, "INTENSITY1": 0.11, "RAW_COLOR": 0.004, "RAW_GI_COLORED": 0.09, "RAW_GI_WHITE": 0.08,
                      "MUL_GI_COLORED": 0.07, "MUL_GI_WHITE": 0.06, "COLOR_INPROGRESS": 0.004, "OUTPUT": 0.012}
#                      [0.0018, 0.079, 0.0018, 0.063, 0.056, 0.050, 0.044, 0.0017, 0.0056]


@pytest.mark.parametrize("prec,slab", PREC_SLAB, ids=PREC_SLAB_IDS)
def test_moving_objects_and_camera_reproject_like_the_oracle(cornell, prec, slab):
    """Temporal maps under motion (cuda.hpp:2265-2416): the tall box turns, the short box slides and the camera
    dollies between frames, so reprojection goes through last-frame L2W / W2C, history taps move and some are
    rejected on object-id mismatch.  Single precision and half precision with the reference's slab arithmetic: EVERY
    radiance plane of every frame within the north star's tolerance over all pixels.  Half precision with the default
    (f32) slab parameters: ray by ray, plus recorded whole-buffer ceilings."""
    import copy
    W, H = 160, 96
    o = Oracle(W, H, prec)
    o.load_scene(cornell)
    r = renderer(W, H, prec, slab)
    r.load_scene(cornell)
    ray_by_ray = prec == abi.F16 and slab == 0
    if ray_by_ray:
        probe_g, probe_o = renderer(W, H, prec, slab), Oracle(W, H, prec)
        probe_g.load_scene(cornell)
        probe_o.load_scene(cornell)
    for k in range(4):
        sc = scenes.moving_cornell(k, cornell)
        fs = abi.default_settings(spp=1)
        o.set_scene(sc)
        o.render_frame(fs)
        r.set_scene(sc)
        r.render(fs)
        if prec == abi.F32:
            assert np.array_equal(r.read("G_OBJID"), o.read("G_OBJID")), k
        if not ray_by_ray:
            for name in RADIANCE:
                assert rel_l2(r.read(name), o.read(name)) < TOL[prec], (k, name, rel_l2(r.read(name), o.read(name)))
        else:
            # a few rays per frame find something else than in the oracle (its half slab test loses boxes the ray does
            # pass; ties at half resolution of t).  Named ray by ray, bounded, everything else held tight;
            # disputed closest hits adjudicated by exhaustive search; every plane under its recorded ceiling.
            c = replay.classify(r, o, sc, prec, probe_g, probe_o, frame=k, spp=1)
            assert c["any"].mean() < 5e-3, (k, int(c["any"].sum()))
            j = replay.adjudicate_gi(c, probe_o)
            assert j["gpu_lost"] <= max(1, j["oracle_lost"]) and j["neither"] == 0, (k, j)
            assert_planes_ray_by_ray(r, o, c, prec, kept_l2=1e-4, kept_px=1e-2, max_outliers=0)
            for name in RADIANCE:
                lim = min(MOVING_F16_CEILING[name], TOL[prec]) if name in ("COLOR_INPROGRESS", "OUTPUT") else MOVING_F16_CEILING[name]
                assert rel_l2(r.read(name), o.read(name)) <= lim, (k, name, rel_l2(r.read(name), o.read(name)))
    # the sequence is not the static one: history was actually reprojected
    still = renderer(W, H, prec, slab)
    still.load_scene(cornell)
    for k in range(4):
        still.set_scene(cornell)
        still.render(abi.default_settings(spp=1))
    assert rel_l2(still.read("OUTPUT"), r.read("OUTPUT")) > 1e-2


def test_instances_sharing_a_mesh_reproject_like_the_oracle():
    """copy_last_frame_transform (cuda.hpp:2253-2263) keeps ONE last-frame transform per VAO: instances that share a mesh
    write the same slot (a race in the reference; the last instance in index order in the oracle's loop and in
    k_copy_last_transform), so the temporal maps of all but one instance of a mesh reproject through another instance's
    transform.  Three frames of the 150-instance field: every radiance plane within tolerance of the oracle on frames that
    use history, and the same frames from two runs (the race made frame 1 differ from run to run)."""
    sc = scenes.instanced_field(150)
    W, H = 160, 90
    runs = []
    for _ in range(2):
        r, o = render_pair(sc, W, H, abi.F32, frames=3, spp=1)
        for name in RADIANCE:
            assert rel_l2(r.read(name), o.read(name)) < TOL[abi.F32], name
        runs.append(r.read("OUTPUT"))
    assert np.array_equal(runs[0], runs[1], equal_nan=True)


@pytest.mark.parametrize("prec,name", [(abi.F32, "f32"), (abi.F16, "f16")])
def test_cpp_host_over_compat_header_renders_the_same_frames(cornell, tmp_path, prec, name):
    """§8(f) rank 3: examples/compat_demo.cpp drives Renderer<DataT> / RDResource / RDScene / CUDARenderGIData of
    include/lprt_compat.hpp; two frames at 2 spp come out bit-identical to the Python binding's."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "compat_demo")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(root, "examples"), "compat_demo"], check=True)
    W, H = 128, 80
    blob = scenes.write_blob(cornell, W, H, str(tmp_path / "cornell.blob"))
    out = str(tmp_path / "out.raw")
    p = subprocess.run([exe, blob, str(W), str(H), "2", "2", name, out], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert "last BLAS: 2 triangles" in p.stdout
    got = np.fromfile(out, dtype=np.float32).reshape(H, W, 4)
    r = lprt_amd.Renderer(W, H, prec)
    r.load_scene(cornell)
    for _ in range(2):
        r.set_scene(cornell)
        r.render(abi.default_settings(spp=2))
    assert np.array_equal(got, r.read("OUTPUT"), equal_nan=True)


@pytest.mark.parametrize("prec", [abi.F32, abi.F16])
@pytest.mark.parametrize("W,H", [(67, 131), (161, 9)])
def test_odd_frame_sizes_hit_every_tile_edge(cornell, W, H, prec):
    """Widths / heights that are multiples of nothing: partial LDS tiles in k_temporal_color (32x8), k_wavelet (64x4
    lattice tiles, strides up to 16 > the frame height) and partial last workgroups everywhere; two frames."""
    r, o = render_pair(cornell, W, H, prec, frames=2, spp=2)
    for name in ("COLOR_INPROGRESS", "MUL_GI_COLORED", "MUL_GI_WHITE", "OUTPUT"):
        assert rel_l2(r.read(name), o.read(name)) < TOL[prec], name


def test_degenerate_scenes_render_without_faults(cornell):
    """No instances at all, no lights, an instance of an empty primitive: frames complete and stay finite/black where
    nothing is visible (empty inputs)."""
    import copy
    W, H = 64, 40
    # 1. geometry uploaded, but the scene holds no instance and no light
    empty = copy.copy(cornell)
    empty.objects, empty.aabbs, empty.lights = [], cornell.aabbs[:0], []
    r = lprt_amd.Renderer(W, H, abi.F32)
    r.set_rdresource(cornell)
    r.set_scene(empty)
    for _ in range(2):
        r.render(abi.default_settings(spp=2))
    out = r.read("OUTPUT")
    assert (r.read("G_COLOR")[..., 3] == 0).all() and np.isfinite(out).all() and (out[..., :3] == 0).all()
    assert len(r.trace(np.zeros(4, dtype=abi.RAY_DTYPE))) == 4
    # 2. Cornell without lights: only the emitter and GI light the frame
    dark = copy.copy(cornell)
    dark.lights = []
    o = Oracle(W, H, abi.F32)
    o.load_scene(dark)
    r2 = lprt_amd.Renderer(W, H, abi.F32)
    r2.load_scene(dark)
    o.render_frame(abi.default_settings(spp=1))
    r2.render(abi.default_settings(spp=1))
    assert rel_l2(r2.read("OUTPUT"), o.read("OUTPUT")) < 1e-3
    # 3. a primitive with no triangles next to ordinary ones
    sc = scenes.small_test_scene()
    sc.vaos = np.concatenate([sc.vaos, np.array([[0, 0, len(sc.indices), 0]], dtype=np.uint32)])
    sc.local_aabbs = np.concatenate([sc.local_aabbs, np.zeros((1, 6), dtype=np.float32)])
    sc.objects = sc.objects + [dict(L2W=np.eye(4, dtype=np.float32), W2L=np.eye(4, dtype=np.float32), materialid=0, objectid=len(sc.vaos) - 1)]
    sc.aabbs = np.concatenate([sc.aabbs, np.array([[-1, -1, -1, 1, 1, 1]], dtype=np.float32)])
    r3 = lprt_amd.Renderer(W, H, abi.F32)
    r3.load_scene(sc)
    r3.render(abi.default_settings(spp=1))
    ref = lprt_amd.Renderer(W, H, abi.F32)
    ref.load_scene(scenes.small_test_scene())
    ref.render(abi.default_settings(spp=1))
    assert np.array_equal(r3.read("OUTPUT"), ref.read("OUTPUT"), equal_nan=True)


@pytest.mark.parametrize("prec", [abi.F32, abi.F16])
def test_axis_rejections_never_change_a_frame(sponza, prec):
    """trace.hpp's stationary / overflowed-axis rejections only drop boxes a ray cannot reach: with them switched off
    (LPRT_OPT_AXIS_REJECT = 0: the reference's plain slab test, 2-5x slower in half precision at this scene's scale) the
    frames are bit-identical, in both precisions, over two frames of the 262 k-triangle atrium."""
    outs = []
    for off in (False, True):
        r = lprt_amd.Renderer(256, 144, prec)
        r.set_option("axis_reject", 0 if off else 1)
        r.load_scene(sponza)
        for _ in range(2):
            r.set_scene(sponza)
            r.render(abi.default_settings(spp=2))
        outs.append((r.read("OUTPUT"), r.read("G_OFFSET"), r.read("INTENSITY1")))
        r.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("prec", [abi.F32, abi.F16])
@pytest.mark.parametrize("scene_name", ["cornell", "textured"])
def test_untested_window_sums_of_the_temporal_filter_never_change_a_frame(scene_name, prec):
    """k_temporal_color sums its 9x9 window without the per-sample finite tests when a workgroup's staged tile holds no
    inf/NaN (in half with packed adds/multiplies, two channels per instruction).  Forcing the tested form
    (LPRT_OPT_SVGF_ALWAYS_TEST) must give bit-identical lobes and frames over three frames."""
    sc = scenes.SCENES[scene_name]()
    outs = []
    for forced in (False, True):
        r = lprt_amd.Renderer(200, 120, prec)
        r.set_option("svgf_always_test", int(forced))
        r.load_scene(sc)
        for _ in range(3):
            r.set_scene(sc)
            r.render(abi.default_settings(spp=2))
        outs.append((r.read("OUTPUT"), r.read("MUL_GI_COLORED"), r.read("MUL_GI_WHITE")))
        r.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("prec", [abi.F32, abi.F16])
def test_random_states_of_background_runs_are_never_read(prec, monkeypatch):
    """The last level of the RNG expansion skips the runs of 64 pixel indices without a covered pixel (rng.hip:
    k_rng_live_runs): nothing may read those states.  With every run expanded (LPRT_RNG_ALL_RUNS, read when the context is
    created) three moving Cornell frames -- a third of each is background -- must come out bit for bit the same, at a frame
    width that is not a multiple of 64 (runs straddle rows)."""
    cornell = scenes.cornell_box()
    outs = []
    for every_run in (False, True):
        if every_run:
            monkeypatch.setenv("LPRT_RNG_ALL_RUNS", "1")
        else:
            monkeypatch.delenv("LPRT_RNG_ALL_RUNS", raising=False)
        r = lprt_amd.Renderer(333, 187, prec)
        r.load_scene(cornell)
        for k in range(3):
            r.set_scene(scenes.moving_cornell(k, cornell))
            r.render(abi.default_settings(spp=2))
        outs.append(tuple(r.read(n) for n in ("OUTPUT", "COLOR_INPROGRESS", "INTENSITY0", "INTENSITY1", "RAW_GI_COLORED", "GI_DIRECTION")))
        r.close()
    assert (np.asarray(outs[0][0])[..., :3] == 0).all(axis=-1).mean() > 0.05  # (there IS background in these frames)
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)


def test_random_states_expanded_beside_the_blas_build_are_the_same_states(monkeypatch):
    """A frame that rebuilds its BLAS set expands its random states on the context's second stream, beside the build (api.hip:
    sample_phase).  With the serial order (LPRT_RNG_BESIDE_BUILD=0, read when the context is created) four rebuilding frames of
    the small test scene must come out bit for bit the same, pre-denoise planes included."""
    sc = scenes.SCENES["small_test"]()
    outs = []
    for beside in (True, False):
        if beside:
            monkeypatch.delenv("LPRT_RNG_BESIDE_BUILD", raising=False)
        else:
            monkeypatch.setenv("LPRT_RNG_BESIDE_BUILD", "0")
        r = lprt_amd.Renderer(320, 180, abi.F16)
        r.load_scene(sc)
        fs = abi.default_settings(spp=3)
        fs.rebuild_blas = 1
        for _ in range(4):
            r.set_scene(sc)
            r.render(fs)
        outs.append(tuple(r.read(n) for n in ("OUTPUT", "COLOR_INPROGRESS", "INTENSITY0", "INTENSITY1", "RAW_GI_COLORED", "GI_DIRECTION")))
        r.close()
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("scene_name", ["cornell", "small_test"])
def test_reusing_round0_shadow_rays_gives_the_same_frames(scene_name):
    """lprt_frame_settings.reuse_direct_light: the shadow rays of round 0 do not depend on the sample, so casting them for
    sample 0 only must not change a single bit of any frame (three frames at 3 spp), while fewer rays are cast."""
    sc = scenes.SCENES[scene_name]()
    outs, rays = [], []
    for reuse in (0, 1):
        r = lprt_amd.Renderer(160, 96, abi.F32)
        r.load_scene(sc)
        r.enable_counters(True)
        for _ in range(3):
            fs = abi.default_settings(spp=3)
            fs.reuse_direct_light = reuse
            r.set_scene(sc)
            r.render(fs)
        outs.append([r.read(n) for n in ("OUTPUT", "INTENSITY0", "INTENSITY1", "COLOR_INPROGRESS")])
        rays.append(r.counters()["rays"])
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)
    assert rays[1] < rays[0]
