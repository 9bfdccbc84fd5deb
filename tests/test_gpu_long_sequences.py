"""Long half-precision sequences: the reference's reprojection extrapolates at the left / top image border
(generate_temporal_map truncates the history position toward zero, cuda.hpp:2300-2306, 2347-2353); in half precision a
static pixel's position is off by up to W * 2^-11 pixels, so border pixels extrapolate their own history every frame and
after ~15 frames the filtered radiance there is negative, grows 1.3-1.5x per frame and pow(negative, 1/2.2) in the
output pass is NaN.  The oracle restates that faithfully; the HIP path must show the same thing at the same place and
time by default, and stay finite with LPRT_OPT_FLOOR_REPROJECTION = 1."""
import numpy as np
import pytest

import lprt_amd
from lprt_amd import abi
from oracle.binding import Oracle
import replay

pytestmark = pytest.mark.gpu
W, H = 480, 270


def nonfinite(x):
    return ~np.isfinite(x.read("OUTPUT")).all(axis=-1)


def test_half_border_extrapolation_appears_where_and_when_the_oracle_shows_it(cornell):
    frames = 20
    with Oracle(W, H, abi.F16) as o, lprt_amd.Renderer(W, H, abi.F16) as r:
        o.load_scene(cornell)
        r.load_scene(cornell)
        first = {"gpu": None, "oracle": None}
        sets = []
        for f in range(frames):
            fs = abi.default_settings(spp=1)
            o.set_scene(cornell); o.render_frame(fs)
            r.set_scene(cornell); r.render(fs)
            g, c = nonfinite(r), nonfinite(o)
            sets.append((g, c))
            if first["gpu"] is None and g.any():
                first["gpu"] = f
            if first["oracle"] is None and c.any():
                first["oracle"] = f
            if first["gpu"] is None and first["oracle"] is None:
                # before the instability shows, the negative-going lobe itself agrees closely
                a, b = r.read("MUL_GI_WHITE"), o.read("MUL_GI_WHITE")
                assert ((a[..., :3] < 0).any(axis=-1) == (b[..., :3] < 0).any(axis=-1)).mean() > 0.999
        assert first["oracle"] is not None, "the oracle no longer shows the border instability within 20 frames: rewrite this test"
        assert first["gpu"] is not None and abs(first["gpu"] - first["oracle"]) <= 1, first
        f0 = max(first["gpu"], first["oracle"])
        g, c = sets[f0]
        # same place: every NaN pixel of one side within 2 pixels of a NaN pixel of the other, all of them at the border
        assert not (g & ~replay.dilate(c, 2)).any() and not (c & ~replay.dilate(g, 2)).any()
        border = np.zeros((H, W), dtype=bool)
        border[:6, :] = border[:, :6] = True
        g, c = sets[-1]
        assert not (g & ~border).any() and not (c & ~border).any()
        assert 0 < g.sum() <= 64 and 0 < c.sum() <= 64


@pytest.mark.parametrize("prec", [abi.F32, abi.F16])
def test_floor_reprojection_keeps_long_sequences_finite_and_leaves_the_interior_alone(cornell, prec):
    with lprt_amd.Renderer(W, H, prec) as ref, lprt_amd.Renderer(W, H, prec) as fix:
        ref.load_scene(cornell)
        fix.load_scene(cornell)
        fix.set_option("floor_reprojection", 1)
        assert fix.get_option("floor_reprojection") == 1 and ref.get_option("floor_reprojection") == 0
        reach = 70  # 5 a-trous levels: 2 * (1 + 2 + 4 + 8 + 16) = 62 pixels per frame, + the 3x3 / bilinear taps
        for f in range(30):
            fs = abi.default_settings(spp=1)
            fix.set_scene(cornell); fix.render(fs)
            out = fix.read("OUTPUT")
            assert np.isfinite(out).all(), f
            assert (fix.read("COLOR_INPROGRESS")[..., :3] >= 0).all(), f
            assert (fix.read("MUL_GI_WHITE")[..., :3] >= 0).all() and (fix.read("MUL_GI_COLORED")[..., :3] >= 0).all(), f
            if f < 2:
                # only history positions left of / above the image change their taps: until their influence has travelled
                # (frame 1 is the first with history) everything further than the filters reach is bit-identical
                ref.set_scene(cornell); ref.render(fs)
                a, b = out[reach:-reach, reach:-reach], ref.read("OUTPUT")[reach:-reach, reach:-reach]
                assert np.array_equal(a, b), f
