"""Oracle texture unit + skybox (rt/rtrt/memory.hpp:334-369, rt/rtrt/cuda.hpp:1742-1855)."""
import numpy as np

import lprt_amd  # noqa: F401
from lprt_amd import abi, scenes
from oracle.binding import Oracle


def _numpy_fetch(img, srgb, uv):
    """Independent restatement in float64 (no fixed-point weights): the oracle must stay within the 1/512 weight step."""
    h, w = img.shape[:2]
    t = img.astype(np.float64) / 255.0
    if srgb:
        c = t[..., :3]
        t[..., :3] = np.where(c <= 0.04045, c / 12.92, ((c + 0.055) / 1.055) ** 2.4)
    out = []
    for u, v in uv.astype(np.float64):
        x, y = (u - np.floor(u)) * w - 0.5, (v - np.floor(v)) * h - 0.5
        i, j = int(np.floor(x)), int(np.floor(y))
        a, b = x - i, y - j
        f = lambda ii, jj: t[jj % h, ii % w]
        out.append((1 - a) * (1 - b) * f(i, j) + a * (1 - b) * f(i + 1, j) + (1 - a) * b * f(i, j + 1) + a * b * f(i + 1, j + 1))
    return np.array(out)


def test_oracle_texture_fetch_properties():
    o = Oracle(8, 8, abi.F32)
    img = scenes.checker_texture(16, 8, 3)
    lin, srg = o.upload_texture(img, False), o.upload_texture(img, True)
    assert (lin, srg) == (0, 1)
    centres = np.array([[(i + 0.5) / 16, (j + 0.5) / 8] for j in range(8) for i in range(16)], dtype=np.float32)
    assert np.array_equal(o.sample_texture(lin, centres), img.reshape(-1, 4).astype(np.float32) / np.float32(255))
    # wrap addressing: whole-number offsets change nothing (coordinates chosen exactly representable)
    assert np.array_equal(o.sample_texture(srg, centres + np.float32(1)), o.sample_texture(srg, centres - np.float32(2)))
    # halfway between two texel centres: the mean of the two
    mid = o.sample_texture(lin, np.array([[1.0 / 16, 0.5 / 8]], dtype=np.float32))[0]
    want = (img[0, 0].astype(np.float32) + img[0, 1].astype(np.float32)) / np.float32(255) / 2
    assert np.allclose(mid, want, atol=1e-6)
    # u = 0 blends the last and the first column
    edge = o.sample_texture(lin, np.array([[0.0, 0.5 / 8]], dtype=np.float32))[0]
    want = (img[0, 15].astype(np.float32) + img[0, 0].astype(np.float32)) / np.float32(255) / 2
    assert np.allclose(edge, want, atol=1e-6)
    # sRGB decode: known values of the IEC 61966-2-1 curve, alpha stays linear
    g = np.zeros((1, 1, 4), dtype=np.uint8)
    g[0, 0] = (0, 128, 255, 128)
    hg = o.upload_texture(g, True)
    px = o.sample_texture(hg, np.array([[0.3, 0.9]], dtype=np.float32))[0]
    assert np.allclose(px, [0.0, 0.21586050, 1.0, 128 / 255], atol=1e-6)
    rng = np.random.default_rng(0)
    uv = rng.uniform(-2, 2, size=(500, 2)).astype(np.float32)
    for handle, srgb in ((lin, False), (srg, True)):
        assert np.abs(o.sample_texture(handle, uv) - _numpy_fetch(img, srgb, uv)).max() < 2.5 / 512 + 1e-4


def test_oracle_textured_frame_and_skybox():
    sc = scenes.textured_scene()
    o = Oracle(96, 54, abi.F32)
    o.load_scene(sc)
    o.render_frame(abi.default_settings(spp=1, svgf=0))
    g = o.read("G_COLOR")
    empty = g[..., 3] == 0
    objid = o.read("G_OBJID")[..., 0]
    alb = o.read("ALBEDO")
    floor = (~empty) & (objid == 0)
    assert floor.sum() > 200 and len(np.unique(alb[floor].round(4), axis=0)) > 50   # textured: many albedo values
    plain = (~empty) & (objid == 2)
    assert len(np.unique(alb[plain], axis=0)) == 1                                   # untextured: the colour factor
    i0 = o.read("INTENSITY0")
    assert (i0[empty][:, :3] > 0).all()
    # the sky texture is a smooth gradient: exposure 1.5 bounds the radiance of empty pixels by 1.5
    assert i0[empty][:, :3].max() <= 1.5 + 1e-6
    # same scene, no skybox -> empty pixels stay black
    sc.skybox = None
    o2 = Oracle(96, 54, abi.F32)
    o2.load_scene(sc)
    o2.render_frame(abi.default_settings(spp=1, svgf=0))
    assert (o2.read("INTENSITY0")[empty] == 0).all()
    # and nothing else changes in the first bounce's direct light of covered pixels
    assert np.array_equal(o2.read("ALBEDO"), alb)


def test_gltf_base_colour_texture(tmp_path):
    """loader.cu:485-541: baseColorTexture -> uv_color = texCoord, texture decoded to RGBA8, flagged sRGB."""
    import base64
    import io
    import json
    from PIL import Image
    from lprt_amd import gltf
    sc = scenes.cornell_box()
    path = gltf.save_gltf(sc, str(tmp_path / "c.gltf"))
    doc = json.load(open(path))
    img = scenes.checker_texture(8, 4, 1)
    buf = io.BytesIO()
    Image.fromarray(img, "RGBA").save(buf, format="PNG")
    doc["images"] = [dict(uri="data:image/png;base64," + base64.b64encode(buf.getvalue()).decode())]
    doc["textures"] = [dict(source=0)]
    doc["materials"][1]["pbrMetallicRoughness"]["baseColorTexture"] = dict(index=0, texCoord=1)
    json.dump(doc, open(path, "w"))
    back = gltf.load_gltf(path)
    assert len(back.textures) == 1 and back.textures[0]["srgb"] and np.array_equal(back.textures[0]["rgba8"], img)
    assert back.materials[2]["uv_color"] == 1 and back.materials[2]["tex_index"] == 0
    mats = back.c_materials([7])
    assert mats[2].uv_color == 1 and mats[2].tex_color == 7 and mats[1].uv_color == 0xFFFFFFFF
