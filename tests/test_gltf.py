"""glTF reader/writer: the loader contract of rt/rtrt/loader.cu + rt/rtrt/hierarchy.hpp."""
import base64
import json
import os
import struct

import numpy as np
import pytest

import lprt_amd  # noqa: F401
from lprt_amd import abi, gltf, scenes


def _same_geometry(a, b):
    assert np.array_equal(a.indices, b.indices) and np.array_equal(a.vaos, b.vaos)
    for f in a.vertices.dtype.names:
        assert np.array_equal(a.vertices[f], b.vertices[f]), f
    assert np.array_equal(a.aabbs, b.aabbs)


@pytest.mark.parametrize("name", ["cornell", "sponza_like"])
def test_roundtrip_generated_scene(tmp_path, name):
    sc = scenes.SCENES[name]()
    path = gltf.save_gltf(sc, str(tmp_path / f"{name}.gltf"))
    back = gltf.load_gltf(path)
    _same_geometry(sc, back)
    assert len(back.materials) == len(sc.materials) + 1  # default material in slot 0 (loader.cu:774-776)
    assert back.materials[0] == dict(color=(1, 1, 1), emission=(0, 0, 0), metallic=0.0, roughness=1.0, double_sided=True)
    for m0, m1 in zip(sc.materials, back.materials[1:]):
        assert np.allclose(m0.get("color", (1, 1, 1)), m1["color"]) and np.allclose(m0.get("emission", (0, 0, 0)), m1["emission"])
        assert m0.get("metallic", 0.0) == m1["metallic"] and m0.get("roughness", 1.0) == m1["roughness"]
        assert bool(m0.get("double_sided", True)) == m1["double_sided"]
    for o0, o1 in zip(sc.objects, back.objects):
        assert np.array_equal(o0["L2W"], o1["L2W"]) and np.array_equal(o0["W2L"], o1["W2L"])
        assert o1["materialid"] == o0["materialid"] + 1 and o1["objectid"] == o0["objectid"]
    assert np.allclose(sc.camera_W2V, back.camera_W2V, atol=1e-6) and back.fovy == pytest.approx(sc.fovy)
    for l0, l1 in zip(sc.lights, back.lights):
        assert l0["type"] == l1["type"] and np.allclose(l0["intensity"], l1["intensity"], rtol=1e-6)
        if l0["type"] == abi.LIGHT_DIRECTIONAL:
            d = np.asarray(l0["direction"], dtype=np.float64)
            assert np.allclose(d / np.linalg.norm(d), l1["direction"], atol=1e-6)
        else:
            assert np.allclose(l0["position"], l1["position"], atol=1e-6)


def _tiny_doc(with_tangent=False, stride=0, extra_nodes=None, material=None):
    """One triangle + one quad in one mesh (two primitives), data: URI buffer, optional interleaving."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], dtype=np.float32)
    nrm = np.array([[0, 0, 1]] * 3 + [[0, 0.6, 0.8]] * 4, dtype=np.float32)
    blob = bytearray()
    views, accs = [], []

    def add(arr, ctype, atype, count=None):
        while len(blob) % 4:
            blob.append(0)
        views.append(dict(buffer=0, byteOffset=len(blob), byteLength=arr.nbytes))
        blob.extend(arr.tobytes())
        accs.append(dict(bufferView=len(views) - 1, componentType=ctype, count=count or len(arr), type=atype))
        return len(accs) - 1

    if stride:
        inter = np.zeros((3, stride // 4), dtype=np.float32)
        inter[:, 0:3], inter[:, 3:6] = pos[:3], nrm[:3]
        v = add(inter, 5126, "VEC3", 3)
        views[-1]["byteStride"] = stride
        accs.append(dict(bufferView=len(views) - 1, byteOffset=12, componentType=5126, count=3, type="VEC3"))
        p0, n0 = v, len(accs) - 1
    else:
        p0, n0 = add(pos[:3], 5126, "VEC3"), add(nrm[:3], 5126, "VEC3")
    i0 = add(np.array([0, 1, 2], dtype=np.uint8), 5121, "SCALAR")
    p1, n1 = add(pos[3:], 5126, "VEC3"), add(nrm[3:], 5126, "VEC3")
    i1 = add(np.array([0, 1, 2, 0, 2, 3], dtype=np.uint16), 5123, "SCALAR")
    prim0 = dict(attributes=dict(POSITION=p0, NORMAL=n0), indices=i0)
    prim1 = dict(attributes=dict(POSITION=p1, NORMAL=n1), indices=i1, material=0)
    if with_tangent:
        tan = np.array([[2, 0, 0, 1]] * 4, dtype=np.float32)
        prim1["attributes"]["TANGENT"] = add(tan, 5126, "VEC4")
    nodes = [dict(mesh=0, translation=[1, 2, 3], rotation=[0, 0.70710678, 0, 0.70710678], scale=[2, 2, 2], children=[1]),
             dict(mesh=0, translation=[0, 0, -1]),
             dict(camera=0, translation=[0, 0, 5])] + (extra_nodes or [])
    return dict(asset=dict(version="2.0"), scene=0, scenes=[dict(nodes=[0] + list(range(2, 3 + len(extra_nodes or []))))], nodes=nodes,
                meshes=[dict(primitives=[prim0, prim1])], materials=[material or dict(pbrMetallicRoughness=dict(baseColorFactor=[.1, .2, .3, 1]))],
                accessors=accs, bufferViews=views, cameras=[dict(type="perspective", perspective=dict(yfov=0.7, znear=0.05, zfar=50))],
                buffers=[dict(byteLength=len(blob), uri="data:application/octet-stream;base64," + base64.b64encode(bytes(blob)).decode())])


def _write(tmp_path, doc, name="t.gltf"):
    p = tmp_path / name
    p.write_text(json.dumps(doc))
    return str(p)


def test_loader_contract(tmp_path):
    sc = gltf.load_gltf(_write(tmp_path, _tiny_doc()))
    # two primitives -> two VAOs over one shared VBO/EBO, indices rebased (loader.cu:330-332)
    assert sc.vaos.tolist() == [[0, 0, 0, 3], [0, 0, 3, 6]]
    assert sc.indices.tolist() == [0, 1, 2, 3, 4, 5, 3, 5, 6]
    # node 0 and its child node 1 each instance both primitives (hierarchy.hpp:236-247)
    assert [(o["objectid"], o["materialid"]) for o in sc.objects] == [(0, 0), (1, 1), (0, 0), (1, 1)]
    # primitive without a material -> default material 0; glTF material 0 -> 1; glTF metallic default is 1, doubleSided false
    assert sc.materials[1]["metallic"] == 1.0 and sc.materials[1]["double_sided"] is False
    assert np.allclose(sc.materials[1]["color"], (.1, .2, .3))
    # T*R*S, rotation 90 degrees about +y: local +x -> world -z
    m = sc.objects[0]["L2W"]
    assert np.allclose(m @ np.array([1, 0, 0, 1], dtype=np.float32), [1, 2, 3 - 2, 1], atol=1e-5)
    child = sc.objects[2]["L2W"]
    assert np.allclose(child @ np.array([0, 0, 0, 1], dtype=np.float32), [1 - 2, 2, 3, 1], atol=1e-5)  # parent * child
    # synthesised tangents (loader.cu:376-389): n=(0,0,1) -> (0,-1,0); n=(0,.6,.8) -> (-.6,0,0)/|.| = (-1,0,0)
    assert np.allclose(sc.vertices["tangent"][0], [0, -1, 0]) and np.allclose(sc.vertices["tangent"][3], [-1, 0, 0])
    assert np.all(sc.vertices["color"] == 1) and np.all(sc.vertices["uv0"] == 0)
    assert sc.z_near == pytest.approx(0.05) and sc.z_far == 50 and sc.fovy == pytest.approx(0.7)
    assert np.allclose(sc.camera_L2W[:3, 3], [0, 0, 5])
    # world AABB of the first instance = transformed local box corners
    lo, hi = sc.aabbs[0][:3], sc.aabbs[0][3:]
    assert np.allclose(lo, [1, 2, 1], atol=1e-5) and np.allclose(hi, [1, 4, 3], atol=1e-5)


def test_tangent_normalised_and_interleaved(tmp_path):
    sc = gltf.load_gltf(_write(tmp_path, _tiny_doc(with_tangent=True, stride=32)))
    assert np.allclose(sc.vertices["tangent"][3:], [1, 0, 0])  # (2,0,0,1) normalised, w dropped (loader.cu:365-374)
    assert np.allclose(sc.vertices["position"][:3], [[0, 0, 0], [1, 0, 0], [0, 1, 0]])
    assert np.allclose(sc.vertices["normal"][:3], [0, 0, 1])


def test_lights_punctual(tmp_path):
    lights = [dict(type="spot", color=[1, .5, .25], intensity=4.0, spot={}), dict(type="directional", intensity=2.0), dict(type="point")]
    extra = [dict(translation=[1, 1, 1], extensions=dict(KHR_lights_punctual=dict(light=0))),
             dict(rotation=[-0.70710678, 0, 0, 0.70710678], extensions=dict(KHR_lights_punctual=dict(light=1))),
             dict(translation=[0, 3, 0], extensions=dict(KHR_lights_punctual=dict(light=2)))]
    doc = _tiny_doc(extra_nodes=extra)
    doc["extensions"] = dict(KHR_lights_punctual=dict(lights=lights))
    sc = gltf.load_gltf(_write(tmp_path, doc))
    spot, sun, point = sc.lights
    assert spot["type"] == abi.LIGHT_POINT and np.allclose(spot["intensity"], [4, 2, 1]) and np.allclose(spot["position"], [1, 1, 1])
    assert sun["type"] == abi.LIGHT_DIRECTIONAL and np.allclose(sun["direction"], [0, -1, 0], atol=1e-6) and np.allclose(sun["intensity"], 2)
    assert point["type"] == abi.LIGHT_POINT and np.allclose(point["intensity"], 1) and np.allclose(point["position"], [0, 3, 0])


def test_glb_and_errors(tmp_path):
    doc = _tiny_doc()
    raw = base64.b64decode(doc["buffers"][0]["uri"].split(",", 1)[1])
    del doc["buffers"][0]["uri"]
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    raw += b"\0" * (-len(raw) % 4)
    glb = struct.pack("<III", 0x46546C67, 2, 12 + 8 + len(js) + 8 + len(raw)) + struct.pack("<II", len(js), 0x4E4F534A) + js + \
        struct.pack("<II", len(raw), 0x004E4942) + raw
    p = tmp_path / "t.glb"
    p.write_bytes(glb)
    a, b = gltf.load_gltf(str(p)), gltf.load_gltf(_write(tmp_path, _tiny_doc()))
    assert np.array_equal(a.vertices, b.vertices) and np.array_equal(a.indices, b.indices)
    with pytest.raises(gltf.GltfError):
        gltf.load_gltf(str(tmp_path / "scene.obj"))
    bad = _tiny_doc()
    del bad["meshes"][0]["primitives"][0]["attributes"]["NORMAL"]
    with pytest.raises(gltf.GltfError, match="missing normal"):
        gltf.load_gltf(_write(tmp_path, bad, "bad.gltf"))
    bad = _tiny_doc()
    bad["meshes"][0]["primitives"][0]["mode"] = 1
    with pytest.raises(gltf.GltfError, match="TRIANGLES"):
        gltf.load_gltf(_write(tmp_path, bad, "bad2.gltf"))
    bad = _tiny_doc()
    bad["nodes"] = bad["nodes"][:2]
    bad["scenes"][0]["nodes"] = [0]
    with pytest.raises(gltf.GltfError, match="no active camera"):
        gltf.load_gltf(_write(tmp_path, bad, "bad3.gltf"))


def _animated_doc():
    doc = _tiny_doc()
    raw = bytearray(base64.b64decode(doc["buffers"][0]["uri"].split(",", 1)[1]))

    def add(arr, atype):
        while len(raw) % 4:
            raw.append(0)
        doc["bufferViews"].append(dict(buffer=0, byteOffset=len(raw), byteLength=arr.nbytes))
        raw.extend(arr.tobytes())
        doc["accessors"].append(dict(bufferView=len(doc["bufferViews"]) - 1, componentType=5126, count=len(arr), type=atype))
        return len(doc["accessors"]) - 1

    t = add(np.array([0.0, 2.0], dtype=np.float32), "SCALAR")
    tr = add(np.array([[0, 0, -1], [0, 0, -3]], dtype=np.float32), "VEC3")
    t3 = add(np.array([0.0, 1.0, 4.0], dtype=np.float32), "SCALAR")
    h = np.sqrt(0.5)
    rot = add(np.array([[0, 0, 0, 1], [0, h, 0, h], [0, 1, 0, 0]], dtype=np.float32), "VEC4")
    t1 = add(np.array([0.5], dtype=np.float32), "SCALAR")
    sc1 = add(np.array([[3, 3, 3]], dtype=np.float32), "VEC3")
    doc["animations"] = [dict(samplers=[dict(input=t, output=tr), dict(input=t3, output=rot), dict(input=t1, output=sc1)],
                              channels=[dict(sampler=0, target=dict(node=1, path="translation")),
                                        dict(sampler=1, target=dict(node=2, path="rotation")),
                                        dict(sampler=2, target=dict(node=1, path="scale")),
                                        dict(sampler=0, target=dict(node=1, path="weights"))])]
    doc["buffers"][0] = dict(byteLength=len(raw), uri="data:application/octet-stream;base64," + base64.b64encode(bytes(raw)).decode())
    return doc


def test_animation_sampling_and_pose(tmp_path):
    """load_animations (loader.cu:692-733) + Sampler::sample / apply_animation (hierarchy.hpp:81-143)."""
    sc = gltf.load_gltf(_write(tmp_path, _animated_doc()))
    assert sc.animated_nodes == [1, 2]
    parent = sc.objects[0]["L2W"].astype(np.float64)
    # loading does not apply the animation: the child sits at its node translation (0, 0, -1)
    assert np.allclose(sc.objects[2]["L2W"], parent @ np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, -1], [0, 0, 0, 1.0]]), atol=1e-6)
    at1 = sc.pose(1.0)
    # child node 1: translation halfway (0,0,-2); the single-key scale channel gives (3,3,3) at any time
    want = parent @ np.array([[3, 0, 0, 0], [0, 3, 0, 0], [0, 0, 3, -2], [0, 0, 0, 1.0]])
    assert np.allclose(at1.objects[2]["L2W"], want, atol=1e-5) and np.allclose(at1.objects[3]["L2W"], want, atol=1e-5)
    assert np.array_equal(at1.objects[0]["L2W"], sc.objects[0]["L2W"])                        # the parent is not animated
    assert np.allclose(at1.objects[2]["W2L"].astype(np.float64) @ at1.objects[2]["L2W"], np.eye(4), atol=1e-5)
    assert not np.array_equal(at1.aabbs[2], sc.aabbs[2]) and np.array_equal(at1.aabbs[0], sc.aabbs[0])
    # time wraps past the last key: 2.5 s -> 0.5 s -> z = -1.5
    assert np.allclose(sc.pose(2.5).objects[2]["L2W"][:3, 3], (parent @ [0, 0, -1.5, 1])[:3], atol=1e-5)
    # camera node 2: rotation key 1 at t = 1 is a quarter turn about +y: the camera looks down -x
    cam = sc.pose(1.0)
    assert np.allclose(cam.camera_L2W[:3, 3], [0, 0, 5], atol=1e-6)
    assert np.allclose(cam.camera_L2W @ np.array([0, 0, -1, 0], dtype=np.float32), [-1, 0, 0, 0], atol=1e-5)
    # between keys the quaternion is lerped WITHOUT renormalising and handed to angle/axis as is (glm::lerp, :115)
    half = sc.pose(0.5)
    q = 0.5 * np.array([0, 0, 0, 1.0]) + 0.5 * np.array([0, np.sqrt(0.5), 0, np.sqrt(0.5)])
    angle = 2 * np.arcsin(np.linalg.norm(q[:3])) if abs(q[3]) > np.cos(0.5) else 2 * np.arccos(q[3])
    fwd = half.camera_L2W @ np.array([0, 0, -1, 0], dtype=np.float32)
    assert np.allclose(fwd[:3], [-np.sin(angle), 0, -np.cos(angle)], atol=1e-5)
    assert abs(angle - np.pi / 4) > 1e-3                                                      # not the slerp angle
    # the geometry arrays are shared, not rebuilt
    assert at1.vertices is sc.vertices and at1.indices is sc.indices
