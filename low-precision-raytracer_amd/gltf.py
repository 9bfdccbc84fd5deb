"""Minimal glTF 2.0 reader/writer for the harness.

`load_gltf` reproduces the OUTPUT CONTRACT of the reference's loader (rt/rtrt/loader.cu:298-442 load_mesh, :485-541
load_material, :544-690 build_object_hierarchy, :737-828 load_gltf2) and of hierarchy::Scene::build_rendering_scene
(rt/rtrt/hierarchy.hpp:222-290) as the flat arrays of `scenes.SceneData`:

  * one shared VBO + EBO, indices rebased (loader.cu:330-332); one VAO per primitive, its own BLAS (:434);
  * NORMAL required, TANGENT normalised or synthesised (:365-389), missing uv -> 0, missing colour -> 1 (:392-417);
  * material 0 is the default white rough dielectric and glTF material k becomes k+1 (:320, :774-776);
  * node TRS or matrix, T*R*S down the tree (hierarchy.hpp:145-149, 240); world AABB from the 8 transformed corners;
  * KHR_lights_punctual: intensity = colour * intensity, spot lights are loaded as POINT (loader.cu:596-619), light
    direction = normalize(L2W*(0,0,-1,0)), position = L2W[3].xyz (hierarchy.hpp:250-254);
  * first perspective camera found depth-first (hierarchy.hpp:203-214);
  * node animations (translation / rotation / scale samplers, loader.cu:692-733) are kept: `scene.pose(t)` applies
    Object::apply_animation (hierarchy.hpp:81-143) and returns the RDScene of that instant.

Supported: .gltf (external .bin or data: URIs) and .glb; float/u8/u16/u32 accessors with byteStride; TRIANGLES only;
base-colour textures (PNG/JPEG through Pillow) are returned as RGBA8 arrays in `SceneData.textures`.
`save_gltf` writes a SceneData back as .gltf + .bin with one bufferView per accessor (the reference's load_data takes
its element count from bufferView.byteLength / stride).
"""
import base64
import json
import math
import os
import struct

import numpy as np

from . import abi, scenes

_COMP = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
_NCOMP = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


class GltfError(ValueError):
    pass


def _read_container(path):
    ext = os.path.splitext(path)[1].lower()
    base = os.path.dirname(os.path.abspath(path))
    if ext == ".glb":
        raw = open(path, "rb").read()
        magic, version, length = struct.unpack_from("<III", raw, 0)
        if magic != 0x46546C67 or version != 2:
            raise GltfError("not a glTF 2.0 binary")
        off, doc, bin_chunk = 12, None, None
        while off < length:
            clen, ctype = struct.unpack_from("<II", raw, off)
            body = raw[off + 8:off + 8 + clen]
            if ctype == 0x4E4F534A:
                doc = json.loads(body.decode("utf-8"))
            elif ctype == 0x004E4942:
                bin_chunk = body
            off += 8 + clen
        return doc, base, bin_chunk
    if ext == ".gltf":
        return json.load(open(path, "r", encoding="utf-8")), base, None
    raise GltfError("the extension of a glTF2 file should be .glb or .gltf")  # loader.cu:760


def _buffers(doc, base, bin_chunk):
    out = []
    for i, b in enumerate(doc.get("buffers", [])):
        uri = b.get("uri")
        if uri is None:
            if bin_chunk is None:
                raise GltfError("buffer without uri outside a .glb")
            out.append(bin_chunk)
        elif uri.startswith("data:"):
            out.append(base64.b64decode(uri.split(",", 1)[1]))
        else:
            out.append(open(os.path.join(base, uri), "rb").read())
    return out


def _accessor(doc, bufs, index):
    if index is None or index < 0:
        raise GltfError("gltf/glb file is broken: accessor index < 0")  # loader.cu:94
    acc = doc["accessors"][index]
    if "sparse" in acc:
        raise GltfError("mesh uses sparse accessor, which involves an unimplemented feature")  # loader.cu:102
    view = doc["bufferViews"][acc["bufferView"]]
    dt = np.dtype(_COMP[acc["componentType"]])
    nc = _NCOMP[acc["type"]]
    stride = view.get("byteStride") or dt.itemsize * nc
    start = view.get("byteOffset", 0) + acc.get("byteOffset", 0)
    count = acc["count"]
    raw = np.frombuffer(bufs[view["buffer"]], dtype=np.uint8, count=(count - 1) * stride + dt.itemsize * nc, offset=start)
    arr = np.lib.stride_tricks.as_strided(raw, shape=(count, dt.itemsize * nc), strides=(stride, 1)).copy()
    arr = arr.view(dt).reshape(count, nc)
    if acc.get("normalized") and dt.kind in "ui":
        arr = arr.astype(np.float32) / np.float32(np.iinfo(dt).max)
    return arr


def _quat_to_mat(q):
    x, y, z, w = (float(v) for v in q)
    n = math.sqrt(x * x + y * y + z * z + w * w) or 1.0
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 0],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w), 0],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y), 0],
                     [0, 0, 0, 1]], dtype=np.float64)


def _sample(track, time, default):
    """Sampler<T>::sample(time, mod=true, default) (hierarchy.hpp:81-113): one key -> that key, no keys -> the node's
    own value, past the last key the time wraps (fmodf), linear blend between the bracketing keys; quaternions are
    blended component-wise WITHOUT renormalising (glm::lerp).  Stateless: the cursor is found from the time."""
    if track is None or len(track[0]) == 0:
        return np.asarray(default, dtype=np.float64)
    times, values = track
    if len(times) == 1:
        return values[0].astype(np.float64)
    t = np.float32(time)
    if t >= times[-1]:
        t = np.float32(np.fmod(t, times[-1]))
    c = 0
    while c + 2 < len(times) and times[c + 1] < t:
        c += 1
    u = np.float32((t - times[c]) / (times[c + 1] - times[c]))
    return ((np.float32(1) - u) * values[c] + u * values[c + 1]).astype(np.float64)


def _quat_angle_axis(q):
    """glm::angle / glm::axis of a possibly non-unit quaternion (x, y, z, w), as Object::transform_matrix feeds them
    to glm::rotate (hierarchy.hpp:145-149).  glm is a submodule absent from the reference checkout: restated from the
    published 0.9.9 sources, unpinned."""
    x, y, z, w = (float(v) for v in q)
    if abs(w) > math.cos(0.5):
        a = 2.0 * math.asin(min(1.0, math.sqrt(x * x + y * y + z * z)))
        angle = 2.0 * math.pi - a if w < 0 else a
    else:
        angle = 2.0 * math.acos(w)
    t = 1.0 - w * w
    axis = np.array([0.0, 0.0, 1.0]) if t <= 0 else np.array([x, y, z]) / math.sqrt(t)
    return angle, axis


def _rotate(angle, axis):
    n = np.linalg.norm(axis)
    if n == 0:
        return np.eye(4)
    x, y, z = axis / n
    c, s_ = math.cos(angle), math.sin(angle)
    k = 1 - c
    m = np.eye(4)
    m[:3, :3] = [[c + x * x * k, x * y * k - z * s_, x * z * k + y * s_],
                 [y * x * k + z * s_, c + y * y * k, y * z * k - x * s_],
                 [z * x * k - y * s_, z * y * k + x * s_, c + z * z * k]]
    return m


def _decompose(m):
    """TRS of an affine matrix without skew (set_transform_matrix -> glm::decompose, hierarchy.hpp:151-157)."""
    t = m[:3, 3].copy()
    sc_ = np.linalg.norm(m[:3, :3], axis=0)
    if np.linalg.det(m[:3, :3]) < 0:
        sc_ = -sc_
    r = m[:3, :3] / sc_
    w = math.sqrt(max(0.0, 1 + r[0, 0] + r[1, 1] + r[2, 2])) / 2
    if w > 1e-6:
        q = np.array([(r[2, 1] - r[1, 2]) / (4 * w), (r[0, 2] - r[2, 0]) / (4 * w), (r[1, 0] - r[0, 1]) / (4 * w), w])
    else:  # half-turn: pick the largest diagonal
        i = int(np.argmax(np.diag(r)))
        j, k = (i + 1) % 3, (i + 2) % 3
        qi = math.sqrt(max(0.0, 1 + r[i, i] - r[j, j] - r[k, k])) / 2
        q = np.zeros(4)
        q[i], q[j], q[k], q[3] = qi, (r[j, i] + r[i, j]) / (4 * qi), (r[k, i] + r[i, k]) / (4 * qi), (r[k, j] - r[j, k]) / (4 * qi)
    return t, q, sc_


def _node_matrix(node, anim=None, time=None):
    if anim is None or time is None:
        if node.get("matrix"):
            return np.array(node["matrix"], dtype=np.float64).reshape(4, 4).T  # glTF matrices are column-major
        m = np.eye(4)
        if node.get("translation"):
            m = m @ scenes.translate(node["translation"])
        if node.get("rotation"):
            m = m @ _quat_to_mat(node["rotation"])
        if node.get("scale"):
            m = m @ scenes.scale(node["scale"])
        return m
    if node.get("matrix"):
        t0, q0, s0 = _decompose(np.array(node["matrix"], dtype=np.float64).reshape(4, 4).T)
    else:
        t0, q0, s0 = node.get("translation", [0, 0, 0]), node.get("rotation", [0, 0, 0, 1]), node.get("scale", [1, 1, 1])
    t = _sample(anim.get("translation"), time, t0)
    q = _sample(anim.get("rotation"), time, q0)
    sc_ = _sample(anim.get("scale"), time, s0)
    return scenes.translate(t) @ _rotate(*_quat_angle_axis(q)) @ scenes.scale(sc_)


def _load_animations(doc, bufs):
    """load_animations (loader.cu:692-733): every channel of every animation lands on its target node; later
    channels for the same node/path replace earlier ones; paths other than translation/rotation/scale are skipped."""
    out = {}
    for anim in doc.get("animations", []):
        for ch in anim["channels"]:
            path = ch["target"].get("path")
            if path not in ("translation", "rotation", "scale") or "node" not in ch["target"]:
                continue
            smp = anim["samplers"][ch["sampler"]]
            times = _accessor(doc, bufs, smp["input"]).reshape(-1).astype(np.float32)
            values = _accessor(doc, bufs, smp["output"]).astype(np.float32)
            out.setdefault(ch["target"]["node"], {})[path] = (times, values)
    return out


def _load_image(doc, base, bufs, index):
    try:
        from PIL import Image
    except Exception as e:  # pragma: no cover
        raise GltfError(f"Pillow is needed to decode textures: {e}")
    import io
    img = doc["images"][index]
    if "uri" in img:
        uri = img["uri"]
        data = base64.b64decode(uri.split(",", 1)[1]) if uri.startswith("data:") else open(os.path.join(base, uri), "rb").read()
    else:
        view = doc["bufferViews"][img["bufferView"]]
        data = bufs[view["buffer"]][view.get("byteOffset", 0):view.get("byteOffset", 0) + view["byteLength"]]
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGBA"), dtype=np.uint8)


def load_gltf(path, width_over_height=16 / 9):
    doc, base, bin_chunk = _read_container(path)
    bufs = _buffers(doc, base, bin_chunk)
    # ---- materials: default material 0, glTF material k -> k + 1 ---------------------------------------------
    materials = [dict(color=(1, 1, 1), emission=(0, 0, 0), metallic=0.0, roughness=1.0, double_sided=True)]
    textures, tex_of_image = [], {}
    for m in doc.get("materials", []):
        pbr = m.get("pbrMetallicRoughness", {})
        mat = dict(color=tuple(pbr.get("baseColorFactor", [1, 1, 1, 1])[:3]), emission=tuple(m.get("emissiveFactor", [0, 0, 0])),
                   metallic=float(pbr.get("metallicFactor", 1.0)), roughness=float(pbr.get("roughnessFactor", 1.0)),
                   double_sided=bool(m.get("doubleSided", False)))
        bct = pbr.get("baseColorTexture")
        if bct is not None and "textures" in doc:
            src = doc["textures"][bct["index"]].get("source")
            if src is not None:
                if src not in tex_of_image:
                    tex_of_image[src] = len(textures)
                    textures.append(dict(rgba8=_load_image(doc, base, bufs, src), srgb=True))
                mat["uv_color"] = int(bct.get("texCoord", 0))
                mat["tex_index"] = tex_of_image[src]
        materials.append(mat)
    # ---- meshes -> primitives (VAOs) ---------------------------------------------------------------------------
    prims, mesh_prims = [], []
    for mesh in doc.get("meshes", []):
        ids = []
        for p in mesh["primitives"]:
            if p.get("mode", 4) != 4:
                raise GltfError("mode of primitive other than TRIANGLES has not been supported yet")  # loader.cu:311
            at = p["attributes"]
            if "POSITION" not in at:
                raise GltfError("POSITION is not set for a mesh")
            if "NORMAL" not in at:
                raise GltfError("missing normal")  # loader.cu:359
            pos = _accessor(doc, bufs, at["POSITION"]).astype(np.float32)
            nrm = _accessor(doc, bufs, at["NORMAL"]).astype(np.float32)
            idx = _accessor(doc, bufs, p["indices"]).reshape(-1).astype(np.uint32) if "indices" in p else np.arange(len(pos), dtype=np.uint32)
            idx = idx[:len(idx) // 3 * 3]
            mesh_obj = scenes.Mesh(pos, nrm, idx)
            if "TEXCOORD_0" in at:
                mesh_obj.uv0 = _accessor(doc, bufs, at["TEXCOORD_0"]).astype(np.float32)
            if "COLOR_0" in at:
                mesh_obj.colors = _accessor(doc, bufs, at["COLOR_0"]).astype(np.float32)[:, :3]
            extra = {}
            if "TANGENT" in at:
                t = _accessor(doc, bufs, at["TANGENT"]).astype(np.float64)[:, :3]
                extra["tangent"] = (t / np.maximum(np.linalg.norm(t, axis=1, keepdims=True), 1e-30)).astype(np.float32)
            if "TEXCOORD_1" in at:
                extra["uv1"] = _accessor(doc, bufs, at["TEXCOORD_1"]).astype(np.float32)
            ids.append(len(prims))
            prims.append((mesh_obj, int(p.get("material", -1)) + 1, extra))
        mesh_prims.append(ids)
    # ---- scene graph + animations -------------------------------------------------------------------------------
    light_defs = doc.get("extensions", {}).get("KHR_lights_punctual", {}).get("lights", [])
    animations = _load_animations(doc, bufs)

    def walk_scene(time):
        instances, lights, cameras = [], [], []

        def walk(node_id, parent):
            node = doc["nodes"][node_id]
            m = parent @ _node_matrix(node, animations.get(node_id), time)
            lid = node.get("extensions", {}).get("KHR_lights_punctual", {}).get("light")
            kinds = (node.get("mesh") is not None) + (node.get("camera") is not None) + (lid is not None)
            if kinds > 1:
                raise GltfError("an object can only be one of mesh, camera or light")  # loader.cu:563
            if node.get("mesh") is not None:
                for vao in mesh_prims[node["mesh"]]:
                    instances.append((vao, prims[vao][1], m))
            elif node.get("camera") is not None:
                cam = doc["cameras"][node["camera"]]
                if cam.get("type") == "perspective":
                    cameras.append((m, cam["perspective"]))
            elif lid is not None:
                ld = light_defs[lid]
                inten = float(ld.get("intensity", 1.0))
                col = ld.get("color", [1, 1, 1])
                d = m @ np.array([0, 0, -1, 0.0])
                d = d[:3] / np.linalg.norm(d[:3])
                kind = ld["type"]
                if kind not in ("point", "spot", "directional"):
                    raise GltfError("unexpected light type")
                lights.append(dict(type=abi.LIGHT_DIRECTIONAL if kind == "directional" else abi.LIGHT_POINT,  # spot -> POINT
                                   intensity=tuple(float(c) * inten for c in col), direction=tuple(d), position=tuple(m[:3, 3])))
            for c in node.get("children", []):
                walk(c, m)

        for n in doc["scenes"][doc.get("scene", 0)]["nodes"]:
            walk(n, np.eye(4))
        if not cameras:
            raise GltfError("no active camera")  # hierarchy.hpp:285
        return instances, lights, cameras[0]

    def set_camera(sc, cam_m):
        # Camera::world_to_view (hierarchy.hpp:67-71): lookAt(M*origin, M*(0,0,-1,1), M*(0,1,0,0))
        eye = cam_m[:3, 3]
        fwd = (cam_m @ np.array([0, 0, -1, 0.0]))[:3]
        w2v = scenes.look_at(eye, eye + fwd, (cam_m @ np.array([0, 1, 0, 0.0]))[:3])
        sc.camera_W2V = w2v.astype(np.float32)
        sc.camera_L2W = np.linalg.inv(w2v).astype(np.float32)

    instances, lights, (cam_m, persp) = walk_scene(None)
    eye = cam_m[:3, 3]
    fwd = (cam_m @ np.array([0, 0, -1, 0.0]))[:3]
    meshes = [p[0] for p in prims]
    sc = scenes.assemble(os.path.splitext(os.path.basename(path))[0], meshes, materials, instances, lights, cam_eye=tuple(eye),
                         cam_target=tuple(eye + fwd), fovy=float(persp["yfov"]), meta=dict(source=path))
    sc.z_near, sc.z_far = float(persp.get("znear", 0.1)), float(persp.get("zfar", 100.0))
    sc.textures = textures
    # attributes assemble() does not synthesise
    vo = 0
    for mesh_obj, _, extra in prims:
        n = len(mesh_obj.positions)
        if "tangent" in extra:
            sc.vertices["tangent"][vo:vo + n] = extra["tangent"]
        if "uv1" in extra:
            sc.vertices["uv1"][vo:vo + n] = extra["uv1"]
        vo += n
    set_camera(sc, cam_m)

    def pose(time):
        """The RDScene at `time` seconds: Object::apply_animation down the tree (hierarchy.hpp:135-143), then
        build_rendering_scene (:222-290).  Resources (VBO/EBO/BLAS/materials) do not change."""
        inst, lts, (cm, _) = walk_scene(float(time))
        out = scenes.with_instance_transforms(sc, {i: m for i, (_, _, m) in enumerate(inst)})
        out.lights = lts
        set_camera(out, cm)
        return out

    sc.pose = pose
    sc.animated_nodes = sorted(animations)
    return sc


# ------------------------------------------------------------------------------------------------------------------
def save_gltf(sc, path):
    """Write a SceneData as <path>.gltf + <path>.bin: one mesh per VAO, one node per instance (matrix), lights,
    camera; one bufferView per accessor.  Scene material k is written as glTF material k, so it is material k + 1
    after `load_gltf` (the loader's default material takes slot 0)."""
    assert path.endswith(".gltf")
    blob = bytearray()
    views, accessors = [], []

    def add(arr, ctype, atype, target=None, minmax=False):
        arr = np.ascontiguousarray(arr)
        while len(blob) % 4:
            blob.append(0)
        v = dict(buffer=0, byteOffset=len(blob), byteLength=arr.nbytes)
        if target:
            v["target"] = target
        blob.extend(arr.tobytes())
        views.append(v)
        a = dict(bufferView=len(views) - 1, componentType=ctype, count=int(arr.shape[0]), type=atype)
        if minmax:
            a["min"] = [float(x) for x in arr.min(axis=0)]
            a["max"] = [float(x) for x in arr.max(axis=0)]
        accessors.append(a)
        return len(accessors) - 1

    meshes = []
    vaos = sc.vaos
    used_material = {}
    for o in sc.objects:
        used_material.setdefault(o["objectid"], o["materialid"])
    for p, (_, _, off, length) in enumerate(vaos):
        idx = sc.indices[off:off + length]
        lo, hi = int(idx.min()), int(idx.max())
        v = sc.vertices[lo:hi + 1]
        prim = dict(attributes=dict(POSITION=add(v["position"], 5126, "VEC3", 34962, True), NORMAL=add(v["normal"], 5126, "VEC3", 34962),
                                    TANGENT=add(np.concatenate([v["tangent"], np.ones((len(v), 1), np.float32)], axis=1), 5126, "VEC4", 34962),
                                    TEXCOORD_0=add(v["uv0"], 5126, "VEC2", 34962), COLOR_0=add(v["color"], 5126, "VEC3", 34962)),
                    indices=add((idx - lo).astype(np.uint32), 5125, "SCALAR", 34963), mode=4)
        if p in used_material:
            prim["material"] = int(used_material[p])  # comes back as id + 1: the loader puts its default material first
        meshes.append(dict(primitives=[prim]))
    materials = []
    for m in sc.materials:
        materials.append(dict(pbrMetallicRoughness=dict(baseColorFactor=[float(c) for c in m.get("color", (1, 1, 1))] + [1.0],
                                                        metallicFactor=float(m.get("metallic", 0.0)), roughnessFactor=float(m.get("roughness", 1.0))),
                              emissiveFactor=[float(c) for c in m.get("emission", (0, 0, 0))], doubleSided=bool(m.get("double_sided", True))))
    nodes = []
    for o in sc.objects:
        if o["materialid"] != used_material[o["objectid"]]:
            raise GltfError("an instance overrides its primitive's material: not expressible in glTF")
        nodes.append(dict(mesh=int(o["objectid"]), matrix=[float(x) for x in np.asarray(o["L2W"], dtype=np.float64).T.reshape(-1)]))
    lights = []
    for l in sc.lights:
        inten = max(l["intensity"]) or 1.0
        kind = {abi.LIGHT_DIRECTIONAL: "directional", abi.LIGHT_SPOT: "spot"}.get(l["type"], "point")
        lights.append(dict(type=kind, color=[float(c) / inten for c in l["intensity"]], intensity=float(inten)))
        if kind == "spot":
            lights[-1]["spot"] = {}
        if kind == "directional":
            d = np.asarray(l["direction"], dtype=np.float64)
            d /= np.linalg.norm(d)
            up = np.array([0, 1, 0.0]) if abs(d[1]) < 0.99 else np.array([1, 0, 0.0])
            m = np.linalg.inv(scenes.look_at((0, 0, 0), d, up))
        else:
            m = scenes.translate(l["position"])
        nodes.append(dict(matrix=[float(x) for x in m.T.reshape(-1)], extensions=dict(KHR_lights_punctual=dict(light=len(lights) - 1))))
    nodes.append(dict(camera=0, matrix=[float(x) for x in np.asarray(sc.camera_L2W, dtype=np.float64).T.reshape(-1)]))
    doc = dict(asset=dict(version="2.0", generator="lprt_amd.gltf"), scene=0, scenes=[dict(nodes=list(range(len(nodes))))], nodes=nodes,
               meshes=meshes, materials=materials, accessors=accessors, bufferViews=views,
               buffers=[dict(uri=os.path.basename(path)[:-5] + ".bin", byteLength=len(blob))],
               cameras=[dict(type="perspective", perspective=dict(yfov=float(sc.fovy), znear=float(sc.z_near), zfar=float(sc.z_far), aspectRatio=16 / 9))],
               extensionsUsed=["KHR_lights_punctual"], extensions=dict(KHR_lights_punctual=dict(lights=lights)))
    with open(path, "w") as f:
        json.dump(doc, f)
    with open(path[:-5] + ".bin", "wb") as f:
        f.write(bytes(blob))
    return path
