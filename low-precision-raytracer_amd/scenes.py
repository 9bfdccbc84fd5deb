"""Synthetic benchmark scenes: the reference ships no assets, so the Cornell box and
the "Sponza-like" atrium are generated here, in the flat-array form the loader/scene manager hand to
the renderer (rt/rtrt/loader.cu:298-442, rt/rtrt/hierarchy.hpp:222-290).

Everything is float32 on the host, exactly like RDVertex<float> / RDScene before the DataT conversion.
"""
from dataclasses import dataclass, field
import math
import numpy as np

from . import abi


# ----------------------------------------------------------------------------------------------
# small transform helpers (float64 internally, rounded to float32 once at the end)
# ----------------------------------------------------------------------------------------------
def translate(t):
    m = np.eye(4)
    m[:3, 3] = t
    return m


def scale(s):
    return np.diag([s[0], s[1], s[2], 1.0])


def rotate_y(deg):
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    m = np.eye(4)
    m[0, 0], m[0, 2], m[2, 0], m[2, 2] = c, s, -s, c
    return m


def rotate_x(deg):
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    m = np.eye(4)
    m[1, 1], m[1, 2], m[2, 1], m[2, 2] = c, -s, s, c
    return m


def look_at(eye, center, up):
    """glm::lookAt (right-handed), row-major world->view."""
    eye, center, up = (np.asarray(v, dtype=np.float64) for v in (eye, center, up))
    f = center - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    m = np.eye(4)
    m[0, :3], m[1, :3], m[2, :3] = s, u, -f
    m[0, 3], m[1, 3], m[2, 3] = -s @ eye, -u @ eye, f @ eye
    return m


def perspective_fov(fovy, width, height, z_near, z_far):
    """glm::perspectiveFov (RH, depth -1..1), row-major view->clip (rt/gui/imgui_window.cu:290-291)."""
    h = math.cos(0.5 * fovy) / math.sin(0.5 * fovy)
    w = h * height / width
    m = np.zeros((4, 4))
    m[0, 0], m[1, 1] = w, h
    m[2, 2] = -(z_far + z_near) / (z_far - z_near)
    m[3, 2] = -1.0
    m[2, 3] = -(2.0 * z_far * z_near) / (z_far - z_near)
    return m


def synth_tangent(n):
    """Tangent synthesis when the glTF has none (rt/rtrt/loader.cu:376-389)."""
    a, b, c = float(n[0]), float(n[1]), float(n[2])
    if abs(a) > 1e-4 or abs(b) > 1e-4:
        t = np.array([-b, a, 0.0])
    else:
        t = np.array([0.0, -c, b])
    return t / np.linalg.norm(t)


# ----------------------------------------------------------------------------------------------
@dataclass
class Mesh:
    """One glTF primitive: local-space vertices + primitive-local triangle indices."""
    positions: np.ndarray
    normals: np.ndarray
    indices: np.ndarray
    uv0: np.ndarray = None
    colors: np.ndarray = None
    uv1: np.ndarray = None

    @property
    def n_tris(self):
        return len(self.indices) // 3


@dataclass
class SceneData:
    name: str
    vertices: np.ndarray          # abi.VERTEX_DTYPE [nv]
    indices: np.ndarray           # uint32 [ni] (rebased to the shared VBO, loader.cu:330-332)
    vaos: np.ndarray              # uint32 [n_vaos, 4] = (ebo_index, vbo_index, ebo_offset, ebo_length)
    materials: list               # list of dict
    objects: list                 # list of dict(L2W, W2L, materialid, objectid)
    aabbs: np.ndarray             # float32 [n_objects, 6]
    lights: list                  # list of dict(type, direction, intensity, position)
    camera_L2W: np.ndarray        # float32 4x4 row-major
    camera_W2V: np.ndarray
    fovy: float
    z_near: float = 0.1
    z_far: float = 100.0
    local_aabbs: np.ndarray = None
    meta: dict = field(default_factory=dict)
    textures: list = field(default_factory=list)   # dict(rgba8=uint8[h, w, 4], srgb=bool); materials name them by "tex_index"
    skybox: dict = None                             # dict(tex_index, delta_x, delta_y, exposure) or None

    @property
    def n_tris(self):
        return len(self.indices) // 3

    def transform_W2C(self, width, height):
        p = perspective_fov(self.fovy, width, height, self.z_near, self.z_far)
        return (p @ self.camera_W2V.astype(np.float64)).astype(np.float32)

    # ---- ctypes views -------------------------------------------------------------------------
    def c_vaos(self):
        arr = (abi.MeshVAO * len(self.vaos))()
        for i, v in enumerate(self.vaos):
            arr[i] = abi.MeshVAO(int(v[0]), int(v[1]), int(v[2]), int(v[3]))
        return arr

    def c_materials(self, tex_handles=None):
        """tex_handles[i] = the handle lprt_upload_texture returned for self.textures[i]."""
        arr = (abi.Material * len(self.materials))()
        for i, m in enumerate(self.materials):
            c = arr[i]
            c.color[:] = [float(x) for x in m.get("color", (1, 1, 1))]
            c.emission[:] = [float(x) for x in m.get("emission", (0, 0, 0))]
            c.metallic = float(m.get("metallic", 0.0))
            c.roughness = float(m.get("roughness", 1.0))
            c.double_sided = int(m.get("double_sided", True))
            c.uv_color = int(m.get("uv_color", 0xFFFFFFFF))
            c.tex_color = int(m.get("tex_color", 0))
            if "tex_index" in m and c.uv_color < 2:
                k = int(m["tex_index"])  # an index nobody uploaded is passed through for the library to reject
                c.tex_color = int(tex_handles[k]) if tex_handles is not None and k < len(tex_handles) else k
        return arr

    def c_skybox(self, tex_handles=None):
        if not self.skybox:
            return None
        k = self.skybox["tex_index"]
        return abi.Skybox(1, int(tex_handles[k]) if tex_handles is not None else int(k), float(self.skybox.get("delta_x", 0.0)),
                          float(self.skybox.get("delta_y", 0.0)), float(self.skybox.get("exposure", 1.0)))

    def c_objects(self):
        arr = (abi.Object * len(self.objects))()
        for i, o in enumerate(self.objects):
            arr[i].transform_L2W[:] = [float(x) for x in np.asarray(o["L2W"], dtype=np.float32).reshape(-1)]
            arr[i].transform_W2L[:] = [float(x) for x in np.asarray(o["W2L"], dtype=np.float32).reshape(-1)]
            arr[i].materialid = int(o["materialid"])
            arr[i].objectid = int(o["objectid"])
        return arr

    def c_aabbs(self):
        arr = (abi.AABB * len(self.aabbs))()
        for i, a in enumerate(self.aabbs):
            arr[i].lower[:] = [float(x) for x in a[:3]]
            arr[i].upper[:] = [float(x) for x in a[3:]]
        return arr

    def c_lights(self):
        arr = (abi.Light * max(1, len(self.lights)))()
        for i, l in enumerate(self.lights):
            arr[i].type = int(l["type"])
            arr[i].maximum_distance = 1e5
            arr[i].direction[:] = [float(x) for x in l.get("direction", (0, 0, -1))]
            arr[i].intensity[:] = [float(x) for x in l["intensity"]]
            arr[i].position[:] = [float(x) for x in l.get("position", (0, 0, 0))]
        return arr

    def c_camera(self, width, height):
        cam = abi.Camera()
        cam.transform[:] = [float(x) for x in self.camera_W2V.reshape(-1)]
        cam.transform_L2W[:] = [float(x) for x in self.camera_L2W.reshape(-1)]
        cam.field_of_view_y = float(self.fovy)
        cam.aspect_ratio = float(width) / float(height)
        cam.z_near, cam.z_far = self.z_near, self.z_far
        return cam


def assemble(name, meshes, materials, instances, lights, cam_eye, cam_target, fovy, meta=None):
    """meshes: list[Mesh]; instances: list[(mesh_index, material_index, L2W 4x4 float64)]."""
    nv = sum(len(m.positions) for m in meshes)
    verts = np.zeros(nv, dtype=abi.VERTEX_DTYPE)
    idx_parts, vaos, local_aabbs = [], [], []
    vo, io = 0, 0
    for m in meshes:
        n = len(m.positions)
        pos = np.asarray(m.positions, dtype=np.float32)
        nrm = np.asarray(m.normals, dtype=np.float32)
        verts["position"][vo:vo + n] = pos
        verts["normal"][vo:vo + n] = nrm
        tan = np.zeros((n, 3))
        nn = nrm.astype(np.float64)
        big = (np.abs(nn[:, 0]) > 1e-4) | (np.abs(nn[:, 1]) > 1e-4)
        tan[big] = np.stack([-nn[big, 1], nn[big, 0], np.zeros(big.sum())], axis=1)
        tan[~big] = np.stack([np.zeros((~big).sum()), -nn[~big, 2], nn[~big, 1]], axis=1)
        tan /= np.linalg.norm(tan, axis=1, keepdims=True)
        verts["tangent"][vo:vo + n] = tan.astype(np.float32)
        verts["color"][vo:vo + n] = 1.0 if m.colors is None else np.asarray(m.colors, dtype=np.float32)
        if m.uv0 is not None:
            verts["uv0"][vo:vo + n] = np.asarray(m.uv0, dtype=np.float32)
        if m.uv1 is not None:
            verts["uv1"][vo:vo + n] = np.asarray(m.uv1, dtype=np.float32)
        idx = np.asarray(m.indices, dtype=np.uint32)
        assert len(idx) % 3 == 0 and idx.max() < n
        idx_parts.append(idx + np.uint32(vo))
        vaos.append((0, 0, io, len(idx)))
        local_aabbs.append(np.concatenate([pos.min(axis=0), pos.max(axis=0)]))
        vo += n
        io += len(idx)
    objects, aabbs = [], []
    for mesh_i, mat_i, l2w in instances:
        l2w32 = np.asarray(l2w, dtype=np.float64).astype(np.float32)
        w2l32 = np.linalg.inv(l2w32.astype(np.float64)).astype(np.float32)  # glm::inverse, hierarchy.hpp:244
        objects.append(dict(L2W=l2w32, W2L=w2l32, materialid=mat_i, objectid=mesh_i))
        lo, hi = local_aabbs[mesh_i][:3], local_aabbs[mesh_i][3:]
        corners = np.array([[(lo, hi)[(i >> k) & 1][k] for k in range(3)] + [1.0] for i in range(8)], dtype=np.float32)
        wc = (l2w32 @ corners.T).T  # hierarchy.hpp:43-61: 8 transformed corners, float32
        wc = wc[:, :3] / wc[:, 3:4]
        aabbs.append(np.concatenate([wc.min(axis=0), wc.max(axis=0)]).astype(np.float32))
    cam_l2w = translate(cam_eye)
    fwd = np.asarray(cam_target, dtype=np.float64) - np.asarray(cam_eye, dtype=np.float64)
    fwd /= np.linalg.norm(fwd)
    w2v = look_at(cam_eye, np.asarray(cam_eye) + fwd, (0, 1, 0))
    cam_l2w = np.linalg.inv(w2v)
    return SceneData(name=name, vertices=verts, indices=np.concatenate(idx_parts).astype(np.uint32),
                     vaos=np.asarray(vaos, dtype=np.uint32), materials=materials, objects=objects,
                     aabbs=np.asarray(aabbs, dtype=np.float32), lights=lights,
                     camera_L2W=cam_l2w.astype(np.float32), camera_W2V=w2v.astype(np.float32), fovy=float(fovy),
                     local_aabbs=np.asarray(local_aabbs, dtype=np.float32), meta=meta or {})


# ----------------------------------------------------------------------------------------------
# mesh builders
# ----------------------------------------------------------------------------------------------
def quad_grid(origin, du, dv, nu=1, nv=1, displace=None):
    """(nu x nv) cells spanning origin + s*du + t*dv; normal = normalize(du x dv)."""
    origin, du, dv = (np.asarray(v, dtype=np.float64) for v in (origin, du, dv))
    s, t = np.meshgrid(np.linspace(0, 1, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    pos = origin + s[..., None] * du + t[..., None] * dv
    n = np.cross(du, dv)
    n /= np.linalg.norm(n)
    nrm = np.broadcast_to(n, pos.shape).copy()
    if displace is not None:
        pos, nrm = displace(pos, nrm, s, t)
    ii, jj = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (ii * (nv + 1) + jj).reshape(-1)
    b = a + (nv + 1)
    idx = np.stack([a, b, b + 1, a, b + 1, a + 1], axis=1).reshape(-1)
    uv = np.stack([s, t], axis=-1).reshape(-1, 2)
    return Mesh(pos.reshape(-1, 3), nrm.reshape(-1, 3), idx.astype(np.uint32), uv0=uv)


def merge(meshes):
    pos, nrm, idx, uv = [], [], [], []
    off = 0
    for m in meshes:
        pos.append(m.positions)
        nrm.append(m.normals)
        idx.append(m.indices.astype(np.int64) + off)
        uv.append(m.uv0 if m.uv0 is not None else np.zeros((len(m.positions), 2)))
        off += len(m.positions)
    return Mesh(np.concatenate(pos), np.concatenate(nrm), np.concatenate(idx).astype(np.uint32), uv0=np.concatenate(uv))


def unit_cube():
    """[-0.5,0.5]^3, 12 triangles, outward face normals, 24 vertices."""
    faces = []
    h = 0.5
    faces.append(quad_grid((-h, -h, h), (1, 0, 0), (0, 1, 0)))    # +z
    faces.append(quad_grid((h, -h, -h), (-1, 0, 0), (0, 1, 0)))   # -z
    faces.append(quad_grid((h, -h, h), (0, 0, -1), (0, 1, 0)))    # +x
    faces.append(quad_grid((-h, -h, -h), (0, 0, 1), (0, 1, 0)))   # -x
    faces.append(quad_grid((-h, h, h), (1, 0, 0), (0, 0, -1)))    # +y
    faces.append(quad_grid((-h, -h, -h), (1, 0, 0), (0, 0, 1)))   # -y
    return merge(faces)


def transformed(mesh, m):
    m = np.asarray(m, dtype=np.float64)
    p = mesh.positions @ m[:3, :3].T + m[:3, 3]
    n = mesh.normals @ np.linalg.inv(m[:3, :3])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    return Mesh(p, n, mesh.indices, uv0=mesh.uv0)


def fluted_column(radius, height, n_around, n_high, flutes=12, depth=0.06):
    th = np.linspace(0, 2 * math.pi, n_around + 1)
    yy = np.linspace(0, height, n_high + 1)
    T, Y = np.meshgrid(th, yy, indexing="ij")
    entasis = 1.0 - 0.12 * (Y / height) ** 2
    r = radius * entasis * (1.0 - depth * (0.5 + 0.5 * np.cos(flutes * T)))
    pos = np.stack([r * np.cos(T), Y, r * np.sin(T)], axis=-1)
    # analytic-ish normals from finite differences of the parametrisation
    dT = np.gradient(pos, axis=0)
    dY = np.gradient(pos, axis=1)
    n = np.cross(dY, dT)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    ii, jj = np.meshgrid(np.arange(n_around), np.arange(n_high), indexing="ij")
    a = (ii * (n_high + 1) + jj).reshape(-1)
    b = a + (n_high + 1)
    idx = np.stack([a, b, b + 1, a, b + 1, a + 1], axis=1).reshape(-1)
    uv = np.stack([T / (2 * math.pi), Y / height], axis=-1).reshape(-1, 2)
    return Mesh(pos.reshape(-1, 3), n.reshape(-1, 3), idx.astype(np.uint32), uv0=uv)


def arch(span, thickness, depth, n_arc, n_sides=4):
    """Half ring in the x-y plane (opening down), rectangular cross-section swept along the arc."""
    R, r = span / 2.0, thickness / 2.0
    phi = np.linspace(0, math.pi, n_arc + 1)
    ring = [(-r, -depth / 2), (r, -depth / 2), (r, depth / 2), (-r, depth / 2)]
    meshes = []
    for k in range(4):
        (a0, z0), (a1, z1) = ring[k], ring[(k + 1) % 4]
        sub = max(1, n_sides)
        s = np.linspace(0, 1, sub + 1)
        P, S = np.meshgrid(phi, s, indexing="ij")
        rad = R + a0 + (a1 - a0) * S
        z = z0 + (z1 - z0) * S
        pos = np.stack([rad * np.cos(P), rad * np.sin(P), z], axis=-1)
        dP = np.gradient(pos, axis=0)
        dS = np.gradient(pos, axis=1)
        n = np.cross(dS, dP)
        n /= np.maximum(np.linalg.norm(n, axis=-1, keepdims=True), 1e-12)
        ii, jj = np.meshgrid(np.arange(n_arc), np.arange(sub), indexing="ij")
        a = (ii * (sub + 1) + jj).reshape(-1)
        b = a + (sub + 1)
        idx = np.stack([a, b, b + 1, a, b + 1, a + 1], axis=1).reshape(-1)
        meshes.append(Mesh(pos.reshape(-1, 3), n.reshape(-1, 3), idx.astype(np.uint32)))
    return merge(meshes)


# ----------------------------------------------------------------------------------------------
# Scene 1: Cornell box (36 triangles, 8 primitives/objects, 6 materials)
# ----------------------------------------------------------------------------------------------
def cornell_box():
    white = dict(color=(0.73, 0.73, 0.73), metallic=0.0, roughness=1.0)
    materials = [
        white,
        dict(color=(0.65, 0.05, 0.05), metallic=0.0, roughness=1.0),
        dict(color=(0.12, 0.45, 0.15), metallic=0.0, roughness=1.0),
        dict(color=(0.73, 0.73, 0.73), metallic=1.0, roughness=0.2),    # tall box: GGX branch
        dict(color=(0.73, 0.73, 0.73), metallic=0.0, roughness=0.05),   # short box: mirror / glassy branch
        dict(color=(1.0, 1.0, 1.0), emission=(17.0, 12.0, 4.0), metallic=0.0, roughness=1.0),
    ]
    meshes = [
        quad_grid((-1, -1, 1), (2, 0, 0), (0, 0, -2)),     # 0 floor   (normal +y)
        quad_grid((-1, 1, -1), (2, 0, 0), (0, 0, 2)),      # 1 ceiling (normal -y)
        quad_grid((-1, -1, -1), (2, 0, 0), (0, 2, 0)),     # 2 back    (normal +z)
        quad_grid((-1, -1, 1), (0, 0, -2), (0, 2, 0)),     # 3 left    (normal +x)
        quad_grid((1, -1, -1), (0, 0, 2), (0, 2, 0)),      # 4 right   (normal -x)
        unit_cube(),                                       # 5 tall box (local unit cube)
        unit_cube(),                                       # 6 short box
        quad_grid((-0.25, 0.0, -0.25), (0.5, 0, 0), (0, 0, 0.5)),  # 7 emitter, local y = 0 (normal -y)
    ]
    eye = np.eye(4)
    tall = translate((-0.33, -0.4, -0.3)) @ rotate_y(17.0) @ scale((0.6, 1.2, 0.6))
    short = translate((0.33, -0.7, 0.3)) @ rotate_y(-18.0) @ scale((0.6, 0.6, 0.6))
    emit = translate((0.0, 0.999, 0.0))
    instances = [(0, 0, eye), (1, 0, eye), (2, 0, eye), (3, 1, eye), (4, 2, eye), (5, 3, tall), (6, 4, short),
                 (7, 5, emit)]
    lights = [dict(type=abi.LIGHT_POINT, position=(0.0, 0.9, 0.0), intensity=(8.0, 8.0, 8.0), direction=(0, 0, -1))]
    sc = assemble("cornell", meshes, materials, instances, lights, cam_eye=(0.0, 0.0, 3.5), cam_target=(0, 0, 0),
                  fovy=0.6911)
    assert sc.n_tris == 36 and len(sc.objects) == 8 and len(sc.materials) == 6
    return sc


# ----------------------------------------------------------------------------------------------
# Scene 2: "Sponza-like" procedural atrium (262,144 +- 1 % triangles, 24 primitives, 12 materials)
# ----------------------------------------------------------------------------------------------
def sponza_like(target_tris=262144, seed=20250304):
    rng = np.random.default_rng(seed)
    L, Hh, Wd = 30.0, 12.0, 18.0  # x: length, y: height, z: width
    x0, z0 = -L / 2, -Wd / 2
    materials = [
        dict(color=(0.8, 0.8, 0.8), metallic=0.0, roughness=1.0),         # 0 default
        dict(color=(0.55, 0.5, 0.45), metallic=0.0, roughness=0.7),       # 1 floor stone
        dict(color=(0.7, 0.68, 0.62), metallic=0.0, roughness=1.0),       # 2 walls
        dict(color=(0.6, 0.6, 0.65), metallic=0.0, roughness=0.9),        # 3 ceiling
        dict(color=(0.5, 0.45, 0.4), metallic=0.0, roughness=0.6),        # 4 gallery slabs
        dict(color=(0.75, 0.72, 0.65), metallic=0.0, roughness=0.5),      # 5 lower columns
        dict(color=(0.9, 0.75, 0.4), metallic=1.0, roughness=0.3),        # 6 upper columns (brass)
        dict(color=(0.65, 0.6, 0.55), metallic=0.0, roughness=0.8),       # 7 arches
        dict(color=(0.7, 0.1, 0.1), metallic=0.0, roughness=0.9),         # 8 drape red
        dict(color=(0.1, 0.25, 0.6), metallic=0.0, roughness=0.9),        # 9 drape blue
        dict(color=(0.1, 0.5, 0.2), metallic=0.0, roughness=1.0),         # 10 drape green
        dict(color=(0.8, 0.8, 0.85), metallic=1.0, roughness=0.4),        # 11 drape silver
    ]

    def bumpy(amp, fx, fz):
        def f(pos, nrm, s, t):
            h = amp * np.sin(fx * s * 2 * math.pi) * np.sin(fz * t * 2 * math.pi)
            pos = pos + nrm * h[..., None]
            return pos, nrm
        return f

    def drape(nu, nv, phase):
        def f(pos, nrm, s, t):
            fold = 0.18 * np.sin(9.0 * s * 2 * math.pi + phase) * (0.3 + 0.7 * t)
            pos = pos + nrm * fold[..., None]
            d_ds = 0.18 * 9.0 * 2 * math.pi * np.cos(9.0 * s * 2 * math.pi + phase) * (0.3 + 0.7 * t)
            n2 = nrm - np.array([1.0, 0.0, 0.0]) * (d_ds / 3.0)[..., None]
            n2 /= np.linalg.norm(n2, axis=-1, keepdims=True)
            return pos, n2
        return quad_grid((-1.5, 0.0, 0.0), (3.0, 0, 0), (0, -5.0, 0), nu, nv, displace=f)

    col_lo = fluted_column(0.42, 5.2, 32, 30)
    col_hi = fluted_column(0.32, 4.6, 32, 30, flutes=16)
    arch_m = arch(2.3, 0.45, 0.7, 24, 4)
    col_x = np.linspace(x0 + 2.0, -x0 - 2.0, 11)
    spacing = col_x[1] - col_x[0]

    def group(mesh, placements):
        return merge([transformed(mesh, m) for m in placements])

    meshes, instances = [], []

    def add(mesh, mat, l2w=None):
        meshes.append(mesh)
        instances.append((len(meshes) - 1, mat, np.eye(4) if l2w is None else l2w))

    zl, zr = z0 + 4.0, -z0 - 4.0
    # columns: 4 groups of 11
    add(group(col_lo, [translate((x, 0.0, zl)) for x in col_x]), 5)
    add(group(col_lo, [translate((x, 0.0, zr)) for x in col_x]), 5)
    add(group(col_hi, [translate((x, 6.0, zl)) for x in col_x]), 6)
    add(group(col_hi, [translate((x, 6.0, zr)) for x in col_x]), 6)
    # arches between neighbouring columns: 4 groups of 10
    a_lo = [translate((0.5 * (col_x[i] + col_x[i + 1]), 4.6, 0.0)) @ scale((spacing / 2.75, 1.0, 1.0)) for i in range(10)]
    a_hi = [translate((0.5 * (col_x[i] + col_x[i + 1]), 10.2, 0.0)) @ scale((spacing / 2.75, 0.8, 1.0)) for i in range(10)]
    add(group(arch_m, [translate((0, 0, zl)) @ m for m in a_lo]), 7)
    add(group(arch_m, [translate((0, 0, zr)) @ m for m in a_lo]), 7)
    add(group(arch_m, [translate((0, 0, zl)) @ m for m in a_hi]), 7)
    add(group(arch_m, [translate((0, 0, zr)) @ m for m in a_hi]), 7)
    # 8 drapes, instanced with their own transforms (exercise L2W / W2L)
    for k in range(8):
        side = zl + 0.6 if k % 2 == 0 else zr - 0.6
        x = col_x[1 + k] + 0.5 * spacing * (k % 2)
        rot = rotate_y(float(rng.uniform(-12, 12)))
        add(drape(56, 64, float(rng.uniform(0, 6.28))), 8 + (k % 4), translate((x, 11.0, side)) @ rot)
    # gallery slabs (upper floors left and right)
    add(quad_grid((x0, 5.6, z0), (0, 0, 4.6), (L, 0, 0), 24, 96, displace=bumpy(0.01, 12, 40)), 4)
    add(quad_grid((x0, 5.6, -z0 - 4.6), (0, 0, 4.6), (L, 0, 0), 24, 96, displace=bumpy(0.01, 12, 40)), 4)
    # shell: walls + ceiling, then the floor takes whatever is left of the budget
    add(quad_grid((x0, 0, z0), (L, 0, 0), (0, Hh, 0), 96, 40, displace=bumpy(0.02, 30, 12)), 2)       # back wall z = z0 (normal +z)
    add(quad_grid((-x0, 0, -z0), (-L, 0, 0), (0, Hh, 0), 96, 40, displace=bumpy(0.02, 30, 12)), 2)    # front wall (normal -z)
    add(quad_grid((x0, 0, -z0), (0, 0, -Wd), (0, Hh, 0), 60, 40, displace=bumpy(0.02, 18, 12)), 2)    # end wall x = x0 (normal +x)
    add(quad_grid((-x0, 0, z0), (0, 0, Wd), (0, Hh, 0), 60, 40, displace=bumpy(0.02, 18, 12)), 2)     # end wall (normal -x)
    add(quad_grid((x0, Hh, z0), (L, 0, 0), (0, 0, Wd), 80, 48, displace=bumpy(0.05, 10, 6)), 3)       # ceiling (normal -y)
    used = sum(m.n_tris for m in meshes)
    remaining = max(2, target_tris - used)
    nu = max(1, int(round(math.sqrt(remaining / 2 * L / Wd))))
    nv = max(1, int(round(remaining / 2 / nu)))
    add(quad_grid((x0, 0.0, -z0), (L, 0, 0), (0, 0, -Wd), nu, nv, displace=bumpy(0.015, 20, 12)), 1)  # floor (normal +y)

    lights = [
        dict(type=abi.LIGHT_DIRECTIONAL, direction=(0.35, -0.85, 0.25), intensity=(2.5, 2.4, 2.2), position=(0, 0, 0)),
        dict(type=abi.LIGHT_POINT, position=(-6.0, 4.0, 0.0), intensity=(60.0, 50.0, 40.0), direction=(0, 0, -1)),
        dict(type=abi.LIGHT_POINT, position=(7.0, 8.5, 1.0), intensity=(40.0, 45.0, 60.0), direction=(0, 0, -1)),
    ]
    sc = assemble("sponza_like", meshes, materials, instances, lights, cam_eye=(-11.0, 2.0, 0.0),
                  cam_target=(0.0, 2.0, 0.0), fovy=1.0, meta=dict(seed=seed, target_tris=target_tris))
    assert len(sc.vaos) == 24 and len(sc.materials) == 12
    assert abs(sc.n_tris - target_tris) <= 0.01 * target_tris, sc.n_tris
    return sc


def small_test_scene(seed=7, n_tris_per_object=40, n_objects=5):
    """Random triangle soups under random affine transforms: exercises ragged BLAS sizes (including a
    1-triangle and a 2-triangle primitive), non-trivial W2L/L2W and several materials."""
    rng = np.random.default_rng(seed)
    meshes, instances, materials = [], [], []
    sizes = [1, 2, 3] + [n_tris_per_object + 7 * i for i in range(max(0, n_objects - 3))]
    for i, nt in enumerate(sizes[:n_objects]):
        centers = rng.uniform(-1, 1, size=(nt, 1, 3))
        tri = centers + rng.normal(scale=0.35, size=(nt, 3, 3))
        pos = tri.reshape(-1, 3)
        nrm = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
        nrm = np.repeat(nrm, 3, axis=0)
        meshes.append(Mesh(pos, nrm, np.arange(3 * nt, dtype=np.uint32), uv0=rng.uniform(0, 1, size=(3 * nt, 2)),
                           colors=rng.uniform(0.2, 1.0, size=(3 * nt, 3))))
        materials.append(dict(color=tuple(rng.uniform(0.2, 0.9, 3)), metallic=float(rng.choice([0.0, 0.5, 1.0])),
                              roughness=float(rng.choice([0.05, 0.3, 0.8, 1.0])),
                              emission=(0.5, 0.2, 0.1) if i == 0 else (0, 0, 0), double_sided=bool(i % 4 != 3)))
        l2w = translate(rng.uniform(-1.5, 1.5, 3)) @ rotate_y(float(rng.uniform(0, 360))) @ rotate_x(
            float(rng.uniform(-30, 30))) @ scale(rng.uniform(0.6, 1.4, 3))
        instances.append((i, i, l2w))
    instances.append((len(meshes) - 1, 0, translate((0.0, -2.5, 0.0)) @ scale((3.0, 0.5, 3.0))))  # second instance
    lights = [dict(type=abi.LIGHT_POINT, position=(0.5, 3.0, 2.0), intensity=(30.0, 28.0, 25.0)),
              dict(type=abi.LIGHT_DIRECTIONAL, direction=(-0.3, -1.0, -0.2), intensity=(1.5, 1.5, 1.6)),
              dict(type=abi.LIGHT_SPOT, position=(-2.0, 2.0, 3.0), intensity=(12.0, 14.0, 16.0))]
    return assemble("small_test", meshes, materials, instances, lights, cam_eye=(0.3, 0.4, 6.0), cam_target=(0, 0, 0),
                    fovy=0.8)


def checker_texture(w=16, h=8, seed=3):
    """Small RGBA8 image with distinct texels everywhere (a checker modulated by a seeded random tint)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    dark = ((xx + yy) & 1) == 0
    img[dark, :3] //= 4
    img[..., 3] = 255
    return img


def textured_scene():
    """Open scene for the texture / skybox paths: a floor whose uv0 runs 0..3 (wrap addressing) with an sRGB
    base-colour texture, a box textured through uv1 with a linear texture, an untextured box, and an
    equirectangular sky texture behind everything (most of the frame is empty pixels)."""
    floor = quad_grid((-2, -1, 2), (4, 0, 0), (0, 0, -4), 4, 4)
    floor.uv0 = floor.uv0 * 3.0 - 0.75
    box = unit_cube()
    box.uv1 = box.uv0 * 1.5
    box.uv0 = None
    plain = unit_cube()
    materials = [dict(color=(0.8, 0.8, 0.8), metallic=0.0, roughness=1.0),
                 dict(color=(0.2, 0.9, 0.2), metallic=0.0, roughness=0.8, uv_color=0, tex_index=0),   # colour factor unused once textured
                 dict(color=(1.0, 1.0, 1.0), metallic=0.5, roughness=0.4, uv_color=1, tex_index=1),
                 dict(color=(0.7, 0.3, 0.2), metallic=0.0, roughness=0.05)]
    instances = [(0, 1, np.eye(4)), (1, 2, translate((-0.6, -0.5, 0.0)) @ rotate_y(25.0)),
                 (2, 3, translate((0.7, -0.6, 0.4)) @ rotate_y(-15.0) @ scale((0.8, 0.8, 0.8)))]
    lights = [dict(type=abi.LIGHT_POINT, position=(0.5, 2.5, 1.5), intensity=(40.0, 38.0, 35.0)),
              dict(type=abi.LIGHT_DIRECTIONAL, direction=(-0.4, -1.0, -0.3), intensity=(1.2, 1.2, 1.3))]
    sc = assemble("textured", [floor, box, plain], materials, instances, lights, cam_eye=(0.2, 0.6, 4.5), cam_target=(0, -0.3, 0), fovy=0.8)
    sky = np.zeros((32, 64, 4), dtype=np.uint8)
    yy, xx = np.mgrid[0:32, 0:64]
    sky[..., 0] = 40 + 3 * xx
    sky[..., 1] = 90 + 5 * yy
    sky[..., 2] = 255 - 2 * xx
    sky[..., 3] = 255
    sc.textures = [dict(rgba8=checker_texture(16, 8, 3), srgb=True), dict(rgba8=checker_texture(5, 7, 11), srgb=False),
                   dict(rgba8=sky, srgb=True)]
    sc.skybox = dict(tex_index=2, delta_x=0.125, delta_y=0.0, exposure=1.5)
    return sc


SCENES = {"textured": textured_scene, "cornell": cornell_box, "sponza_like": sponza_like, "small_test": small_test_scene}


def with_instance_transforms(sc, transforms):
    """The RDScene of another frame: same resources, new local-to-world matrices for some instances
    ({instance index: 4x4}); W2L and the world AABBs follow as in hierarchy::Scene::build_rendering_scene
    (rt/rtrt/hierarchy.hpp:236-247)."""
    import copy
    out = copy.copy(sc)
    out.objects = [dict(o) for o in sc.objects]
    out.aabbs = sc.aabbs.copy()
    for i, l2w in transforms.items():
        l2w32 = np.asarray(l2w, dtype=np.float64).astype(np.float32)
        o = out.objects[i]
        o["L2W"] = l2w32
        o["W2L"] = np.linalg.inv(l2w32.astype(np.float64)).astype(np.float32)
        lo, hi = sc.local_aabbs[o["objectid"]][:3], sc.local_aabbs[o["objectid"]][3:]
        corners = np.array([[(lo, hi)[(c >> k) & 1][k] for k in range(3)] + [1.0] for c in range(8)], dtype=np.float32)
        wc = (l2w32 @ corners.T).T
        wc = wc[:, :3] / wc[:, 3:4]
        out.aabbs[i] = np.concatenate([wc.min(axis=0), wc.max(axis=0)]).astype(np.float32)
    return out


def moving_cornell(k, base=None, climb=0.02):
    """Frame k of a Cornell-box sequence in which everything moves (the animation path of rt/rtrt/hierarchy.hpp:81-149 as
    the scene manager would hand it over): the tall box turns 6 degrees per frame, the short box slides, and the camera
    dollies in while drifting right and UP (climb per frame), so that reprojection moves history taps across image rows."""
    base = base or cornell_box()
    tall = translate((-0.33, -0.4, -0.3)) @ rotate_y(17.0 + 6.0 * k) @ scale((0.6, 1.2, 0.6))
    short = translate((0.33 - 0.05 * k, -0.7, 0.3 + 0.03 * k)) @ rotate_y(-18.0) @ scale((0.6, 0.6, 0.6))
    sc = with_instance_transforms(base, {5: tall, 6: short})
    eye = np.array([0.05 * k, climb * k, 3.5 - 0.1 * k])
    w2v = look_at(eye, (0, 0, 0), (0, 1, 0))
    sc.camera_W2V, sc.camera_L2W = w2v.astype(np.float32), np.linalg.inv(w2v).astype(np.float32)
    return sc


def write_blob(sc, width, height, path):
    """Flat little-endian dump of what the loader + scene manager hand to the renderer (examples/compat_demo.cpp
    reads it): "LPRTSCN1", six uint32 counts, then vertices, indices, VAOs, materials, objects, AABBs, lights, camera,
    world-to-clip, each array in its include/lprt.h layout."""
    import ctypes as C
    vaos, mats, objs, boxes, lights = sc.c_vaos(), sc.c_materials(), sc.c_objects(), sc.c_aabbs(), sc.c_lights()
    cam = sc.c_camera(width, height)
    w2c = np.ascontiguousarray(sc.transform_W2C(width, height), dtype=np.float32)
    with open(path, "wb") as f:
        f.write(b"LPRTSCN1")
        f.write(np.array([len(sc.vertices), len(sc.indices), len(sc.vaos), len(sc.materials), len(sc.objects), len(sc.lights)],
                         dtype="<u4").tobytes())
        f.write(np.ascontiguousarray(sc.vertices).tobytes())
        f.write(np.ascontiguousarray(sc.indices, dtype="<u4").tobytes())
        f.write(bytes(vaos))
        f.write(bytes(mats))
        f.write(bytes(objs))
        f.write(bytes(boxes))
        f.write(bytes(lights)[:C.sizeof(abi.Light) * len(sc.lights)])
        f.write(bytes(cam))
        f.write(w2c.tobytes())
    return path


def instanced_field(n_instances=150, seed=5):
    """Many instances of a few small meshes (more than the 32 the flat TLAS scan handles): exercises the TLAS tree
    walk, repeated BLASes under different transforms and several materials."""
    rng = np.random.default_rng(seed)
    meshes = [unit_cube(), fluted_column(0.3, 1.0, 8, 3), quad_grid((-0.5, 0, -0.5), (1, 0, 0), (0, 0, 1), 2, 2)]
    materials = [dict(color=tuple(rng.uniform(0.2, 0.9, 3)), metallic=float(rng.choice([0.0, 1.0])),
                      roughness=float(rng.choice([0.05, 0.4, 1.0]))) for _ in range(7)]
    side = int(math.ceil(math.sqrt(n_instances)))
    instances = []
    for i in range(n_instances):
        gx, gz = i % side, i // side
        t = translate((1.6 * (gx - side / 2) + rng.uniform(-0.2, 0.2), rng.uniform(-0.3, 0.3), 1.6 * (gz - side / 2) + rng.uniform(-0.2, 0.2)))
        l2w = t @ rotate_y(float(rng.uniform(0, 360))) @ rotate_x(float(rng.uniform(-20, 20))) @ scale(rng.uniform(0.5, 1.2, 3))
        instances.append((int(rng.integers(0, len(meshes))), int(rng.integers(0, len(materials))), l2w))
    floor = quad_grid((-16, -0.9, 16), (32, 0, 0), (0, 0, -32), 8, 8)
    meshes.append(floor)
    instances.append((len(meshes) - 1, 0, np.eye(4)))
    lights = [dict(type=abi.LIGHT_POINT, position=(0.0, 6.0, 0.0), intensity=(200.0, 190.0, 180.0)),
              dict(type=abi.LIGHT_DIRECTIONAL, direction=(0.3, -1.0, 0.2), intensity=(1.0, 1.0, 1.1))]
    return assemble("instanced_field", meshes, materials, instances, lights, cam_eye=(0.0, 5.0, 14.0), cam_target=(0, 0, 0), fovy=0.9)


SCENES["instanced_field"] = instanced_field
