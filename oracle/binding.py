"""TEST INFRASTRUCTURE: ctypes binding of oracle/liboracle.so (see oracle/half.hpp header note)."""
import ctypes as C
import os
import subprocess
import sys
import weakref

import numpy as np

import lprt_amd  # noqa: F401  (import shim for the hyphenated package directory)
from lprt_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("oracle_api.cpp", "render.hpp", "core.hpp", "half.hpp")]
    srcs.append(os.path.join(_HERE, "..", "include", "lprt.h"))
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "liboracle.so"], check=True, capture_output=True)
    if os.path.isdir("/root/reference"):
        ref = os.path.join(_HERE, "_ref", "libref_half.so")
        if force or not os.path.exists(ref):
            subprocess.run(["make", "-C", _HERE, "ref"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        try:  # PyTorch bundles a libgomp with the system SONAME: let it load first so the process keeps ONE OpenMP runtime
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(build())
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_int, C.c_uint32, C.c_uint32]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_set_threads.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_tie_order.argtypes = [C.c_void_p, C.c_int]
        L.orc_upload_geometry.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                          C.c_uint32]
        L.orc_upload_materials.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.orc_set_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_render_frame.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_upload_texture.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
        L.orc_tex2d.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_reset_history.argtypes = [C.c_void_p]
        L.orc_read_buffer.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
        L.orc_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_get_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_fmath.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p]
        L.orc_fmath.restype = None
        L.orc_reset_counters.argtypes = [C.c_void_p]
        L.orc_get_stage_seconds.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_reset_stage_seconds.argtypes = [C.c_void_p]
        L.orc_f32_to_f16.restype = C.c_uint16
        L.orc_f32_to_f16.argtypes = [C.c_float]
        L.orc_f16_to_f32.restype = C.c_float
        L.orc_f16_to_f32.argtypes = [C.c_uint16]
        L.orc_half_op.restype = C.c_uint16
        L.orc_half_op.argtypes = [C.c_int, C.c_uint16, C.c_uint16]
        L.orc_material_brdf.argtypes = [C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_xorwow.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_uint32, C.c_void_p]
        L.orc_xorwow_state.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
        L.orc_xorwow_matrix.argtypes = [C.c_int, C.c_void_p]
        L.orc_blas_node_count.restype = C.c_uint32
        L.orc_blas_node_count.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_tlas_node_count.restype = C.c_uint32
        L.orc_tlas_node_count.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


def ref_half_lib():
    """The reference's own half.hpp behind a C shim (oracle/_ref/, built only where /root/reference exists)."""
    p = os.path.join(_HERE, "_ref", "libref_half.so")
    if not os.path.exists(p):
        return None
    L = C.CDLL(p)
    L.ref_f32_to_f16.restype = C.c_uint16
    L.ref_f32_to_f16.argtypes = [C.c_float]
    L.ref_f16_to_f32.restype = C.c_float
    L.ref_f16_to_f32.argtypes = [C.c_uint16]
    L.ref_half_op.restype = C.c_uint16
    L.ref_half_op.argtypes = [C.c_int, C.c_uint16, C.c_uint16]
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _zeros(shape, dtype):
    return np.zeros(shape, dtype=dtype)


alloc_out = _zeros  # tests/guarded.py swaps in arrays that end at a PROT_NONE page
_LIVE = weakref.WeakSet()


def close_all():
    for o in list(_LIVE):
        o.close()


def fmath(op, x):
    """include/lprt_fmath.h as the host evaluates it: op in ('log', 'sin', 'cos', 'tan')."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    lib().orc_fmath(("log", "sin", "cos", "tan").index(op), _ptr(x), x.size, _ptr(out))
    return out


class Oracle:
    """CPU restatement of Renderer<DataT> (rt/gui/imgui_window.cu:78-367)."""

    def __init__(self, width, height, precision=abi.F32, threads=None):
        self.L = lib()
        self.W, self.H, self.precision = width, height, precision
        self.h = self.L.orc_create(precision, width, height)
        if not self.h:
            raise RuntimeError("orc_create failed")
        # default: the CPUs this process may run on, at most 32 (a GPU box shows 256 logical CPUs but grants a share of them:
        # 255 spinning OpenMP workers only get in each other's way)
        self.L.orc_set_threads(self.h, threads or max(1, min(len(os.sched_getaffinity(0)), 32)))
        _LIVE.add(self)

    def close(self):
        if getattr(self, "h", None):
            h, self.h = self.h, None
            _LIVE.discard(self)
            self.L.orc_destroy(h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def load_scene(self, sc):
        self.scene = sc
        verts = np.ascontiguousarray(sc.vertices)
        idx = np.ascontiguousarray(sc.indices, dtype=np.uint32)
        vaos = sc.c_vaos()
        rc = self.L.orc_upload_geometry(self.h, _ptr(verts), len(verts), _ptr(idx), len(idx), C.byref(vaos), len(vaos))
        assert rc == 0, rc
        self.tex_handles = [self.upload_texture(t["rgba8"], t.get("srgb", True)) for t in sc.textures]
        mats = sc.c_materials(self.tex_handles)
        assert self.L.orc_upload_materials(self.h, C.byref(mats), len(mats)) == 0
        self.set_scene(sc)

    def upload_texture(self, rgba8, srgb=True):
        img = np.ascontiguousarray(rgba8, dtype=np.uint8)
        handle = C.c_uint32()
        assert self.L.orc_upload_texture(self.h, _ptr(img), img.shape[1], img.shape[0], int(bool(srgb)), C.byref(handle)) == 0
        return handle.value

    def sample_texture(self, handle, uv):
        uv = np.ascontiguousarray(uv, dtype=np.float32)
        out = alloc_out((len(uv), 4), np.float32)
        assert self.L.orc_tex2d(self.h, handle, _ptr(uv), len(uv), _ptr(out)) == 0
        return out

    def set_scene(self, sc):
        objs, aabbs, lights = sc.c_objects(), sc.c_aabbs(), sc.c_lights()
        cam = sc.c_camera(self.W, self.H)
        w2c = np.ascontiguousarray(sc.transform_W2C(self.W, self.H), dtype=np.float32)
        sky = sc.c_skybox(getattr(self, "tex_handles", None))
        rc = self.L.orc_set_scene(self.h, C.byref(objs), C.byref(aabbs), len(sc.objects), C.byref(lights),
                                  len(sc.lights), C.byref(cam), _ptr(w2c), C.byref(sky) if sky else None)
        assert rc == 0

    def render_frame(self, settings):
        assert self.L.orc_render_frame(self.h, C.byref(settings)) == 0

    def reset_history(self):
        self.L.orc_reset_history(self.h)

    def read(self, name):
        bid = abi.BUF[name] if isinstance(name, str) else name
        ch = abi.BUF_CHANNELS[bid]
        shape = (self.H, self.W, ch)
        out = alloc_out((abi.BUF_PLANES[bid],) + shape if bid in abi.BUF_PLANES else shape, np.float32)
        rc = self.L.orc_read_buffer(self.h, bid, _ptr(out), out.size)
        assert rc == 0, rc
        return out

    def set_tie_order(self, mode):
        """Which of two triangles with equal split keys goes left in the BLAS build (render.hpp: tie_order); before load_scene."""
        assert self.L.orc_set_tie_order(self.h, int(mode)) == 0

    def trace_exhaustive(self, rays):
        """Closest hits by the reference's triangle test against every triangle, no boxes (the yardstick for disputed hits)."""
        return self.trace(rays, 2)

    def trace(self, rays, find_any=False):
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        hits = alloc_out(len(rays), abi.HIT_DTYPE)
        self.L.orc_trace_rays(self.h, _ptr(rays), len(rays), int(find_any), _ptr(hits))
        return hits

    def counters(self):
        c = abi.Counters()
        self.L.orc_get_counters(self.h, C.byref(c))
        return c.as_dict()

    def reset_counters(self):
        self.L.orc_reset_counters(self.h)

    def stage_seconds(self):
        arr = (C.c_double * len(abi.STAGES))()
        self.L.orc_get_stage_seconds(self.h, arr)
        return dict(zip(abi.STAGES, list(arr)))

    def reset_stage_seconds(self):
        self.L.orc_reset_stage_seconds(self.h)
