"""Host-side mirror of the reference's renderer interface over the C-ABI (include/lprt.h).

`Renderer` plays the role of `Renderer<DataT>` (rt/gui/imgui_window.cu:78-367): construction allocates the
per-pixel buffers, `set_rdresource` takes the loader's output, `render(rd_scene)` runs one frame.  There is
no CPU fallback: if csrc/liblprt_hip.so is missing or no GPU is present this module raises.
"""
import ctypes as C
import os
import sys
import weakref

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class LprtError(RuntimeError):
    pass


def library_path():
    # LPRT_LIBRARY: another build of the same sources (A/B runs of compile-time knobs, tools/build_variant.sh)
    return os.environ.get("LPRT_LIBRARY") or os.path.join(_HERE, "csrc", "liblprt_hip.so")


# every symbol include/lprt.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("lprt_create", C.c_int, [C.POINTER(_P), C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]),
    ("lprt_destroy", None, [_P]),
    ("lprt_last_error", C.c_char_p, [_P]),
    ("lprt_set_stream", C.c_int, [_P, _P]),
    ("lprt_synchronize", C.c_int, [_P]),
    ("lprt_upload_geometry", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32]),
    ("lprt_upload_materials", C.c_int, [_P, _P, C.c_uint32]),
    ("lprt_upload_texture", C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]),
    ("lprt_build_blas", C.c_int, [_P]),
    ("lprt_set_scene", C.c_int, [_P, _P, _P, C.c_uint32, _P, C.c_uint32, _P, _P, _P]),
    ("lprt_render_frame", C.c_int, [_P, _P]),
    ("lprt_render_phase", C.c_int, [_P, _P, C.c_int]),
    ("lprt_set_sample_rows", C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ("lprt_reset_history", C.c_int, [_P]),
    ("lprt_read_buffer", C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    ("lprt_output_device_ptr", C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    ("lprt_trace_rays", C.c_int, [_P, _P, C.c_uint32, C.c_int, _P]),
    ("lprt_enable_counters", C.c_int, [_P, C.c_int]),
    ("lprt_get_counters", C.c_int, [_P, _P]),
    ("lprt_get_stage_counters", C.c_int, [_P, C.c_int, _P]),
    ("lprt_reset_counters", C.c_int, [_P]),
    ("lprt_set_output_target", C.c_int, [_P, _P, C.c_uint32, C.c_uint32]),
    ("lprt_history_row_bytes", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_size_t)]),
    ("lprt_export_history_rows", C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, _P]),
    ("lprt_import_history_rows", C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, _P]),
    ("lprt_debug_sample_texture", C.c_int, [_P, C.c_uint32, _P, C.c_uint32, _P]),
    ("lprt_debug_fmath", C.c_int, [_P, C.c_int, _P, C.c_uint32, _P]),
    ("lprt_debug_rng_states", C.c_int, [_P, C.c_uint64, C.c_uint32, C.c_uint32, _P, C.c_size_t]),
    ("lprt_set_option", C.c_int, [_P, C.c_int, C.c_int]),
    ("lprt_get_option", C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    ("lprt_get_abi_info", None, [_P, C.c_uint32]),
    ("lprt_enable_stage_timing", C.c_int, [_P, C.c_int]),
    ("lprt_set_stage_timing_mask", C.c_int, [_P, C.c_uint32]),
    ("lprt_get_stage_times", C.c_int, [_P, _P, _P, C.c_uint32]),
    ("lprt_reset_stage_times", C.c_int, [_P]),
    ("lprt_get_blas_info", C.c_int, [_P, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("lprt_read_blas_order", C.c_int, [_P, C.c_uint32, _P, C.c_uint32]),
    ("lprt_read_blas_nodes", C.c_int, [_P, C.c_uint32, _P, _P, _P, C.c_uint32]),
    ("lprt_version", C.c_char_p, []),
]


def load_library():
    """dlopen the HIP library and bind every entry point; raises LprtError when it is not built."""
    global _LIB
    if _LIB is None:
        # This harness shares its process with PyTorch (device tensors, streams, torch.distributed), and PyTorch's wheel
        # carries its own libamdhip64.so / libhsa-runtime64.so with the same SONAMEs as /opt/rocm's.  Whichever HIP runtime
        # is loaded first serves liblprt_hip.so; loading ours first and torch later would leave TWO HIP and TWO HSA runtimes
        # driving one GPU from one process.  So torch goes first wherever it is installed (a C/C++ host links the system
        # runtime alone and is not affected).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        path = library_path()
        if not os.path.exists(path):
            raise LprtError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)")
        lib = C.CDLL(path)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        check_abi(lib)
        _LIB = lib
    return _LIB


def check_abi(lib):
    """Compare the loaded library's struct sizes and table lengths with this binding's mirror of include/lprt.h: an
    out-array or struct of the wrong size is an error at load time, never a stray write."""
    info = abi.AbiInfo()
    lib.lprt_get_abi_info(C.byref(info), C.sizeof(info))
    want = abi.expected_abi_info()
    got = {name: int(getattr(info, name)) for name, _ in abi.AbiInfo._fields_}
    bad = {k: (got[k], v) for k, v in want.items() if got[k] != v}
    if bad:
        raise LprtError(f"liblprt_hip.so does not match this binding's mirror of include/lprt.h (library, binding): {bad}")


_LIVE = weakref.WeakSet()  # renderers not yet closed (tests close what a test left open: no GC-time lprt_destroy)


def close_all():
    """Destroy every context that is still open.  Deterministic teardown for test harnesses and at-exit handlers."""
    for r in list(_LIVE):
        r.close()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _out(shape, dtype):
    """Host array the library writes into (its contents are fully overwritten, or it is zero-filled where the callers
    rely on that).  Test harnesses swap `alloc_out` for an allocator that ends the array at a PROT_NONE page
    (tests/guarded.py), so that a native write past the end faults at the write."""
    return alloc_out(shape, dtype)


def _zeros(shape, dtype):
    return np.zeros(shape, dtype=dtype)


alloc_out = _zeros


class Renderer:
    """One renderer instance for rows [row_begin, row_end) of a width x height frame on one GPU."""

    def __init__(self, width, height, precision=abi.F32, device=0, row_begin=0, row_end=None):
        self.L = load_library()
        self.W, self.H = int(width), int(height)
        self.row_begin = int(row_begin)
        self.row_end = self.H if row_end is None else int(row_end)
        self.rows = self.row_end - self.row_begin
        self.precision = precision
        self.h = C.c_void_p()
        rc = self.L.lprt_create(C.byref(self.h), device, self.W, self.H, self.row_begin, self.row_end, precision)
        if rc != abi.OK:
            msg = self.L.lprt_last_error(None).decode()
            self.h = None
            raise LprtError(f"lprt_create failed ({rc}): {msg}")
        self.scene = None
        _LIVE.add(self)

    # ---- plumbing -------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != abi.OK:
            raise LprtError(f"{what} failed ({rc}): {self.L.lprt_last_error(self.h).decode()}")

    def close(self):
        """lprt_destroy.  Call it (or use the renderer as a context manager): destruction then happens at a point the
        caller chose, with the HIP runtime and every stream the context was given still alive."""
        if getattr(self, "h", None):
            h, self.h = self.h, None
            _LIVE.discard(self)
            self.L.lprt_destroy(h)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        # Last resort only.  At interpreter shutdown the HIP runtime (and PyTorch, whose stream the context may have been
        # given) may already be gone: the process is exiting, so the device memory is left to the driver.
        if sys is None or sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        """Enqueue all work on this hipStream_t (e.g. torch.cuda.current_stream().cuda_stream)."""
        self._check(self.L.lprt_set_stream(self.h, C.c_void_p(int(hip_stream))), "lprt_set_stream")

    def synchronize(self):
        self._check(self.L.lprt_synchronize(self.h), "lprt_synchronize")

    # ---- loader side: Renderer::set_rdresource (imgui_window.cu:332-337) ----------------------------
    def set_rdresource(self, sc, build=True):
        self.scene = sc
        verts = np.ascontiguousarray(sc.vertices)
        idx = np.ascontiguousarray(sc.indices, dtype=np.uint32)
        vaos = sc.c_vaos()
        self._check(self.L.lprt_upload_geometry(self.h, _ptr(verts), len(verts), _ptr(idx), len(idx), C.byref(vaos),
                                                len(vaos)), "lprt_upload_geometry")
        self.tex_handles = [self.upload_texture(t["rgba8"], t.get("srgb", True)) for t in sc.textures]
        mats = sc.c_materials(self.tex_handles)
        self._check(self.L.lprt_upload_materials(self.h, C.byref(mats), len(mats)), "lprt_upload_materials")
        if build:
            self.build_blas()

    def build_blas(self):
        self._check(self.L.lprt_build_blas(self.h), "lprt_build_blas")

    def upload_texture(self, rgba8, srgb=True):
        """RDTexture(data, w, h, sRGB) (rt/rtrt/memory.hpp:334-369); returns the handle materials / the skybox name."""
        img = np.ascontiguousarray(rgba8, dtype=np.uint8)
        assert img.ndim == 3 and img.shape[2] == 4
        handle = C.c_uint32()
        self._check(self.L.lprt_upload_texture(self.h, _ptr(img), img.shape[1], img.shape[0], int(bool(srgb)), C.byref(handle)),
                    "lprt_upload_texture")
        return handle.value

    def sample_texture(self, handle, uv):
        """Device texture fetches at normalised coordinates uv[n, 2] -> float32 [n, 4] (test hook)."""
        uv = np.ascontiguousarray(uv, dtype=np.float32)
        out = _out((len(uv), 4), np.float32)
        self._check(self.L.lprt_debug_sample_texture(self.h, handle, _ptr(uv), len(uv), _ptr(out)), "lprt_debug_sample_texture")
        return out

    # ---- scene-manager side: the RDScene of this frame ------------------------------------------------
    def prepare_scene(self, sc=None):
        """The C arrays of an RDScene, built once: a caller that hands over the same scene every frame (bench.py) does
        not pay the Python loops again."""
        sc = sc or self.scene
        objs, aabbs, lights = sc.c_objects(), sc.c_aabbs(), sc.c_lights()
        cam = sc.c_camera(self.W, self.H)
        w2c = np.ascontiguousarray(sc.transform_W2C(self.W, self.H), dtype=np.float32)
        sky = sc.c_skybox(getattr(self, "tex_handles", None))
        return (objs, aabbs, len(sc.objects), lights, len(sc.lights), cam, w2c, sky)

    def set_scene(self, sc=None, prepared=None):
        objs, aabbs, n, lights, nl, cam, w2c, sky = prepared if prepared is not None else self.prepare_scene(sc)
        self._check(self.L.lprt_set_scene(self.h, C.byref(objs), C.byref(aabbs), n, C.byref(lights), nl, C.byref(cam), _ptr(w2c),
                                          C.byref(sky) if sky else None), "lprt_set_scene")

    def load_scene(self, sc):
        self.set_rdresource(sc)
        self.set_scene(sc)

    # ---- Renderer::render (imgui_window.cu:339-347) ----------------------------------------------------
    def render(self, settings=None):
        settings = settings or abi.default_settings()
        self._settings_keepalive = settings
        self._check(self.L.lprt_render_frame(self.h, C.byref(settings)), "lprt_render_frame")

    render_frame = render

    def render_phase(self, settings, phase):
        """One half of a frame (abi.PHASE_SAMPLE, then abi.PHASE_RESOLVE); sharded rendering exchanges rows in between."""
        self._settings_keepalive = settings
        self._check(self.L.lprt_render_phase(self.h, C.byref(settings), int(phase)), "lprt_render_phase")

    def set_sample_rows(self, first_row=0, n_rows=0):
        """Image rows the per-sample stages run on (n_rows = 0: the whole window)."""
        self._check(self.L.lprt_set_sample_rows(self.h, int(first_row), int(n_rows)), "lprt_set_sample_rows")

    def reset_history(self):
        self._check(self.L.lprt_reset_history(self.h), "lprt_reset_history")

    def read(self, name):
        bid = abi.BUF[name] if isinstance(name, str) else name
        shape = (self.rows, self.W, abi.BUF_CHANNELS[bid])
        out = _out((abi.BUF_PLANES[bid],) + shape if bid in abi.BUF_PLANES else shape, np.float32)
        self._check(self.L.lprt_read_buffer(self.h, bid, _ptr(out), out.size), "lprt_read_buffer")
        return out

    def output_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self.L.lprt_output_device_ptr(self.h, C.byref(p), C.byref(n)), "lprt_output_device_ptr")
        return p.value, n.value

    # ---- ray batches / instrumentation ---------------------------------------------------------------------
    def trace(self, rays, find_any=False):
        rays = np.ascontiguousarray(rays, dtype=abi.RAY_DTYPE)
        hits = _out(len(rays), abi.HIT_DTYPE)
        self._check(self.L.lprt_trace_rays(self.h, _ptr(rays), len(rays), int(find_any), _ptr(hits)), "lprt_trace_rays")
        return hits

    def enable_counters(self, on=True):
        self._check(self.L.lprt_enable_counters(self.h, int(on)), "lprt_enable_counters")

    def counters(self):
        c = abi.Counters()
        self._check(self.L.lprt_get_counters(self.h, C.byref(c)), "lprt_get_counters")
        return c.as_dict()

    def reset_counters(self):
        self._check(self.L.lprt_reset_counters(self.h), "lprt_reset_counters")

    def stage_counters(self, stage):
        c = abi.Counters()
        sid = abi.STAGES.index(stage) if isinstance(stage, str) else stage
        self._check(self.L.lprt_get_stage_counters(self.h, sid, C.byref(c)), "lprt_get_stage_counters")
        return c.as_dict()

    def set_output_target(self, dev_ptr, first_row, n_rows):
        self._check(self.L.lprt_set_output_target(self.h, C.c_void_p(int(dev_ptr) if dev_ptr else 0), first_row, n_rows),
                    "lprt_set_output_target")

    # ---- sharded frames over several frames: temporal history of halo rows (dist.ShardedFrame.exchange_history) ----
    def history_row_bytes(self, groups=abi.HIST_ALL):
        n = C.c_size_t()
        self._check(self.L.lprt_history_row_bytes(self.h, groups, C.byref(n)), "lprt_history_row_bytes")
        return n.value

    def export_history_rows(self, first_row, n_rows, dev_ptr, groups=abi.HIST_ALL):
        self._check(self.L.lprt_export_history_rows(self.h, groups, first_row, n_rows, C.c_void_p(int(dev_ptr))), "lprt_export_history_rows")

    def import_history_rows(self, first_row, n_rows, dev_ptr, groups=abi.HIST_ALL):
        self._check(self.L.lprt_import_history_rows(self.h, groups, first_row, n_rows, C.c_void_p(int(dev_ptr))), "lprt_import_history_rows")

    def fmath(self, op, x):
        """include/lprt_fmath.h as the device evaluates it (test hook): op in ('log', 'sin', 'cos', 'tan')."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        out = _out(x.shape, np.float32)
        self._check(self.L.lprt_debug_fmath(self.h, ("log", "sin", "cos", "tan").index(op), _ptr(x), x.size, _ptr(out)), "lprt_debug_fmath")
        return out

    def rng_states(self, seed, first, count):
        out = _out((count, abi.RNG_STATE_WORDS), np.uint32)
        self._check(self.L.lprt_debug_rng_states(self.h, int(seed), first, count, _ptr(out), out.size), "lprt_debug_rng_states")
        return out

    # ---- behaviour knobs (lprt_option): speed only, never results ------------------------------------------------------
    def set_option(self, name, value):
        self._check(self.L.lprt_set_option(self.h, abi.OPTIONS.index(name), int(value)), "lprt_set_option")

    def get_option(self, name):
        v = C.c_int()
        self._check(self.L.lprt_get_option(self.h, abi.OPTIONS.index(name), C.byref(v)), "lprt_get_option")
        return v.value

    def enable_stage_timing(self, on=True):
        self._check(self.L.lprt_enable_stage_timing(self.h, int(on)), "lprt_enable_stage_timing")

    def set_stage_timing_mask(self, stages):
        """Time only the named stages (an iterable of abi.STAGES names); empty turns timing off."""
        mask = 0
        for s in stages:
            mask |= 1 << abi.STAGES.index(s)
        self._check(self.L.lprt_set_stage_timing_mask(self.h, mask), "lprt_set_stage_timing_mask")

    def stage_times(self):
        ms = (C.c_float * len(abi.STAGES))()
        n = (C.c_uint32 * len(abi.STAGES))()
        self._check(self.L.lprt_get_stage_times(self.h, ms, n, len(abi.STAGES)), "lprt_get_stage_times")
        return {s: (float(ms[i]), int(n[i])) for i, s in enumerate(abi.STAGES)}

    def reset_stage_times(self):
        self._check(self.L.lprt_reset_stage_times(self.h), "lprt_reset_stage_times")

    def blas_info(self, vao):
        a, b, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._check(self.L.lprt_get_blas_info(self.h, vao, C.byref(a), C.byref(b), C.byref(c)), "lprt_get_blas_info")
        return dict(n_tris=a.value, n_nodes=b.value, depth=c.value)

    def blas_order(self, vao):
        n = self.blas_info(vao)["n_tris"]
        out = _out(n, np.uint32)
        self._check(self.L.lprt_read_blas_order(self.h, vao, _ptr(out), n), "lprt_read_blas_order")
        return out

    def blas_nodes(self, vao):
        """(root box [6], child boxes [n_nodes, 2, 6], leaf words [n_nodes, 2]) of the heap-ordered node records (include/lprt.h)."""
        n = self.blas_info(vao)["n_nodes"]
        root, boxes, words = _out(6, np.float32), _out((n, 2, 6), np.float32), _out((n, 2), np.uint32)
        self._check(self.L.lprt_read_blas_nodes(self.h, vao, _ptr(root), _ptr(boxes), _ptr(words), n), "lprt_read_blas_nodes")
        return root, boxes, words
