"""ctypes mirror of include/lprt.h (the C-ABI boundary).  Field order and sizes must match the header."""
import ctypes as C

MAX_DIRECT_LIGHT = 4
F32, F16 = 0, 1
LIGHT_SPOT, LIGHT_POINT, LIGHT_DIRECTIONAL = 0, 1, 2
OK = 0

f32 = C.c_float
u32 = C.c_uint32


class Vertex(C.Structure):  # RDVertex<float>, rt/rtrt/memory.hpp:26-41
    _fields_ = [("position", f32 * 3), ("normal", f32 * 3), ("tangent", f32 * 3), ("color", f32 * 3),
                ("uv0", f32 * 2), ("uv1", f32 * 2)]


class MeshVAO(C.Structure):  # CUDAMeshVAO, rt/rtrt/memory.hpp:560-566
    _fields_ = [("ebo_index", u32), ("vbo_index", u32), ("ebo_offset", u32), ("ebo_length", u32)]


class Material(C.Structure):  # CUDAMaterial, rt/rtrt/memory.hpp:303-332
    _fields_ = [("color", f32 * 3), ("emission", f32 * 3), ("metallic", f32), ("roughness", f32),
                ("double_sided", u32), ("uv_color", u32), ("tex_color", u32), ("reserved", u32)]


class Object(C.Structure):  # CUDAObjectConstantData, rt/rtrt/memory.hpp:279-301
    _fields_ = [("transform_L2W", f32 * 16), ("transform_W2L", f32 * 16), ("materialid", u32), ("objectid", u32)]


class AABB(C.Structure):
    _fields_ = [("lower", f32 * 3), ("upper", f32 * 3)]


class Light(C.Structure):  # RDLight, rt/rtrt/memory.hpp:43-51
    _fields_ = [("type", u32), ("inner_cone_angle", f32), ("outer_cone_angle", f32), ("maximum_distance", f32),
                ("direction", f32 * 3), ("intensity", f32 * 3), ("position", f32 * 3)]


class Camera(C.Structure):  # RDCamera, rt/rtrt/memory.hpp:382-405
    _fields_ = [("transform", f32 * 16), ("transform_L2W", f32 * 16), ("field_of_view_y", f32),
                ("aspect_ratio", f32), ("z_near", f32), ("z_far", f32)]


class Skybox(C.Structure):
    _fields_ = [("valid", u32), ("texture", u32), ("delta_x", f32), ("delta_y", f32), ("exposure", f32)]


class FrameSettings(C.Structure):  # Renderer::Settings + DemoSetting (+ spp, seeds)
    _fields_ = [("spp", u32), ("gi_on", u32), ("svgf", u32), ("add_direct_out", u32), ("add_gi_colored", u32),
                ("add_gi_white", u32), ("demodulate", u32), ("svgf_color_mix_weight", f32),
                ("svgf_moments_mix_weight", f32), ("taa_mix_weight", f32), ("seeds", C.POINTER(C.c_uint64)),
                ("rebuild_blas", u32), ("reuse_direct_light", u32)]


class Ray(C.Structure):
    _fields_ = [("source", f32 * 3), ("direction", f32 * 3), ("min_distance", f32), ("max_distance", f32),
                ("skip_vao", u32), ("skip_inobject_offset", u32)]


class Hit(C.Structure):
    _fields_ = [("hit", u32), ("inscene_objid", u32), ("inobject_offset", u32), ("t", f32), ("barycentric", f32 * 3)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "tlas_node_visits", "aabbw_tests", "object_visits",
                                         "blas_node_visits", "tri_tests", "fallback_tests")] + [("wave_steps", C.c_uint64 * 3),
                                                                                                ("lane_steps", C.c_uint64 * 3)]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_[:7]}
        d["wave_steps"], d["lane_steps"] = [int(v) for v in self.wave_steps], [int(v) for v in self.lane_steps]
        return d


HIST_FILTER, HIST_TAA = 1, 2   # groups of temporal-history planes (include/lprt.h)
HIST_ALL = HIST_FILTER | HIST_TAA
HIST_SAMPLES = 4               # this frame's per-lobe sample means, exchanged between the two render phases
PHASE_SAMPLE, PHASE_RESOLVE = 1, 2
BUF = dict(COLOR_INPROGRESS=0, MUL_GI_COLORED=1, MUL_GI_WHITE=2, INTENSITY0=3, INTENSITY1=4, ALBEDO=5, OUTPUT=6,
           G_POSITION=7, G_NORMAL_DEPTH=8, G_TANGENT=9, G_UV0UV1=10, G_COLOR=11, G_OBJID=12, G_OFFSET=13,
           RAW_GI_COLORED=14, RAW_GI_WHITE=15, RAW_COLOR=16, GI_DIRECTION=17, LIGHT_DIR_MAXT=18, LIGHT_MASK=19,
           SHADE_POSITION=20, SKIP=21)
BUF_CHANNELS = {0: 3, 1: 3, 2: 3, 3: 3, 4: 3, 5: 3, 6: 4, 7: 4, 8: 4, 9: 4, 10: 4, 11: 4, 12: 1, 13: 1, 14: 3, 15: 3,
                16: 3, 17: 4, 18: 4, 19: 1, 20: 3, 21: 2}
BUF_PLANES = {18: MAX_DIRECT_LIGHT}  # leading dimension of buffers that are [planes][rows][W][channels]
STAGES = ["build_blas", "build_tlas", "primary", "temporal_map", "rng", "shade1", "trace_di1", "trace_gi", "shade2",
          "trace_di2", "compose", "svgf_gradient", "svgf_temporal", "svgf_moments", "svgf_wavelet", "finalize", "backup", "ray_sort"]
# device kernel behind each stage id (the names rocprofv3 --kernel-trace reports)
STAGE_KERNELS = {"primary": "k_fill_gbuffer", "temporal_map": "k_temporal_map", "rng": "k_rng_expand", "shade1": "k_shade",
                 "trace_di1": "k_trace_di", "trace_gi": "k_trace_gi", "shade2": "k_shade2", "trace_di2": "k_trace_di",
                 "compose": "k_accum_di", "svgf_gradient": "k_depth_gradient", "svgf_temporal": "k_temporal_color",
                 "svgf_moments": "k_moments", "svgf_wavelet": "k_wavelet", "finalize": "k_finalize",
                 "backup": "k_copy_last_transform", "build_tlas": "k_build_tlas"}

# lprt_option, in enum order
OPTIONS = ["wide_walk", "tlas_mode", "axis_reject", "svgf_always_test", "blas_builder", "blas_local", "debug_sync", "ray_stream",
           "stream_wide_closest", "stream_refill_min", "stream_node_min", "stream_sched", "sample_batch",
           "floor_reprojection", "stream_xcd", "reference_slab", "ray_sort"]
RNG_STATE_WORDS = 6


class AbiInfo(C.Structure):  # lprt_abi_info
    _fields_ = [(n, u32) for n in ("info_bytes", "stage_count", "buffer_count", "option_count", "sizeof_vertex", "sizeof_mesh_vao",
                                   "sizeof_material", "sizeof_object", "sizeof_aabb", "sizeof_light", "sizeof_camera", "sizeof_skybox",
                                   "sizeof_frame_settings", "sizeof_ray", "sizeof_hit", "sizeof_counters", "rng_state_words")]


def expected_abi_info():
    """What this mirror of include/lprt.h believes; renderer.check_abi compares it with the loaded library."""
    return dict(info_bytes=C.sizeof(AbiInfo), stage_count=len(STAGES), buffer_count=len(BUF), option_count=len(OPTIONS),
                sizeof_vertex=C.sizeof(Vertex), sizeof_mesh_vao=C.sizeof(MeshVAO), sizeof_material=C.sizeof(Material),
                sizeof_object=C.sizeof(Object), sizeof_aabb=C.sizeof(AABB), sizeof_light=C.sizeof(Light), sizeof_camera=C.sizeof(Camera),
                sizeof_skybox=C.sizeof(Skybox), sizeof_frame_settings=C.sizeof(FrameSettings), sizeof_ray=C.sizeof(Ray),
                sizeof_hit=C.sizeof(Hit), sizeof_counters=C.sizeof(Counters), rng_state_words=RNG_STATE_WORDS)


# the same stages when GI / round-1 shadow rays run as ray streams (large scenes, LPRT_OPT_RAY_STREAM)
STREAM_STAGE_KERNELS = {"trace_gi": "k_stream_gi", "trace_di2": "k_stream_di"}

# numpy dtypes with identical layout (for bulk ray/hit arrays)
import numpy as np

RAY_DTYPE = np.dtype([("source", "<f4", 3), ("direction", "<f4", 3), ("min_distance", "<f4"), ("max_distance", "<f4"),
                      ("skip_vao", "<u4"), ("skip_inobject_offset", "<u4")])
HIT_DTYPE = np.dtype([("hit", "<u4"), ("inscene_objid", "<u4"), ("inobject_offset", "<u4"), ("t", "<f4"),
                      ("barycentric", "<f4", 3)])
VERTEX_DTYPE = np.dtype([("position", "<f4", 3), ("normal", "<f4", 3), ("tangent", "<f4", 3), ("color", "<f4", 3),
                         ("uv0", "<f4", 2), ("uv1", "<f4", 2)])
assert RAY_DTYPE.itemsize == C.sizeof(Ray) and HIT_DTYPE.itemsize == C.sizeof(Hit)
assert VERTEX_DTYPE.itemsize == C.sizeof(Vertex) == 64
assert len(BUF) == len(BUF_CHANNELS)


def default_settings(spp=1, gi_on=True, svgf=True, seeds=None):
    """Flags of SURVEY.md §8(d): gi on, svgf on, alpha_c = alpha_m = 0.1, taa weight 1."""
    fs = FrameSettings()
    fs.spp, fs.gi_on, fs.svgf = spp, int(gi_on), int(svgf)
    fs.add_direct_out = fs.add_gi_colored = fs.add_gi_white = 1
    fs.demodulate = 0
    fs.svgf_color_mix_weight = fs.svgf_moments_mix_weight = 0.1
    fs.taa_mix_weight = 1.0
    if seeds is not None:
        arr = (C.c_uint64 * len(seeds))(*[int(s) for s in seeds])
        fs._seed_keepalive = arr
        fs.seeds = C.cast(arr, C.POINTER(C.c_uint64))
    return fs
