/*
 * lprt.h — C ABI of the MI355X-native low-precision path tracer (hot path only).
 *
 * This is the drop-in boundary for the rendering hot path of
 * ioissss/Low-precision-raytracer (rt/trace + rt/rtrt). The reference has no
 * FFI; its "operator surface" is C++ templates launched from
 * rt/gui/imgui_window.cu:121-329. Every entry point below names the reference
 * symbol it replaces (paths relative to the reference root).
 *
 * Conventions:
 *   - matrices are row-major data[r][c] acting on column vectors
 *     (rt/rtrt/memory.hpp:285-293);
 *   - all scene inputs are float32 on the host; the device converts to the
 *     context's DataT (float or packed half) exactly where the reference
 *     converts (float16(float), round-to-nearest-even);
 *   - every function returns LPRT_OK (0) or a negative error code; the text of
 *     the last error is available through lprt_last_error(); no exceptions
 *     cross this boundary (reference: RT_CHECK_CUDA throws,
 *     rt/utils/cu_exception.cu:9-13);
 *   - one context per device; calls on one context must be serialised by the
 *     caller (reference: single GL thread, default stream).
 */
#ifndef LPRT_H
#define LPRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPRT_MAX_DIRECT_LIGHT 4u /* rt/gui/imgui_window.cu:81 */
#define LPRT_MAX_BOUNCES 2u      /* rt/gui/imgui_window.cu:356 */

enum {
    LPRT_OK = 0,
    LPRT_ERR_INVALID = -1,   /* bad argument / bad state */
    LPRT_ERR_DEVICE = -2,    /* HIP runtime error (text in lprt_last_error) */
    LPRT_ERR_NO_DEVICE = -3, /* no usable GPU: there is no CPU fallback */
    LPRT_ERR_LIMIT = -4      /* a documented capacity limit was exceeded */
};

typedef enum { LPRT_F32 = 0, LPRT_F16 = 1 } lprt_precision; /* DataT, rt/gui/imgui_window.cu:78-80 */

/* RDVertex<float>, rt/rtrt/memory.hpp:26-41 (packed, 64 B). */
typedef struct {
    float position[3];
    float normal[3];
    float tangent[3];
    float color[3];
    float uv0[2];
    float uv1[2];
} lprt_vertex;

/* CUDAMeshVAO, rt/rtrt/memory.hpp:560-566: one glTF primitive = one BLAS. */
typedef struct {
    uint32_t ebo_index; /* must be 0: the loader builds one shared EBO (rt/rtrt/loader.cu:797-802) */
    uint32_t vbo_index; /* must be 0 */
    uint32_t ebo_offset;
    uint32_t ebo_length;
} lprt_mesh_vao;

/* CUDAMaterial<float>, rt/rtrt/memory.hpp:303-332 (only the fields shade() reads). */
typedef struct {
    float color[3];
    float emission[3];
    float metallic;
    float roughness;
    uint32_t double_sided;
    uint32_t uv_color;  /* 0/1: sample tex_color with uv0/uv1; >=2: no texture (rt/rtrt/cuda.hpp:1851) */
    uint32_t tex_color; /* handle from lprt_upload_texture, ignored when uv_color >= 2 */
    uint32_t reserved;
} lprt_material;

/* CUDAObjectConstantData<float>, rt/rtrt/memory.hpp:279-301. */
typedef struct {
    float transform_L2W[16];
    float transform_W2L[16];
    uint32_t materialid;
    uint32_t objectid; /* VAO index */
} lprt_object;

/* world-space AABB of one object, RDScene::objects_aabb, rt/rtrt/memory.hpp:699. */
typedef struct {
    float lower[3];
    float upper[3];
} lprt_aabb;

/* RDLight<float>, rt/rtrt/memory.hpp:43-51. */
enum { LPRT_LIGHT_SPOT = 0, LPRT_LIGHT_POINT = 1, LPRT_LIGHT_DIRECTIONAL = 2 };
typedef struct {
    uint32_t type;
    float inner_cone_angle;
    float outer_cone_angle;
    float maximum_distance;
    float direction[3];
    float intensity[3];
    float position[3];
} lprt_light;

/* RDCamera<float>, rt/rtrt/memory.hpp:382-405. */
typedef struct {
    float transform[16];     /* world -> view */
    float transform_L2W[16]; /* camera local -> world */
    float field_of_view_y;
    float aspect_ratio;
    float z_near;
    float z_far;
} lprt_camera;

/* RDSkybox<float>::Ref, rt/rtrt/memory.hpp:664-690. */
typedef struct {
    uint32_t valid;
    uint32_t texture; /* handle from lprt_upload_texture */
    float delta_x;
    float delta_y;
    float exposure;
} lprt_skybox;

/* Renderer::Settings + DemoSetting, rt/gui/imgui_window.cu:94-100, rt/rtrt/cuda.hpp:2161-2168.
 * spp is a build-side extension; spp = 1 is one reference frame. */
typedef struct {
    uint32_t spp;
    uint32_t gi_on;
    uint32_t svgf;
    uint32_t add_direct_out;
    uint32_t add_gi_colored;
    uint32_t add_gi_white;
    uint32_t demodulate;
    float svgf_color_mix_weight;
    float svgf_moments_mix_weight;
    float taa_mix_weight;
    /* 3*spp seeds (A_s: temporal map / B_s: shade #1 / C_s: shade #2), replacing the
     * reference's three rand() calls per frame (rt/gui/imgui_window.cu:145,159,199).
     * NULL: derived from the frame counter as A = 3(f*spp+s)+1, B = A+1, C = A+2. */
    const uint64_t *seeds;
    uint32_t rebuild_blas; /* rebuild every BLAS on the GPU before the frame (config C4) */
    uint32_t reuse_direct_light; /* opt-in: the shadow rays of round 0 depend on the G-buffer and the lights only, so with
                                  * gi_on they are cast for sample 0 and reused by the other samples of the frame: the
                                  * image is bit-identical, frames/s goes up (atrium, 8 spp: +17 %) and the Mray/s of
                                  * executed rays goes down; 0 (default) casts every ray of every sample */
} lprt_frame_settings;

/* Ray / hit records for lprt_trace_rays (Ray, Intersection, SkipGeometry: rt/rtrt/cuda.hpp:38-57,187-190). */
typedef struct {
    float source[3];
    float direction[3];
    float min_distance;
    float max_distance;    /* use 1e5f for "no limit", as the reference does (cuda.hpp:624) */
    uint32_t skip_vao;     /* UINT32_MAX: skip nothing */
    uint32_t skip_inobject_offset;
} lprt_ray;

typedef struct {
    uint32_t hit; /* 0 = Intersection::NONE */
    uint32_t inscene_objid;
    uint32_t inobject_offset;
    float t;
    float barycentric[3];
} lprt_hit;

/* Work counters for the roofline's algorithmic bytes. */
typedef struct {
    uint64_t rays;
    uint64_t tlas_node_visits; /* distinct TLAS nodes whose box was tested */
    uint64_t aabbw_tests;      /* per-object world-box tests at TLAS leaves */
    uint64_t object_visits;    /* BLAS traversals entered */
    uint64_t blas_node_visits; /* distinct BLAS nodes whose box was tested */
    uint64_t tri_tests;
    uint64_t fallback_tests;   /* fp32 edge-zone re-tests (cuda.hpp:267-328) */
    /* lane occupancy of the traversal loops (instrumented kernels only): a wave-step is one pass of a wave through one phase
     * of the walk, a lane-step one lane taking part in it; lane_steps / (64 * wave_steps) = the share of lanes doing work.
     * phase 0: instance entry / ray retirement, 1: box tests of a node step, 2: triangle tests at a leaf */
    uint64_t wave_steps[3];
    uint64_t lane_steps[3];
} lprt_counters;

typedef enum {
    /* RTRTProcedureBuffer planes (rt/rtrt/cuda.hpp:1475-1634); all read back as float32 */
    LPRT_BUF_COLOR_INPROGRESS = 0, /* [rows][W][3] */
    LPRT_BUF_MUL_GI_COLORED = 1,   /* [rows][W][3] (after SVGF when svgf=1) */
    LPRT_BUF_MUL_GI_WHITE = 2,     /* [rows][W][3] */
    LPRT_BUF_INTENSITY0 = 3,       /* [rows][W][3] */
    LPRT_BUF_INTENSITY1 = 4,       /* [rows][W][3] */
    LPRT_BUF_ALBEDO = 5,           /* [rows][W][3] */
    LPRT_BUF_OUTPUT = 6,           /* [rows][W][4] RGBA32F display surface (cuda.hpp:2441-2455) */
    /* G-buffer planes (rt/rtrt/cuda.hpp:950-1012) */
    LPRT_BUF_G_POSITION = 7,       /* [rows][W][4] */
    LPRT_BUF_G_NORMAL_DEPTH = 8,   /* [rows][W][4] */
    LPRT_BUF_G_TANGENT = 9,        /* [rows][W][4] */
    LPRT_BUF_G_UV0UV1 = 10,        /* [rows][W][4] */
    LPRT_BUF_G_COLOR = 11,         /* [rows][W][4] RGBA8 widened to float 0..255 */
    LPRT_BUF_G_OBJID = 12,         /* [rows][W][1] exact integers as float */
    LPRT_BUF_G_OFFSET = 13,        /* [rows][W][1] */
    /* pre-denoise split lobes (averaged over spp), for SVGF input parity */
    LPRT_BUF_RAW_GI_COLORED = 14,  /* [rows][W][3] */
    LPRT_BUF_RAW_GI_WHITE = 15,    /* [rows][W][3] */
    LPRT_BUF_RAW_COLOR = 16,       /* [rows][W][3] color_inprogress before add_denoised_color */
    LPRT_BUF_GI_DIRECTION = 17,    /* [rows][W][4] last sample's GI ray direction (TraceGICommand), w = 1 where a ray was cast */
    /* the shadow-ray commands the LAST shade of the frame left (TraceLightCommand, rt/rtrt/cuda.hpp:1014-1032): round 1 of
     * the last sample with gi_on, round 0 otherwise.  Parity tests replay them through lprt_trace_rays and the oracle to
     * tell "a ray found something else" apart from "the arithmetic differs".  Whole-window contexts only (no sample rows). */
    LPRT_BUF_LIGHT_DIR_MAXT = 18,  /* [4 slots][rows][W][4] direction.xyz, maximum_t */
    LPRT_BUF_LIGHT_MASK = 19,      /* [rows][W][1] bit s set: slot s holds a command */
    LPRT_BUF_SHADE_POSITION = 20,  /* [rows][W][3] source of round 1's shadow rays (the GI hit); round 0's is G_POSITION */
    LPRT_BUF_SKIP = 21,            /* [rows][W][2] SkipGeometry of those rays: vao index, inobject_offset */
    LPRT_BUF_COUNT = 22
} lprt_buffer_id;

/* Per-kernel GPU time (HIP events on the context's stream) and work counters; one id per kernel family.
 * The reference times the same stages on the host around cudaDeviceSynchronize
 * (rt/gui/imgui_window.cu:86-91, labels at :154-280). */
typedef enum {
    LPRT_STAGE_BUILD_BLAS = 0,     /* all kernels of lprt_build_blas */
    LPRT_STAGE_BUILD_TLAS = 1,     /* k_build_tlas */
    LPRT_STAGE_PRIMARY = 2,        /* k_fill_gbuffer            "Primary Ray" */
    LPRT_STAGE_TEMPORAL_MAP = 3,   /* k_temporal_map            "Generate Temporal Map" */
    LPRT_STAGE_RNG = 4,            /* k_rng_expand x3 per seed  (curand_init in the reference's shade) */
    LPRT_STAGE_SHADE1 = 5,         /* k_shade<first>            "Pixel Shade #1" */
    LPRT_STAGE_TRACE_DI1 = 6,      /* k_trace_di<first>         "Trace DI Ray #1" + "Accumulate DI #1" */
    LPRT_STAGE_TRACE_GI = 7,       /* k_trace_gi ("Trace GI #1" + "Shade #2" of the hit, in registers) or k_stream_gi */
    LPRT_STAGE_SHADE2 = 8,         /* k_shade2 behind a GI ray stream; inside k_trace_gi otherwise (no launches of its own) */
    LPRT_STAGE_TRACE_DI2 = 9,      /* k_trace_di<second>        "Trace DI #2" + "Accumulate DI #2" + write_clean_color + spp mean */
    LPRT_STAGE_COMPOSE = 10,       /* k_accum_di behind a shadow-ray stream; inside the last k_trace_di of a sample otherwise */
    LPRT_STAGE_SVGF_GRADIENT = 11, /* k_depth_gradient */
    LPRT_STAGE_SVGF_TEMPORAL = 12, /* k_temporal_color */
    LPRT_STAGE_SVGF_MOMENTS = 13,  /* k_moments */
    LPRT_STAGE_SVGF_WAVELET = 14,  /* k_wavelet<1,2,4,8,16>: five launches per frame */
    LPRT_STAGE_FINALIZE = 15,      /* k_finalize                add_denoised_color + "TAA" + output + pixel ids */
    LPRT_STAGE_BACKUP = 16,        /* k_copy_last_transform     "Backup Temporal Data" */
    LPRT_STAGE_RAY_SORT = 17,      /* k_ray_keys + k_ray_bins_scan + k_ray_scatter in front of a ray stream (LPRT_OPT_RAY_SORT) */
    LPRT_STAGE_COUNT = 18
} lprt_stage_id;

typedef struct lprt_ctx lprt_ctx;

/* ---- lifetime ------------------------------------------------------------------------- */

/* Replaces Renderer<DataT>::Renderer(width, height, n_vaos) (rt/gui/imgui_window.cu:349-357):
 * allocates every per-pixel buffer of RTRTProcedureBuffer<DataT,4> + GBuffer + output surface
 * for image rows [row_begin, row_end) of a width x height frame (row_begin = 0, row_end =
 * height on one GPU; a row strip plus halo when frames are sharded). */
int lprt_create(lprt_ctx **out, int device, uint32_t width, uint32_t height, uint32_t row_begin,
                uint32_t row_end, lprt_precision precision);
void lprt_destroy(lprt_ctx *ctx);
const char *lprt_last_error(const lprt_ctx *ctx); /* ctx may be NULL: last create() error */

/* All work is enqueued on this hipStream_t (default: a stream owned by the context). */
int lprt_set_stream(lprt_ctx *ctx, void *hip_stream);
int lprt_synchronize(lprt_ctx *ctx);

/* ---- resources (loader side of the boundary) -------------------------------------------- */

/* Replaces RDResource<DataT>::{objects_vbo, objects_ebo, objects_vao, compute_m}
 * (rt/rtrt/memory.hpp:568-648, rt/rtrt/loader.cu:774-811) and
 * CUDARenderGIData::fill_objects_cuda_vao (rt/rtrt/cuda.hpp:758-764).
 * The device derives the DataT vertex copy, M_shift, M_shift_f32 and position_f32. */
int lprt_upload_geometry(lprt_ctx *ctx, const lprt_vertex *vbo, uint32_t n_vertices,
                         const uint32_t *ebo, uint32_t n_indices, const lprt_mesh_vao *vaos,
                         uint32_t n_vaos);

/* Replaces RDResource::materials + CUDARenderGIData::refresh_materials (rt/rtrt/cuda.hpp:749-755). */
int lprt_upload_materials(lprt_ctx *ctx, const lprt_material *materials, uint32_t n_materials);

/* Replaces RDTexture(data, w, h, sRGB) (rt/rtrt/memory.hpp:334-369): tightly packed RGBA8 rows, sampled with
 * normalised coordinates, wrap addressing, bilinear filtering and (srgb != 0) an sRGB decode per texel.  The handle
 * goes into lprt_material.tex_color / lprt_skybox.texture; upload textures before the materials that name them. */
int lprt_upload_texture(lprt_ctx *ctx, const uint8_t *rgba8, uint32_t width, uint32_t height,
                        uint32_t srgb, uint32_t *out_handle);

/* Replaces ObjectBVH<DataT>(positions, indices) for every primitive (rt/trace/object_bvh.hpp:49-63,
 * called at rt/rtrt/loader.cu:434) + fill_objects_bvh (rt/rtrt/cuda.hpp:767-773): GPU build, level by level
 * (widest-axis median split of every segment by an LSD radix sort of centroid keys, implicit heap-ordered
 * hierarchy, bottom-up boxes). */
int lprt_build_blas(lprt_ctx *ctx);

/* ---- per-frame scene (scene-manager side of the boundary) --------------------------------- */

/* Replaces RDScene<DataT> hand-over + CUDARenderGIData::get_render_input
 * (rt/rtrt/cuda.hpp:662-710): uploads objects/lights/camera and rebuilds the TLAS on the GPU
 * (SceneBVH::update, rt/trace/scene_bvh.hpp:50-70). transform_W2C is
 * render_data->transform_W2C (rt/gui/imgui_window.cu:293-294), used only for reprojection. */
int lprt_set_scene(lprt_ctx *ctx, const lprt_object *objects, const lprt_aabb *aabbs,
                   uint32_t n_objects, const lprt_light *lights, uint32_t n_lights,
                   const lprt_camera *camera, const float transform_W2C[16],
                   const lprt_skybox *skybox);

/* Replaces Renderer::render = render_gbuffer (traced primary rays, rt/rtrt/cuda.hpp:854-872)
 * + render_cuda (rt/gui/imgui_window.cu:121-283). Asynchronous on the context's stream. */
int lprt_render_frame(lprt_ctx *ctx, const lprt_frame_settings *settings);

/* The same frame in two calls, for sharded rendering: LPRT_PHASE_SAMPLE runs render_gbuffer over the window and the spp
 * loop of render_cuda (RNG, shade, shadow/GI rays, write_clean_color; imgui_window.cu:169-238) over the sample rows;
 * LPRT_PHASE_RESOLVE runs the temporal maps, svgf_denoise, add_denoised_color, TAA and the output write
 * (imgui_window.cu:141-167, 240-283) over the window.  lprt_render_frame = both.  Between the two, a context whose sample
 * rows are narrower than its window imports the sample means of its other rows (LPRT_HIST_SAMPLES) from the contexts
 * that traced them. */
#define LPRT_PHASE_SAMPLE 1
#define LPRT_PHASE_RESOLVE 2
int lprt_render_phase(lprt_ctx *ctx, const lprt_frame_settings *settings, int phase);

/* Image rows [first_row, first_row + n_rows) of the window on which LPRT_PHASE_SAMPLE runs the per-sample stages;
 * n_rows = 0 (the default): the whole window.  With the rows a rank owns here, the halo rows cost no rays: their sample
 * means arrive from the neighbours (RNG streams are keyed by the global pixel index, so they are the values this
 * context would have computed). */
int lprt_set_sample_rows(lprt_ctx *ctx, uint32_t first_row, uint32_t n_rows);

/* Drops temporal history (frame counter back to 0): the state right after Renderer construction. */
int lprt_reset_history(lprt_ctx *ctx);

/* ---- results ------------------------------------------------------------------------------ */

/* Synchronises, converts to float32 and copies rows [row_begin,row_end) of one buffer. */
int lprt_read_buffer(lprt_ctx *ctx, lprt_buffer_id id, float *dst, size_t n_floats);

/* Device pointer to the RGBA32F output rows (tex_draw_target, rt/gui/imgui_window.cu:351):
 * [row_end-row_begin][width][4] floats, valid until lprt_destroy. */
int lprt_output_device_ptr(lprt_ctx *ctx, void **ptr, size_t *bytes);

/* Sharded frames: besides its own output plane the context also writes image rows
 * [first_row, first_row + n_rows) of the final RGBA32F image to dst (row-major, width*4 floats per row),
 * e.g. this rank's slot of the tensor handed to the RCCL all-gather.  dst = NULL turns it off.
 *
 * CONTRACT while a target is set: the target rows are what this context is asked for, so the a-trous levels of the
 * denoiser run only on the rows that can still influence them (level l within 64 - 2^(l+2) rows of the target strip).
 * Outside that reach the denoised planes of the window (LPRT_BUF_MUL_GI_*, LPRT_BUF_OUTPUT and the level-1 output that
 * serves as the NEXT frame's colour history) keep what earlier frames left there: only the TARGET rows of those planes
 * and of the context's own output plane are valid, and before the next frame the caller must import LPRT_HIST_FILTER for
 * every window row outside the target strip from the context that owns it (lprt_import_history_rows; dist.py does so
 * every frame).  A caller that sets a target without that exchange gets stale history on its halo rows. */
int lprt_set_output_target(lprt_ctx *ctx, void *dst_device, uint32_t first_row, uint32_t n_rows);

/* Sharded frames over several frames: the per-pixel state frame n+1 reads of frame n (SVGF colour and moment history
 * of both lobes, TAA history, frame counters) is only right on the rows a context OWNS; its halo rows lack part of
 * their filter support.  After each frame a rank exports the owned rows its neighbours hold as halo and imports its
 * own halo rows from their owners.  Rows are image rows inside the context's window; the packed buffer is device
 * memory of n_rows * lprt_history_row_bytes(groups) bytes (plane after plane of the chosen groups), one kernel on the
 * context's stream. */
#define LPRT_HIST_FILTER 1u /* SVGF colour + moment history of both lobes, frame counters: needed on the whole halo */
#define LPRT_HIST_TAA 2u    /* TAA history: read only within +-1 row (+ motion) of the rows a context owns */
#define LPRT_HIST_SAMPLES 4u /* not history: this frame's per-lobe sample means (the denoiser's input), exchanged between
                              * LPRT_PHASE_SAMPLE and LPRT_PHASE_RESOLVE when halo rows are not traced locally */
int lprt_history_row_bytes(lprt_ctx *ctx, uint32_t groups, size_t *bytes);
int lprt_export_history_rows(lprt_ctx *ctx, uint32_t groups, uint32_t first_row, uint32_t n_rows, void *dst_device);
int lprt_import_history_rows(lprt_ctx *ctx, uint32_t groups, uint32_t first_row, uint32_t n_rows, const void *src_device);

/* cast_ray<find_any,false> over a host ray batch (rt/rtrt/cuda.hpp:622-631), for hit-record parity
 * and ray-rate measurement; rays are rounded to DataT on the device like a TraceGICommand. */
int lprt_trace_rays(lprt_ctx *ctx, const lprt_ray *rays, uint32_t n_rays, int find_any,
                    lprt_hit *hits);

/* Counters are collected only while enabled (instrumented kernel variants). lprt_get_counters sums all
 * stages (ray batches of lprt_trace_rays are booked under LPRT_STAGE_PRIMARY). */
int lprt_enable_counters(lprt_ctx *ctx, int enable);
int lprt_get_counters(lprt_ctx *ctx, lprt_counters *out);
int lprt_get_stage_counters(lprt_ctx *ctx, lprt_stage_id stage, lprt_counters *out);
int lprt_reset_counters(lprt_ctx *ctx);

/* Per-stage HIP-event timing (ms, accumulated since the last reset) and launch counts. */
int lprt_enable_stage_timing(lprt_ctx *ctx, int enable);
/* Only the stages whose bit (1u << lprt_stage_id) is set are bracketed by events; 0 turns timing off.  Two event
 * records per launch cost about 4 us of stream time each on this stack, so a measurement that wants the frame rate
 * AND one kernel's duration instruments that kernel only. */
int lprt_set_stage_timing_mask(lprt_ctx *ctx, uint32_t stage_mask);
/* ms / launches: arrays of n entries each (either may be NULL); n must equal LPRT_STAGE_COUNT of the library that is
 * loaded (lprt_get_abi_info), otherwise LPRT_ERR_INVALID and nothing is written. */
int lprt_get_stage_times(lprt_ctx *ctx, float *ms, uint32_t *launches, uint32_t n);
int lprt_reset_stage_times(lprt_ctx *ctx);

/* Acceleration-structure read-back (tests): n_nodes = 2^depth - 1 heap slots of node records (children of slot i at 2i,
 * 2i + 1); tri_order = the primitive's triangles in leaf order; per slot both child boxes (boxes12 =
 * [n_nodes][2][lo xyz, hi xyz]) and both leaf words ([n_nodes][2]: 0xffffffff = the child is an inner node with a record of
 * its own, else (first << 3) | count of the triangles the child holds as a leaf, in leaf order).  Slots the builder did not
 * reach hold unspecified data: walk the tree from slot 1. */
int lprt_get_blas_info(lprt_ctx *ctx, uint32_t vao_index, uint32_t *n_triangles, uint32_t *n_nodes,
                       uint32_t *depth);
int lprt_read_blas_order(lprt_ctx *ctx, uint32_t vao_index, uint32_t *tri_order, uint32_t n);
int lprt_read_blas_nodes(lprt_ctx *ctx, uint32_t vao_index, float *root6, float *boxes12, uint32_t *leaf_words2, uint32_t n_nodes);

/* Test hook: the texture fetch shade() makes (normalised coordinates, wrap, bilinear with 1.8 fixed-point weights,
 * optional sRGB decode) at uv[n][2]; rgba = [n][4] floats. */
int lprt_debug_sample_texture(lprt_ctx *ctx, uint32_t handle, const float *uv, uint32_t n, float *rgba);

/* Test hook: xorwow state words v[0..4], d of curand_init(seed, n, 0) for global pixels n in
 * [first, first + count), as the frame kernels receive them; out = [count][6] uint32. */
int lprt_debug_rng_states(lprt_ctx *ctx, uint64_t seed, uint32_t first, uint32_t count, uint32_t *out,
                          size_t out_words /* must be 6 * count */);

/* Test hook: the single-precision transcendentals that shape rays (include/lprt_fmath.h) as the DEVICE evaluates them:
 * op 0 logf, 1 sinf, 2 cosf, 3 tanf; out[i] = f(in[i]).  The parity tests compare them bit for bit with the host's
 * evaluation of the same header. */
int lprt_debug_fmath(lprt_ctx *ctx, int op, const float *in, uint32_t n, float *out);

const char *lprt_version(void);

/* ---- behaviour knobs --------------------------------------------------------------------- */

/* Every A/B switch of the kernels is a per-context option.  The initial values come from the environment ONCE, in
 * lprt_create (variable named beside each option); nothing reads the environment after that, so a frame's behaviour cannot
 * change under the caller's feet.  Results never depend on these (the parity tests hold that); speed does -- with the one
 * exception of LPRT_OPT_REFERENCE_SLAB, a parity instrument that says so below. */
typedef enum {
    LPRT_OPT_WIDE_WALK = 0,        /* two BLAS levels per traversal step.  -1 (default): any-hit rays when some BLAS tree is
                                    * >= 6 levels deep; 0: never; 1: any-hit rays always.  lprt_trace_rays follows the same
                                    * rule for find_any != 0; with 1 its closest-hit batches take the two-level walk too
                                    * (hit-record parity of that walk).  LPRT_WIDE_WALK */
    LPRT_OPT_TLAS_MODE = 1,        /* -1 (default): flat scan for <= 32 instances, tree otherwise; 0: tree; 1: flat scan
                                    * (only honoured for <= 32 instances).  LPRT_TLAS_MODE */
    LPRT_OPT_AXIS_REJECT = 2,      /* 1 (default): stationary / overflowed-axis box rejections; 0: the reference's plain
                                    * slab test.  LPRT_NO_AXIS_REJECT (set = 0) */
    LPRT_OPT_SVGF_ALWAYS_TEST = 3, /* 0 (default); 1: SVGF window sums always take the per-sample finite tests.
                                    * LPRT_SVGF_ALWAYS_TEST (set = 1) */
    LPRT_OPT_BLAS_BUILDER = 4,     /* -1 (default): the surface-area builder when some primitive has >= 64 triangles, else the median
                                    * split; 0: level-by-level median split (the reference's rule, object_bvh.hpp:76-134); 1: one
                                    * Morton sort; 2: binned surface-area heuristic, depth-bounded, leaves of <= 2 triangles at any
                                    * depth (csrc/build.hip: k_sah_level).  Hit records never depend on it except at exact-t ties
                                    *  and, in half precision, where box culling depends on the tree.
                                    * LPRT_BLAS_BUILDER=median|morton|sah */
    LPRT_OPT_BLAS_LOCAL = 5,       /* 1 (default): subtrees of <= 1024 triangles are ordered in LDS; 0: global passes only.
                                    * LPRT_BLAS_LOCAL */
    LPRT_OPT_DEBUG_SYNC = 6,       /* 1: synchronise and report after every stage.  LPRT_DEBUG_SYNC (set = 1) */
    LPRT_OPT_RAY_STREAM = 7,       /* GI and shadow rays as ray streams: persistent waves that refill finished lanes from a
                                    * work counter (csrc/stream.hpp), the per-pixel stages behind them as kernels of their own.
                                    * -1 (default): GI rays and the shadow rays of round 1 when some BLAS tree is >= 6 levels
                                    * deep (large scenes: incoherent rays of very different length; the coherent shadow rays
                                    * of round 0 keep the fused per-pixel kernel); 0: never; 1: always, round 0 included.
                                    * lprt_trace_rays follows the same rule.  LPRT_RAY_STREAM */
    LPRT_OPT_STREAM_WIDE_CLOSEST = 8, /* closest-hit streams walk two BLAS levels per step as well (any-hit streams follow
                                    * LPRT_OPT_WIDE_WALK).  LPRT_STREAM_WIDE_CLOSEST */
    LPRT_OPT_STREAM_REFILL_MIN = 9, /* 1..64 (default 12): idle lanes a stream wave collects before it hands out parked rays */
    LPRT_OPT_STREAM_NODE_MIN = 10,  /* 1..64 (default 20): lanes still descending below which the node loop yields */
    LPRT_OPT_STREAM_SCHED = 11,     /* 0 (default): a stream wave runs every phase of the walk in every pass; 1: one phase per pass,
                                     * the one most of its lanes wait for (measured slower: 47.3 vs 44.5 ms on the atrium, the
                                     * lanes are not waiting for each other's phases but for the stream's tail).  LPRT_STREAM_SCHED */
    LPRT_OPT_SAMPLE_BATCH = 12,     /* 1..8 (default 8): samples of a frame that run through each per-sample stage in one launch
                                     * (the samples are independent up to the spp mean, which is summed in sample order either way);
                                     * 1 = one launch per sample and stage, the reference's loop order.  LPRT_SAMPLE_BATCH */
    LPRT_OPT_FLOOR_REPROJECTION = 13, /* 0 (default): the reference's reprojection taps -- the fractional history position is
                                     * truncated TOWARD ZERO (cuda.hpp:2300-2306, 2347-2353), so a position a fraction of a pixel
                                     * outside the left / top image border gets the tap weights (1 + e, -e): an extrapolation.  In
                                     * half precision the position of a static pixel is off by up to W * 2^-11 pixels, the
                                     * border pixels extrapolate their own history frame after frame, and after ~15 frames the
                                     * filtered radiance there turns negative, grows by 1.3-1.5x per frame and the output pass
                                     * writes NaN (the oracle reproduces it pixel for pixel).  1: floor() -- such a
                                     * position keeps the one tap inside the image at weight 1.  Every pixel whose history
                                     * position is not negative is unaffected.  LPRT_FLOOR_REPROJECTION */
    LPRT_OPT_STREAM_XCD = 14,       /* 1 (default): a ray stream's items are cut into 8 contiguous ranges (image bands) with a work
                                     * counter each; workgroups that share an XCD (equal blockIdx % 8) drain one range first -- rays
                                     * that start next to each other meet in one XCD's L2 -- and steal from the other ranges when
                                     * theirs is empty.  0: one counter, batches go to whichever wave asks next.  Results never
                                     * depend on it.  LPRT_STREAM_XCD */
    LPRT_OPT_REFERENCE_SLAB = 15,   /* 0 (default): the slab parameters of half boxes are evaluated in f32 (difference in half, product
                                     * with the f32 reciprocal, min/max and slack in f32): nothing overflows, and the GPU never culls a
                                     * box the ray reaches.  1: ray_aabb_intersection_scene / _object evaluated in float16 operation for
                                     * operation as rt/rtrt/cuda.hpp:70-149 (half subtraction, half DIVISION, half slack products; a
                                     * parameter beyond 65504 drops its axis) -- the reference's own lossy test, several times slower,
                                     * for parity work: it separates "the GPU culls less than the reference" from "the two trees
                                     * differ".  Unlike every other option this one CHANGES half-precision results (which boxes a ray
                                     * is admitted to); single precision is unaffected (its default arithmetic is the reference's).
                                     * LPRT_REFERENCE_SLAB */
    LPRT_OPT_RAY_SORT = 16,         /* the order a ray stream hands its rays out in (csrc/pipeline.hip "ray sort"): the stream items that
                                     * carry a ray, counting-sorted on the device by (light slot for shadow rays, Morton code of the
                                     * origin's cell on a 32^3 grid over the scene, direction octant), so that the 64 rays of a batch
                                     * start in one cell and head the same way.  Bit 0: GI rays, bit 1: shadow rays that run as a stream;
                                     * 0 / -1 (default): pixel-block order -- measured on the atrium the sorted order buys 0 % (GI rays) and
                                     * 11 % (round-1 shadow rays) of the traversal time and the sort costs several times that
                                     * (DESIGN.md 4.1).  Results never depend on it.  LPRT_RAY_SORT */
    LPRT_OPT_COUNT = 17
} lprt_option;
int lprt_set_option(lprt_ctx *ctx, lprt_option option, int value);
int lprt_get_option(lprt_ctx *ctx, lprt_option option, int *value);
/* One more switch exists as an environment variable only (read in lprt_create, like the initial option values): LPRT_RNG_ALL_RUNS
 * (set = 1) expands random states for every pixel of the window.  The default expands none for the runs of 64 pixel indices that
 * hold no covered pixel of the frame's G-buffer: nothing reads them (rt/rtrt/cuda.hpp:1763-1764 initialises a state per
 * thread, but shade() and the TAA jitter draw numbers for covered pixels only), so frames do not depend on it
 * (tests/test_gpu_frame.py::test_random_states_of_background_runs_are_never_read).  LPRT_RNG_BESIDE_BUILD=0: a frame with
 * rebuild_blas expands its random states after the build instead of beside it on the context's second stream. */

/* Sizes and counts of the LOADED library, for bindings that mirror this header in another language: a binding compares
 * them with its own mirror at load time, so that an out-array or struct of the wrong size is an error at start-up and
 * never a stray write.  info_bytes = sizeof(lprt_abi_info) as the caller knows it; the library fills min(info_bytes, its
 * own size) bytes. */
typedef struct {
    uint32_t info_bytes;
    uint32_t stage_count, buffer_count, option_count;
    uint32_t sizeof_vertex, sizeof_mesh_vao, sizeof_material, sizeof_object, sizeof_aabb, sizeof_light, sizeof_camera,
        sizeof_skybox, sizeof_frame_settings, sizeof_ray, sizeof_hit, sizeof_counters;
    uint32_t rng_state_words; /* uint32 words per pixel written by lprt_debug_rng_states */
} lprt_abi_info;
void lprt_get_abi_info(lprt_abi_info *out, uint32_t info_bytes);

#ifdef __cplusplus
}
#endif
#endif /* LPRT_H */
