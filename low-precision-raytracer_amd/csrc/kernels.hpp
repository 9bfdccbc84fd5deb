// Host-callable launchers of the device kernels (one explicit instantiation per DataT).
#pragma once
#include "ctx.hpp"

namespace lprt {

// Every kernel that casts rays exists in two variants, compiled from the same sources into two namespaces:
//   fastslab  the product's default: slab parameters of half boxes in f32 (trace.hpp);
//   refslab   half instantiation only (LPRT_OPT_REFERENCE_SLAB): ray_aabb_intersection_* evaluated in float16 operation for
//             operation as rt/rtrt/cuda.hpp:70-149 does -- half subtraction, half division, half slack -- so that the fp16
//             parity tests can tell "the GPU culls less than the reference's lossy half test" from "the trees differ".
// The second variant is a translation unit of its own (pipeline_refslab.hip, trace_kernels_refslab.hip define
// LPRT_REFSLAB_TU and include the source), so the default kernels' code and register allocation do not change by a bit.
#ifdef LPRT_REFSLAB_TU
#define LPRT_SLAB_NS refslab
#else
#define LPRT_SLAB_NS fastslab
#endif
#define LPRT_VAR_BEGIN namespace LPRT_SLAB_NS {
#define LPRT_VAR_END }
#define LPRT_TRACE_LAUNCHERS                                                                                                                          \
    /* trace_kernels.hip; wide: the two-levels-per-step BLAS walk (trace.hpp) */                                                                     \
    template <typename T>                                                                                                                             \
    void launch_trace_rays(hipStream_t st, const SceneDev<T> &sc, const lprt_ray *rays, uint32_t n, bool any, bool wide, lprt_hit *hits,              \
                           Counters *counters);                                                                                                       \
    /* pipeline.hip */                                                                                                                                \
    template <typename T> void launch_fill_gbuffer(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, Counters *c);                       \
    /* the last round of a sample also runs write_clean_color; wide: deep BLAS trees, two levels per step */                                         \
    template <typename T>                                                                                                                             \
    void launch_trace_di(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, bool first, bool gi_on, bool add_direct, uint32_t slots,      \
                         bool reuse, Counters *c, bool wide);                                                                                         \
    template <typename T> void launch_trace_gi(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, uint32_t slots, Counters *c);           \
    /* ray streams (stream.hpp): persistent waves that refill finished lanes.  `counter` is a zeroed device word owned by the launch. */           \
    template <typename T>                                                                                                                             \
    /* order / n_sorted: the sorted item list of launch_ray_sort_* (null: the stream runs over its items in pixel-block order) */                \
    void launch_stream_gi(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, uint32_t slots, uint32_t *counter, Counters *c, bool wide,  \
                          const uint32_t *order, const uint32_t *n_sorted);                                                                           \
    template <typename T>                                                                                                                             \
    void launch_stream_di(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, bool first, uint32_t slots, uint32_t *counter, Counters *c,  \
                          bool wide, const uint32_t *order, const uint32_t *n_sorted);                                                                \
    template <typename T>                                                                                                                             \
    void launch_stream_rays(hipStream_t st, const SceneDev<T> &sc, const lprt_ray *rays, uint32_t n, bool any, bool wide, lprt_hit *hits,            \
                            uint32_t *counter, Counters *counters);

// trace_kernels.hip
template <typename T>
void launch_make_objrecs(hipStream_t st, const ObjSlot<T> *slots, const ObjConst<T> *objects, const BlasInfo *blas, const T *roots, uint32_t n,
                         uint32_t n_vaos, ObjRec<T> *out);
template <typename T> void launch_convert_materials(hipStream_t st, const lprt_material *in, MaterialT<T> *out, uint32_t n);
// up to LPRT_MAX_COPY_JOBS flat device-to-device byte copies in one launch (history-plane pack / unpack)
#define LPRT_MAX_COPY_JOBS 12
struct CopyJobs {
    const void *src[LPRT_MAX_COPY_JOBS];
    void *dst[LPRT_MAX_COPY_JOBS];
    size_t bytes[LPRT_MAX_COPY_JOBS];
    uint32_t n;
};
void launch_copy_jobs(hipStream_t st, const CopyJobs &jobs, size_t longest);
void launch_fmath(hipStream_t st, int op, const float *in, uint32_t n, float *out); // lprt_debug_fmath
void launch_sample_texture(hipStream_t st, const Texture *textures, const float *lut, uint32_t handle, const float *uv, uint32_t n, float *rgba);


// pipeline.hip
// A batch of samples runs through each per-sample stage in ONE launch (FrameDev: sample slots): slot z holds sample first + z.
#define LPRT_STREAM_COUNTER_WORDS 128u // per stream launch: work counter c of 8 at word 16 * c (stream.hpp)
struct SampleBatch {
    uint32_t first, count;                    // count <= LPRT_MAX_SAMPLE_BATCH
    uint32_t rng_d[LPRT_MAX_SAMPLE_BATCH];    // d word of each slot's shade seed
    uint32_t reuse_direct;                    // lprt_frame_settings.reuse_direct_light: round 0 runs for sample 0 only
};
namespace fastslab {
LPRT_TRACE_LAUNCHERS
}
namespace refslab { // only the h16 instantiations exist
LPRT_TRACE_LAUNCHERS
}
template <typename T> void launch_temporal_map(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, bool first, uint32_t rng_d, bool floor_taps);
template <typename T> void launch_shade(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, bool no_gi, const SampleBatch &sb); // round 0; round 1 runs inside k_trace_gi / as k_shade2
template <typename T> void launch_spp_mean(hipStream_t st, const FrameDev<T> &fb, uint32_t first, uint32_t count, uint32_t spp); // after a batch's last stage

// launch_stream_gi leaves a hit record per pixel for launch_shade2 (shade #2 of the hits); launch_stream_di leaves a
// visibility byte per pixel and light slot for launch_accum_di (accumulate_di_light, and in the last round of a sample
// write_clean_color) -- together what the fused per-pixel kernels do.
// Ray sort (pipeline.hip): the stream items that carry a ray, counting-sorted by (cell of the ray's origin on a 32^3 grid over
// the scene's bounds, Morton order | direction octant) -- `order` (item numbers) and *n_sorted for launch_stream_*.
// keys: one word per item; bins / offsets: LPRT_RAYSORT_BINS + 4 words each.  which: 0 GI rays, 1 / 2 shadow rays of round 0 / 1
// (keyed by light slot first: a run of the stream aims at one light).
#define LPRT_RAYSORT_BINS (1u << 18)
struct RaySortBufs {
    uint32_t *keys, *order, *bins, *n_sorted;
};
template <typename T> uint32_t ray_sort_items(const FrameDev<T> &fb, uint32_t n_lights, uint32_t slots, int which);
template <typename T> void launch_ray_sort(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, uint32_t slots, int which, const RaySortBufs &b);
template <typename T> void launch_shade2(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, uint32_t slots);
template <typename T> void launch_accum_di(hipStream_t st, const FrameDev<T> &fb, bool first, bool gi_on, bool add_direct, uint32_t slots, bool reuse);

// svgf.hip
template <typename T> void launch_depth_gradient(hipStream_t st, const FrameDev<T> &fb);
template <typename T> void launch_temporal_color(hipStream_t st, const FrameDev<T> &fb, float color_w, float moments_w, bool always_test); // + the moments of pixels with >= 4 frames of history
template <typename T> void launch_moments(hipStream_t st, const FrameDev<T> &fb, float moments_w);
template <typename T> void launch_wavelet(hipStream_t st, const FrameDev<T> &fb, int level, bool always_test, uint32_t row_lo, uint32_t row_hi);
template <typename T>
void launch_finalize(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, const V4<T> *gi_colored, const V4<T> *gi_white, V4<T> *taa_next,
                     bool add_colored, bool add_white, bool demodulate, float taa_w, float4 *ext_out, uint32_t ext_first_row, uint32_t ext_rows);
template <typename T> void launch_copy_last_transform(hipStream_t st, const SceneDev<T> &sc, M4<T> *last_L2W, uint32_t n_objects);
template <typename T> void launch_read_v4(hipStream_t st, const V4<T> *src, float *dst, uint32_t n, int channels);
void launch_read_u32(hipStream_t st, const uint32_t *src, float *dst, uint32_t n, bool rgba8);
void launch_read_u8(hipStream_t st, const uint8_t *src, float *dst, uint32_t n);

// rng.hip
struct RngStore {
    bool ready = false;
    DevBuf d_mats, d_w3, d_w2, d_w1;
    void init(hipStream_t st);
    // xorwow states of curand_init(seed_i, n, 0) for global pixels n in [first, first + count) -> planes[i][5][count], d_out[i]
    void generate(hipStream_t st, const uint64_t *seeds, uint32_t n_seeds, uint32_t total_pixels, uint32_t first, uint32_t count, uint32_t *planes,
                  uint32_t *d_out);
    // g_color (optional; honoured when the two ranges coincide): the window's G-buffer colour plane, pixel `first` at [0].  A run of
    // 64 pixel indices without a covered pixel (alpha 0) in the range gets no states: nothing reads them (shading and the TAA
    // jitter draw random numbers for covered pixels only), and a third of the Cornell frame is background.
    void generate(hipStream_t st, const uint64_t *seeds, uint32_t n_seeds, uint32_t n_head, uint32_t total_pixels, uint32_t first, uint32_t count,
                  uint32_t first2, uint32_t count2, uint32_t *planes, uint32_t *d_out, const uint32_t *g_color = nullptr);
    DevBuf d_runs; // [0]: number of runs with a covered pixel, [1 ..]: their indices (relative to the range's first run), any order
};

} // namespace lprt
