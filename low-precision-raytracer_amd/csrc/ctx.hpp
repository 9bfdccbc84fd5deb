// Host-side context: owns every device allocation of one renderer instance.
// Replaces the owners behind the reference's operator surface: RDResource (memory.hpp:568-648),
// CUDARenderGIData (cuda.hpp:635-792), RTRTProcedureBuffer + SVGFBuffer (cuda.hpp:1117-1166, 1475-1634),
// GBuffer (memory.hpp:155-253) and the output surface (imgui_window.cu:349-357).
#pragma once
#include "types.hpp"
#include <hip/hip_runtime.h>
#include <string>
#include <vector>

namespace lprt {

struct HipError {
    hipError_t code;
    const char *what;
    const char *file;
    int line;
};
#define LPRT_HIP(expr)                                                                                                 \
    do {                                                                                                               \
        hipError_t _e = (expr);                                                                                        \
        if (_e != hipSuccess)                                                                                          \
            throw ::lprt::HipError{_e, #expr, __FILE__, __LINE__};                                                     \
    } while (0)

// plain device allocation that can grow; contents are not preserved on growth
struct DevBuf {
    void *ptr = nullptr;
    size_t bytes = 0;
    void reserve(size_t n) {
        if (n <= bytes)
            return;
        release();
        LPRT_HIP(hipMalloc(&ptr, n));
        bytes = n;
    }
    void release() {
        if (ptr)
            (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    template <typename U> U *as() const { return static_cast<U *>(ptr); }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
};

// pinned host staging for small per-frame uploads: acquire() waits for the copies issued from it last time (normally long
// finished), release() marks the point on the stream after which it may be overwritten
struct PinnedStage {
    void *ptr = nullptr;
    size_t bytes = 0;
    hipEvent_t ev = nullptr;
    bool in_flight = false;
    void acquire(size_t n) {
        if (in_flight) {
            LPRT_HIP(hipEventSynchronize(ev));
            in_flight = false;
        }
        if (n > bytes) {
            if (ptr)
                (void)hipHostFree(ptr);
            ptr = nullptr;
            LPRT_HIP(hipHostMalloc(&ptr, n, hipHostMallocDefault));
            bytes = n;
        }
    }
    void release(hipStream_t st) {
        if (!ev)
            LPRT_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        LPRT_HIP(hipEventRecord(ev, st));
        in_flight = true;
    }
    ~PinnedStage() {
        if (ev)
            (void)hipEventDestroy(ev);
        if (ptr)
            (void)hipHostFree(ptr);
    }
    PinnedStage() = default;
    PinnedStage(const PinnedStage &) = delete;
    PinnedStage &operator=(const PinnedStage &) = delete;
};

struct StageTimer {
    bool enabled = false;
    uint32_t mask = 0xffffffffu; // bit s set: stage s is bracketed by events
    float ms[LPRT_STAGE_COUNT] = {};
    uint32_t launches[LPRT_STAGE_COUNT] = {};
    struct Pending {
        int stage;
        hipEvent_t a, b;
    };
    std::vector<Pending> pending;
    std::vector<hipEvent_t> pool;
};

// per-pixel planes of one SVGF lobe (SVGFBuffer, cuda.hpp:1117-1166); colour and variance travel
// together as one 4-vector {r,g,b,var} so a filter tap is a single 16-B/8-B load
template <typename T> struct SvgfLobe {
    V4<T> *cv_a;   // colour in/out (the mul_gi_* plane) + var
    V4<T> *cv_b;   // color_history (level-1 output; next frame's temporal history) + var
    V4<T> *cv_c;   // color_buffer + var
    T *illum;      // illum_buffer
    T *var_a, *var_b, *var_c; // the .w of cv_a/b/c once more as planes of their own: the 3x3 variance prefilter of the
                              // a-trous levels reads 9 image neighbours, and 4-B loads cost a quarter of the cache lines
    T *miu1_hist, *miu2_hist; // miu1_1 / miu2_1
    T *miu1_cur, *miu2_cur;   // miu1_2 / miu2_2
};

template <typename T> struct TemporalMapT { // TemporalMap<DataT>, cuda.hpp:1055-1065 (26 B f16 / 36 B f32 -> padded)
    T weights[4];
    uint16_t pos[4][2];
    uint32_t frame_count;
};

// what a finished closest-hit ray of a ray stream leaves for the per-pixel stage that follows (k_shade2): Intersection<DataT>, packed
template <typename T> struct alignas(8) GiHit { // 24 B (f32) / 16 B (f16)
    uint32_t objid_hit; // bit 31: hit; low bits: inscene_objid
    uint32_t inobject_offset;
    T bx, by, bz, t;
};

// every per-pixel work buffer (RTRTProcedureBuffer<DataT,4>::Ref, cuda.hpp:1538-1561) for the window
template <typename T> struct FrameDev {
    Window win;
    // G-buffer (7 surfaces, cuda.hpp:153-159)
    V4<T> *g_position, *g_normal_depth, *g_tangent, *g_uv;
    uint32_t *g_color, *g_objid, *g_offset;
    // shade <-> trace hand-over
    V4<T> *gi_multiplier;   // gi_material_multiplier[0] (w: unused)
    V4<T> *intensity0, *intensity1;
    uint2 *skips;
    V4<T> *gi_direction;    // TraceGICommand (x/y implicit: the pixel itself; the source is the G-buffer position)
    uint8_t *gi_valid;
    // light commands, SoA by slot: slot s of pixel p at [s * npix + p]
    V4<T> *lc_dir_maxt;     // direction.xyz, maximum_t
    V4<T> *lc_multiplier;   // material_multiplier.xyz
    uint8_t *lc_mask;       // bit s set: slot s active (slot != 255)
    // PixelShaderInput (cuda.hpp:1034-1048) lives in registers between trace_gi and shade #2; only the hit position is
    // kept, as the source of round 1's shadow rays
    V4<T> *sc_position;
    // filterable cache
    V4<T> *albedo, *color_inprogress, *taa_history;
    float4 *acc_color, *acc_gi_colored, *acc_gi_white; // spp accumulators (f32)
    V4<T> *raw_color, *raw_gi_colored, *raw_gi_white;  // pre-denoise copies for parity read-back
    SvgfLobe<T> lobe[2];
    V2<T> *depth_gradient;
    TemporalMapT<T> *svgf_map, *taa_map;
    uint8_t *valid_frame_count, *prev_frame_count;
    uint2 *last_frame_id;
    M4<T> *last_frame_L2W;
    M4<T> last_frame_W2C;
    float4 *output; // RGBA32F
    // RNG: xorwow state planes v0..v4 for the window's pixels (d is uniform per seed)
    uint32_t *rng_state; // [5][npix]
    // ray streams (stream.hpp): what the persistent traversal kernels leave for the per-pixel stages that follow
    GiHit<T> *gi_hit;    // [npix]: the GI ray's intersection, input of shade #2
    uint8_t *vis;        // [4 slots][npix]: 1 = the shadow ray of that light slot is blocked
    // Sample slots.  The samples of a frame are independent up to the spp mean, so a BATCH of them runs through each
    // stage in one launch (gridDim.z / the stream's item range = the samples of the batch): the planes one sample's stages
    // hand to each other -- gi_multiplier, intensity0/1, skips, gi_direction, gi_valid, the light commands, sc_position,
    // gi_hit, vis, the RNG states -- exist once per slot; slot z of plane X is X + z * smp_stride elements (the slot-major
    // light planes and vis: 4 * smp_stride), of the RNG planes rng_state + z * rng_stride.  What write_clean_color
    // leaves per sample goes to smp_* and k_spp_mean sums the slots in sample order (the float sums of the sample loop,
    // in the same order).
    uint32_t smp_stride, rng_stride;
    V4<T> *smp_color, *smp_gi_colored, *smp_gi_white; // DataT: the sample loop adds DataT values in f32, (float)(DataT) is exact
};
#define LPRT_MAX_SAMPLE_BATCH 8u
// the view of one sample slot
template <typename T> __host__ __device__ inline FrameDev<T> at_sample(FrameDev<T> f, uint32_t z) {
    const size_t o = (size_t)z * f.smp_stride;
    f.gi_multiplier += o;
    f.intensity0 += o;
    f.intensity1 += o;
    f.skips += o;
    f.gi_direction += o;
    f.gi_valid += o;
    f.lc_dir_maxt += o * LPRT_MAX_DIRECT_LIGHT;
    f.lc_multiplier += o * LPRT_MAX_DIRECT_LIGHT;
    f.lc_mask += o;
    f.sc_position += o;
    f.vis += o * LPRT_MAX_DIRECT_LIGHT;
    f.gi_hit += o;
    f.smp_color += o;
    f.smp_gi_colored += o;
    f.smp_gi_white += o;
    f.rng_state += (size_t)z * f.rng_stride;
    return f;
}

} // namespace lprt
