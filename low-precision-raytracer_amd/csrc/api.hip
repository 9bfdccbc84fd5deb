// C-ABI entry points (include/lprt.h) over the templated renderer implementation.
#include "build.hpp"
#include "kernels.hpp"
#include "stream.hpp"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <vector>

// A ray-casting launcher in the variant the context asks for (kernels.hpp): the reference-arithmetic half slab tests exist
// for DataT = half only (in single precision the default arithmetic IS the reference's).  Used inside Ctx<T> members.
#define LPRT_TRACE(fn)                                                                                                                \
    (this->reference_slab() ? [] {                                                                                                    \
        if constexpr (sizeof(T) == 2)                                                                                                 \
            return &::lprt::refslab::fn<T>;                                                                                           \
        else                                                                                                                          \
            return &::lprt::fastslab::fn<T>;                                                                                          \
    }()                                                                                                                               \
                            : &::lprt::fastslab::fn<T>)

namespace lprt {

// RDTexture owners (memory.hpp:334-369): RGBA8 texel arrays + the descriptor table the kernels index by handle
struct TextureStore {
    std::vector<std::unique_ptr<DevBuf>> texels;
    std::vector<Texture> host;
    DevBuf table, lut;
    uint32_t add(hipStream_t st, const uint8_t *rgba8, uint32_t w, uint32_t h, bool srgb) {
        if (!lut.ptr) {
            float t[256];
            for (int i = 0; i < 256; i++) { // IEC 61966-2-1 decode, evaluated in double, rounded once
                double c = i / 255.0;
                t[i] = (float)(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
            }
            lut.reserve(sizeof(t));
            LPRT_HIP(hipMemcpyAsync(lut.ptr, t, sizeof(t), hipMemcpyHostToDevice, st));
            LPRT_HIP(hipStreamSynchronize(st));
        }
        std::unique_ptr<DevBuf> b(new DevBuf);
        const size_t bytes = (size_t)w * h * 4;
        b->reserve(bytes);
        LPRT_HIP(hipMemcpyAsync(b->ptr, rgba8, bytes, hipMemcpyHostToDevice, st));
        host.push_back(Texture{b->as<uint32_t>(), w, h, srgb ? 1u : 0u, 0u});
        texels.push_back(std::move(b));
        DevBuf fresh; // the table grows by reallocation; frames in flight still read the old one until the sync below
        fresh.reserve(sizeof(Texture) * host.size());
        LPRT_HIP(hipMemcpyAsync(fresh.ptr, host.data(), sizeof(Texture) * host.size(), hipMemcpyHostToDevice, st));
        LPRT_HIP(hipStreamSynchronize(st));
        std::swap(table.ptr, fresh.ptr);
        std::swap(table.bytes, fresh.bytes);
        return (uint32_t)host.size() - 1u;
    }
};

struct CtxBase {
    std::string error;
    TextureStore tex;
    int device = 0;
    Window win{};
    lprt_precision precision = LPRT_F32;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    // a second stream of the context's own for work that runs BESIDE the frame's stream between two events (sample_phase: the
    // RNG expansion under the BLAS build); created on first use
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    void ensure_side_stream() {
        if (side_stream)
            return;
        LPRT_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
        LPRT_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
        LPRT_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
    }
    virtual ~CtxBase() {}
    virtual void upload_geometry(const lprt_vertex *, uint32_t, const uint32_t *, uint32_t, const lprt_mesh_vao *, uint32_t) = 0;
    virtual void upload_materials(const lprt_material *, uint32_t) = 0;
    virtual void build_blas() = 0;
    virtual void set_scene(const lprt_object *, const lprt_aabb *, uint32_t, const lprt_light *, uint32_t, const lprt_camera *,
                           const float *, const lprt_skybox *) = 0;
    virtual void trace_rays(const lprt_ray *, uint32_t, int, lprt_hit *) = 0;
    virtual void get_counters(int stage, lprt_counters *) = 0;
    virtual void rng_states(uint64_t, uint32_t, uint32_t, uint32_t *, size_t) = 0;
    void *ext_out = nullptr;
    uint32_t ext_first_row = 0, ext_rows = 0;
    virtual void reset_counters() = 0;
    virtual void blas_info(uint32_t, uint32_t *, uint32_t *, uint32_t *) = 0;
    virtual void read_blas_order(uint32_t, uint32_t *, uint32_t) = 0;
    virtual void read_blas_nodes(uint32_t, float *, float *, uint32_t *, uint32_t) = 0;
    virtual void render_phase(const lprt_frame_settings &, int phase) = 0;
    void render_frame(const lprt_frame_settings &fs) {
        render_phase(fs, LPRT_PHASE_SAMPLE);
        render_phase(fs, LPRT_PHASE_RESOLVE);
    }
    // rows of the window the per-sample stages (RNG, shade, trace) run on; the rest of the window gets its sample means
    // from the contexts that own those rows (LPRT_HIST_SAMPLES) between the two phases
    uint32_t sample_first = 0, sample_rows = 0; // 0 rows: the whole window
    int next_phase = LPRT_PHASE_SAMPLE;
    // behaviour knobs (lprt_option): initialised from the environment once, in lprt_create; never read from it afterwards
    int opts[LPRT_OPT_COUNT] = {-1, -1, 1, 0, -1, 1, 0, -1, 0, 12, 20, 0, (int)LPRT_MAX_SAMPLE_BATCH, 0, 1, 0, -1};
    bool reference_slab() const { return opts[LPRT_OPT_REFERENCE_SLAB] != 0; }
    bool rng_all_runs = false, rng_beside_build_on = true;
    void options_from_environment() {
        if (const char *w = std::getenv("LPRT_WIDE_WALK"))
            opts[LPRT_OPT_WIDE_WALK] = w[0] == '1' ? 1 : 0;
        if (const char *m = std::getenv("LPRT_TLAS_MODE"))
            opts[LPRT_OPT_TLAS_MODE] = m[0] == '1' ? 1 : 0;
        if (std::getenv("LPRT_NO_AXIS_REJECT"))
            opts[LPRT_OPT_AXIS_REJECT] = 0;
        if (std::getenv("LPRT_SVGF_ALWAYS_TEST"))
            opts[LPRT_OPT_SVGF_ALWAYS_TEST] = 1;
        if (const char *b = std::getenv("LPRT_BLAS_BUILDER"))
            opts[LPRT_OPT_BLAS_BUILDER] = std::string(b) == "morton" ? 1 : std::string(b) == "sah" ? 2 : std::string(b) == "median" ? 0 : -1;
        if (const char *l = std::getenv("LPRT_BLAS_LOCAL"))
            opts[LPRT_OPT_BLAS_LOCAL] = std::string(l) == "0" ? 0 : 1;
        if (std::getenv("LPRT_DEBUG_SYNC"))
            opts[LPRT_OPT_DEBUG_SYNC] = 1;
        rng_all_runs = std::getenv("LPRT_RNG_ALL_RUNS") != nullptr;
        if (const char *b = std::getenv("LPRT_RNG_BESIDE_BUILD"))
            rng_beside_build_on = b[0] != '0'; // A/B: states for the runs of background pixels too (RngStore::generate)
        if (const char *r = std::getenv("LPRT_RAY_STREAM"))
            opts[LPRT_OPT_RAY_STREAM] = r[0] == '1' ? 1 : 0;
        if (const char *r = std::getenv("LPRT_SAMPLE_BATCH"))
            opts[LPRT_OPT_SAMPLE_BATCH] = std::max(1, std::min((int)LPRT_MAX_SAMPLE_BATCH, std::atoi(r)));
        if (const char *r = std::getenv("LPRT_FLOOR_REPROJECTION"))
            opts[LPRT_OPT_FLOOR_REPROJECTION] = r[0] == '1' ? 1 : 0;
        if (const char *r = std::getenv("LPRT_STREAM_XCD"))
            opts[LPRT_OPT_STREAM_XCD] = r[0] == '0' ? 0 : 1;
        if (const char *r = std::getenv("LPRT_STREAM_SCHED"))
            opts[LPRT_OPT_STREAM_SCHED] = r[0] == '1' ? 1 : 0;
        if (const char *r = std::getenv("LPRT_STREAM_NODE_MIN"))
            opts[LPRT_OPT_STREAM_NODE_MIN] = std::max(1, std::min(64, std::atoi(r)));
        if (const char *r = std::getenv("LPRT_STREAM_REFILL_MIN"))
            opts[LPRT_OPT_STREAM_REFILL_MIN] = std::max(1, std::min(64, std::atoi(r)));
        if (const char *r = std::getenv("LPRT_RAY_SORT"))
            opts[LPRT_OPT_RAY_SORT] = std::max(0, std::min(3, std::atoi(r)));
        if (const char *r = std::getenv("LPRT_REFERENCE_SLAB"))
            opts[LPRT_OPT_REFERENCE_SLAB] = r[0] == '1' ? 1 : 0;
        if (const char *r = std::getenv("LPRT_STREAM_WIDE_CLOSEST"))
            opts[LPRT_OPT_STREAM_WIDE_CLOSEST] = r[0] == '1' ? 1 : 0;
    }
    // rows [first_row, first_row + n_rows) inside the window?  (the sum in 64 bits: first_row + n_rows may wrap in 32)
    bool rows_inside_window(uint32_t first_row, uint32_t n_rows) const {
        return first_row >= win.row_begin && (uint64_t)first_row + n_rows <= (uint64_t)win.row_end;
    }
    void set_sample_rows(uint32_t first_row, uint32_t n_rows) {
        if (n_rows == 0) {
            sample_first = sample_rows = 0;
            return;
        }
        if (!rows_inside_window(first_row, n_rows))
            throw std::string("sample rows must lie inside this context's row window");
        sample_first = first_row;
        sample_rows = n_rows;
    }
    virtual void reset_history() = 0;
    virtual void read_buffer(lprt_buffer_id, float *, size_t) = 0;
    virtual void output_ptr(void **, size_t *) = 0;
    struct HistPlane {
        void *ptr;
        size_t bytes_per_pixel;
        uint32_t group; // LPRT_HIST_FILTER | LPRT_HIST_TAA | LPRT_HIST_SAMPLES
    };
    virtual std::vector<HistPlane> history_planes() = 0;
    size_t history_row_bytes(uint32_t groups) {
        size_t b = 0;
        for (auto &pl : history_planes())
            if (groups & pl.group)
                b += pl.bytes_per_pixel;
        return b * win.W;
    }
    // rows [first_row, first_row + n_rows) of the history planes of the given groups <-> one packed device buffer, plane
    // after plane: ONE kernel launch (a plane's rows are contiguous, so each plane is a flat byte range)
    void copy_history_rows(uint32_t groups, uint32_t first_row, uint32_t n_rows, void *packed, bool to_packed) {
        if (!rows_inside_window(first_row, n_rows))
            throw std::string("history rows must lie inside this context's row window");
        CopyJobs jobs{};
        size_t off = 0, longest = 0;
        for (auto &pl : history_planes()) {
            if (!(groups & pl.group))
                continue;
            const size_t row = pl.bytes_per_pixel * win.W, bytes = row * n_rows;
            char *plane = static_cast<char *>(pl.ptr) + row * (first_row - win.row_begin);
            char *pk = static_cast<char *>(packed) + off;
            if (jobs.n >= LPRT_MAX_COPY_JOBS)
                throw std::string("internal: too many history planes");
            jobs.src[jobs.n] = to_packed ? plane : pk;
            jobs.dst[jobs.n] = to_packed ? pk : plane;
            jobs.bytes[jobs.n] = bytes;
            jobs.n++;
            longest = std::max(longest, bytes);
            off += bytes;
        }
        if (jobs.n && longest)
            launch_copy_jobs(stream, jobs, longest);
    }
    bool counters_enabled = false;
    StageTimer timer;
    uint64_t frame_index = 0;

    // per-stage HIP events on the context's stream (only when enabled: they serialise nothing but cost a few us)
    hipEvent_t get_event() {
        hipEvent_t e;
        if (!timer.pool.empty()) {
            e = timer.pool.back();
            timer.pool.pop_back();
        } else {
            LPRT_HIP(hipEventCreate(&e));
        }
        return e;
    }
    template <typename F> void stage(int id, F &&f) {
        const bool debug_sync = opts[LPRT_OPT_DEBUG_SYNC] != 0;
        if ((!timer.enabled || !(timer.mask & (1u << id))) && !debug_sync) {
            f();
            return;
        }
        if (debug_sync) {
            f();
            hipError_t e = hipStreamSynchronize(stream);
            if (e != hipSuccess)
                throw std::string("stage ") + std::to_string(id) + " failed: " + hipGetErrorString(e);
            std::fprintf(stderr, "[lprt] stage %d ok\n", id);
            return;
        }
        hipEvent_t a = get_event(), b = get_event();
        LPRT_HIP(hipEventRecord(a, stream));
        f();
        LPRT_HIP(hipEventRecord(b, stream));
        timer.pending.push_back({id, a, b});
        timer.launches[id]++;
    }
    void collect_times() {
        for (auto &p : timer.pending) {
            LPRT_HIP(hipEventSynchronize(p.b));
            float ms = 0;
            LPRT_HIP(hipEventElapsedTime(&ms, p.a, p.b));
            timer.ms[p.stage] += ms;
            timer.pool.push_back(p.a);
            timer.pool.push_back(p.b);
        }
        timer.pending.clear();
    }
};

template <typename T> struct Impl : CtxBase {
    GeometryStore<T> geo;
    TlasStore<T> tlas;
    DevBuf d_materials32, d_materials, d_scene32, d_objects, d_lights, d_counters, d_rays, d_hits, d_objrecs;
    bool objrecs_dirty = true;
    uint32_t n_materials = 0, n_objects = 0, n_lights = 0;
    lprt_camera camera{};
    lprt_skybox skybox{};
    PinnedStage h_scene[2];
    int h_scene_next = 0;
    float w2c[16] = {};
    bool have_scene = false;

    // RTRTProcedureBuffer + GBuffer + SVGFBuffer x2 + output surface, one arena
    DevBuf arena, d_readback, d_last_L2W, d_rng_planes;
    FrameDev<T> fb{};
    V4<T> *taa_a = nullptr, *taa_b = nullptr;
    RngStore rng;
    size_t arena_bytes = 0;
    bool svgf_ran = false;

    DevBuf d_debug;
    Impl() {
        d_counters.reserve(sizeof(Counters) * LPRT_STAGE_COUNT);
        d_debug.reserve(64);
        LPRT_HIP(hipMemset(d_debug.ptr, 0, 64));
    }
    // LPRT_BOUNDS builds: report the first failed index check
    void check_debug() {
        uint32_t dbg[4] = {0, 0, 0, 0};
        LPRT_HIP(hipMemcpy(dbg, d_debug.ptr, sizeof(dbg), hipMemcpyDeviceToHost));
        if (dbg[0]) {
            LPRT_HIP(hipMemset(d_debug.ptr, 0, 64));
            throw std::string("device bounds check ") + std::to_string(dbg[0]) + " failed: " + std::to_string(dbg[1]) + " vs " + std::to_string(dbg[2]);
        }
    }
    template <typename U> U *carve(size_t &off, size_t count) {
        off = (off + 255) & ~size_t(255);
        U *p = reinterpret_cast<U *>(static_cast<char *>(arena.ptr) + off);
        off += sizeof(U) * count;
        return p;
    }
    void layout(bool dry, size_t &off) {
        const size_t n = win.npix();
        off = 0;
        (void)dry;
        fb.win = win;
        fb.g_position = carve<V4<T>>(off, n);
        fb.g_normal_depth = carve<V4<T>>(off, n);
        fb.g_tangent = carve<V4<T>>(off, n);
        fb.g_uv = carve<V4<T>>(off, n);
        fb.g_color = carve<uint32_t>(off, n);
        fb.g_objid = carve<uint32_t>(off, n);
        fb.g_offset = carve<uint32_t>(off, n);
        fb.albedo = carve<V4<T>>(off, n);
        fb.color_inprogress = carve<V4<T>>(off, n);
        taa_a = carve<V4<T>>(off, n);
        taa_b = carve<V4<T>>(off, n);
        fb.taa_history = taa_a;
        fb.acc_color = carve<float4>(off, n);
        fb.acc_gi_colored = carve<float4>(off, n);
        fb.acc_gi_white = carve<float4>(off, n);
        fb.raw_color = carve<V4<T>>(off, n);
        fb.raw_gi_colored = carve<V4<T>>(off, n);
        fb.raw_gi_white = carve<V4<T>>(off, n);
        for (int l = 0; l < 2; l++) {
            fb.lobe[l].cv_a = carve<V4<T>>(off, n);
            fb.lobe[l].cv_b = carve<V4<T>>(off, n);
            fb.lobe[l].cv_c = carve<V4<T>>(off, n);
            fb.lobe[l].illum = carve<T>(off, n);
            fb.lobe[l].var_a = carve<T>(off, n);
            fb.lobe[l].var_b = carve<T>(off, n);
            fb.lobe[l].var_c = carve<T>(off, n);
            fb.lobe[l].miu1_hist = carve<T>(off, n);
            fb.lobe[l].miu2_hist = carve<T>(off, n);
            fb.lobe[l].miu1_cur = carve<T>(off, n);
            fb.lobe[l].miu2_cur = carve<T>(off, n);
        }
        fb.depth_gradient = carve<V2<T>>(off, n);
        fb.svgf_map = carve<TemporalMapT<T>>(off, n);
        fb.taa_map = carve<TemporalMapT<T>>(off, n);
        fb.valid_frame_count = carve<uint8_t>(off, n);
        fb.prev_frame_count = carve<uint8_t>(off, n);
        fb.last_frame_id = carve<uint2>(off, n);
        fb.output = carve<float4>(off, n);
        off = (off + 255) & ~size_t(255);
    }
    // The per-sample planes (FrameDev: sample slots), `slots` of each, in an allocation of their own that grows with the
    // largest batch a frame has asked for.  Zero-initialised like the arena (thrust::device_vector, cuda.hpp:1599-1608).
    DevBuf slot_arena;
    uint32_t slot_capacity = 0, last_slot = 0, last_i0_slot = 0;
    template <typename U> U *carve_slots(size_t &off, size_t count) {
        off = (off + 255) & ~size_t(255);
        U *p = reinterpret_cast<U *>(static_cast<char *>(slot_arena.ptr) + off);
        off += sizeof(U) * count;
        return p;
    }
    void layout_slots(uint32_t slots, size_t &off) {
        const size_t n = (size_t)win.npix() * slots;
        off = 0;
        fb.smp_stride = win.npix();
        fb.gi_multiplier = carve_slots<V4<T>>(off, n);
        fb.intensity0 = carve_slots<V4<T>>(off, n);
        fb.intensity1 = carve_slots<V4<T>>(off, n);
        fb.skips = carve_slots<uint2>(off, n);
        fb.gi_direction = carve_slots<V4<T>>(off, n);
        fb.gi_valid = carve_slots<uint8_t>(off, n);
        fb.lc_dir_maxt = carve_slots<V4<T>>(off, n * LPRT_MAX_DIRECT_LIGHT);
        fb.lc_multiplier = carve_slots<V4<T>>(off, n * LPRT_MAX_DIRECT_LIGHT);
        fb.lc_mask = carve_slots<uint8_t>(off, n);
        fb.sc_position = carve_slots<V4<T>>(off, n);
        fb.gi_hit = carve_slots<GiHit<T>>(off, n);
        fb.vis = carve_slots<uint8_t>(off, n * LPRT_MAX_DIRECT_LIGHT);
        fb.smp_color = carve_slots<V4<T>>(off, n);
        fb.smp_gi_colored = carve_slots<V4<T>>(off, n);
        fb.smp_gi_white = carve_slots<V4<T>>(off, n);
        off = (off + 255) & ~size_t(255);
    }
    void ensure_sample_slots(uint32_t slots) {
        if (slots <= slot_capacity)
            return;
        size_t total = 0;
        layout_slots(slots, total); // sizes only (pointers relative to the old / a null allocation)
        slot_arena.reserve(total);  // contents are scratch of one frame's sample stages: nothing to preserve
        layout_slots(slots, total);
        LPRT_HIP(hipMemsetAsync(slot_arena.ptr, 0, total, stream));
        slot_capacity = slots;
        last_slot = last_i0_slot = 0;
    }
    void ensure_frame_buffers() {
        ensure_sample_slots(1);
        if (arena.ptr)
            return;
        size_t total = 0;
        layout(true, total); // pointers relative to a null arena: sizes only
        arena.reserve(total);
        arena_bytes = total;
        layout(false, total);
        LPRT_HIP(hipMemsetAsync(arena.ptr, 0, arena_bytes, stream)); // thrust::device_vector zero-initialises (cuda.hpp:1599-1608)
        rng.init(stream);
    }
    void ensure_last_L2W() {
        size_t need = sizeof(M4<T>) * (size_t)std::max(geo.n_vaos, 1u);
        if (d_last_L2W.bytes < need) {
            d_last_L2W.reserve(need);
            LPRT_HIP(hipMemsetAsync(d_last_L2W.ptr, 0, need, stream));
        }
        fb.last_frame_L2W = d_last_L2W.as<M4<T>>();
    }
    void reset_history() override {
        frame_index = 0;
        next_phase = LPRT_PHASE_SAMPLE;
        if (arena.ptr)
            LPRT_HIP(hipMemsetAsync(arena.ptr, 0, arena_bytes, stream));
    }
    // the per-pixel planes of fb restricted to image rows [first, first + n): same storage, shifted pointers.  The
    // slot-major light-command planes and the RNG planes are scratch of the per-sample stages, indexed with the view's own
    // pixel count, and stay where they are.
    FrameDev<T> rows_view(uint32_t first, uint32_t n) const {
        FrameDev<T> v = fb;
        const size_t skip = (size_t)(first - win.row_begin) * win.W;
        auto sh = [skip](auto *&ptr) { ptr += skip; };
        sh(v.g_position), sh(v.g_normal_depth), sh(v.g_tangent), sh(v.g_uv), sh(v.g_color), sh(v.g_objid), sh(v.g_offset);
        sh(v.gi_multiplier), sh(v.intensity0), sh(v.intensity1), sh(v.skips), sh(v.gi_direction), sh(v.gi_valid), sh(v.lc_mask);
        sh(v.sc_position), sh(v.albedo), sh(v.color_inprogress), sh(v.taa_history);
        sh(v.acc_color), sh(v.acc_gi_colored), sh(v.acc_gi_white), sh(v.raw_color), sh(v.raw_gi_colored), sh(v.raw_gi_white);
        for (int l = 0; l < 2; l++) {
            SvgfLobe<T> &lb = v.lobe[l];
            sh(lb.cv_a), sh(lb.cv_b), sh(lb.cv_c), sh(lb.illum), sh(lb.var_a), sh(lb.var_b), sh(lb.var_c);
            sh(lb.miu1_hist), sh(lb.miu2_hist), sh(lb.miu1_cur), sh(lb.miu2_cur);
        }
        sh(v.depth_gradient), sh(v.svgf_map), sh(v.taa_map), sh(v.valid_frame_count), sh(v.prev_frame_count), sh(v.last_frame_id), sh(v.output);
        v.win.row_begin = first;
        v.win.row_end = first + n;
        return v;
    }
    SceneDev<T> frame_scene() {
        SceneDev<T> sc = scene_dev();
        // camera + reprojection matrices in DataT (RDCamera<DataT>, transform_W2C)
        for (int k = 0; k < 16; k++)
            sc.cam_L2W.m[k] = (T)camera.transform_L2W[k];
        sc.cam_fovy = (T)camera.field_of_view_y;
        sc.sky_valid = skybox.valid ? 1u : 0u;
        sc.sky_texture = skybox.texture;
        sc.sky_dx = (T)skybox.delta_x;
        sc.sky_dy = (T)skybox.delta_y;
        sc.sky_exposure = (T)skybox.exposure;
        return sc;
    }
    // shadow rays walk deep BLAS trees two levels per step (trace.hpp); LPRT_OPT_WIDE_WALK overrides for A/B runs
    // GI / shadow rays as ray streams (stream.hpp): large scenes by default, LPRT_OPT_RAY_STREAM overrides
    // which BLAS builder a build uses (LPRT_OPT_BLAS_BUILDER): -1 = the surface-area builder when some primitive is large enough for
    // its hierarchy to matter (>= 64 triangles), the median split otherwise (the Cornell box: 2-12 triangles per primitive)
    int builder_kind() const {
        if (opts[LPRT_OPT_BLAS_BUILDER] >= 0)
            return opts[LPRT_OPT_BLAS_BUILDER];
        for (const BlasInfo &bi : geo.host_blas)
            if (bi.n_tris >= 64u)
                return 2;
        return 0;
    }
    bool ray_streams() const { return opts[LPRT_OPT_RAY_STREAM] < 0 ? geo.max_depth >= 6 : opts[LPRT_OPT_RAY_STREAM] != 0; }
    DevBuf d_stream_counters; // LPRT_STREAM_COUNTER_WORDS zeroed words per stream launch of a frame: 8 work counters, one per 64-byte line
    uint32_t stream_launches = 0;
    uint32_t *next_stream_counter() { return d_stream_counters.as<uint32_t>() + LPRT_STREAM_COUNTER_WORDS * (size_t)stream_launches++; }
    bool wide_anyhit() const { return opts[LPRT_OPT_WIDE_WALK] < 0 ? geo.max_depth >= 6 : opts[LPRT_OPT_WIDE_WALK] != 0; }
    // ray sort (pipeline.hip): the items of a ray stream in (origin cell, direction octant) order.  Bit 0 of the option: GI rays,
    // bit 1: the shadow rays of round 1 (and of round 0 when they run as a stream); -1: both
    int ray_sort() const { return opts[LPRT_OPT_RAY_SORT] < 0 ? 0 : opts[LPRT_OPT_RAY_SORT]; } // measured: costs more than it buys (pipeline.hip)
    DevBuf d_sort_keys, d_sort_order, d_sort_bins;
    // sorts the items of the stream about to be launched (which: 0 GI, 1 / 2 shadow rays of round 0 / 1); null order: not sorted
    RaySortBufs sort_stream(const SceneDev<T> &sc, const FrameDev<T> &sv, uint32_t slots, int which) {
        RaySortBufs b{nullptr, nullptr, nullptr, nullptr};
        if (!((ray_sort() >> (which == 0 ? 0 : 1)) & 1))
            return b;
        const uint32_t n_items = ray_sort_items<T>(sv, sc.n_lights, slots, which);
        if (!n_items)
            return b;
        // sized for the largest stream of the batch at once (the shadow rays: items x light slots): growing later would free a
        // buffer the previous stream may still be reading
        const size_t n_max = std::max<size_t>(n_items, ray_sort_items<T>(sv, sc.n_lights, slots, 2));
        d_sort_keys.reserve(sizeof(uint32_t) * n_max);
        d_sort_order.reserve(sizeof(uint32_t) * n_max);
        d_sort_bins.reserve(sizeof(uint32_t) * (LPRT_RAYSORT_BINS + 4u));
        b = RaySortBufs{d_sort_keys.as<uint32_t>(), d_sort_order.as<uint32_t>(), d_sort_bins.as<uint32_t>(), d_sort_bins.as<uint32_t>() + LPRT_RAYSORT_BINS};
        launch_ray_sort<T>(stream, sc, sv, slots, which, b);
        return b;
    }
    uint32_t jitter_d = 0; // d word of this frame's seed A, drawn in the sample phase, used by the temporal maps
    void render_phase(const lprt_frame_settings &fs, int phase) override {
        if (phase != next_phase)
            throw std::string("render phases out of order: LPRT_PHASE_SAMPLE, then LPRT_PHASE_RESOLVE");
        if (phase == LPRT_PHASE_SAMPLE)
            sample_phase(fs);
        else
            resolve_phase(fs);
        next_phase = phase == LPRT_PHASE_SAMPLE ? LPRT_PHASE_RESOLVE : LPRT_PHASE_SAMPLE;
    }
    // primary rays over the window; RNG, shade #1/#2, shadow and GI rays, spp mean over the sample rows
    void sample_phase(const lprt_frame_settings &fs) {
        require_ready();
        ensure_frame_buffers();
        ensure_last_L2W();
        const uint32_t spp = fs.spp ? fs.spp : 1;
        // samples per launch of the per-sample stages (LPRT_OPT_SAMPLE_BATCH; FrameDev: sample slots)
        const uint32_t batch_max = std::min<uint32_t>({spp, (uint32_t)opts[LPRT_OPT_SAMPLE_BATCH], LPRT_MAX_SAMPLE_BATCH});
        ensure_sample_slots(batch_max);
        const bool first = frame_index == 0;
        const bool whole = sample_rows == 0 || (sample_first == win.row_begin && sample_rows == win.rows());
        FrameDev<T> sv = whole ? fb : rows_view(sample_first, sample_rows);
        const uint32_t total_px = win.W * win.H, first_px = win.row_begin * win.W, npix = win.npix();
        const uint32_t s_first_px = sv.win.row_begin * win.W, s_npix = sv.win.npix();
        auto seed_of = [&](uint32_t s, int which) -> uint64_t {
            if (fs.seeds)
                return fs.seeds[3 * s + (uint32_t)which];
            return 3 * (frame_index * spp + s) + 1 + (uint64_t)which;
        };
        // every curand_init of this frame: seed A of sample 0 (TAA jitter, frames >= 1) over the window + seed B of each
        // sample (shade #1 with GI) over the sample rows -- one batch when the two row ranges coincide.  shade #2 runs with
        // NO_GI and draws nothing (cuda.hpp:1900-1999).
        std::vector<uint64_t> seeds_a, seeds_b;
        if (!first)
            seeds_a.push_back(seed_of(0, 0));
        if (fs.gi_on)
            for (uint32_t s = 0; s < spp; s++)
                seeds_b.push_back(seed_of(s, 1));
        std::vector<uint32_t> seed_d(seeds_a.size() + seeds_b.size() + 1, 0u);
        const size_t b_planes_at = 5 * (size_t)npix * seeds_a.size();
        const bool have_seeds = !seeds_a.empty() || !seeds_b.empty();
        if (have_seeds)
            d_rng_planes.reserve(sizeof(uint32_t) * (b_planes_at + 5 * (size_t)s_npix * seeds_b.size()));
        // one batch: seed A over the window, the B seeds over the sample rows (the same range when `whole`); g_color: the frame's
        // G-buffer when it is in place -- runs of 64 pixels without a covered one get no states -- else null
        auto expand_rng = [&](hipStream_t st, const uint32_t *g_color) {
            std::vector<uint64_t> all(seeds_a);
            all.insert(all.end(), seeds_b.begin(), seeds_b.end());
            rng.generate(st, all.data(), (uint32_t)all.size(), (uint32_t)seeds_a.size(), total_px, first_px, npix, s_first_px, s_npix,
                         d_rng_planes.as<uint32_t>(), seed_d.data(), g_color);
        };
        // A rebuilding frame starts with ~0.7 ms in which a few dozen workgroups work through the top levels of the surface-area
        // build while the rest of the chip idles; the RNG expansion (pure vector instructions, 0.5 ms for the atrium's 9 seeds)
        // needs nothing of the build and runs beside it on the context's second stream -- for every run of pixels: the G-buffer
        // that tells the background runs apart is traced after the build -- and is joined in front of the first shading pass.
        const bool rng_beside_build = have_seeds && fs.rebuild_blas && rng_beside_build_on && opts[LPRT_OPT_DEBUG_SYNC] == 0;
        if (rng_beside_build) {
            ensure_side_stream();
            LPRT_HIP(hipEventRecord(ev_fork, stream)); // after everything that still reads last frame's states
            LPRT_HIP(hipStreamWaitEvent(side_stream, ev_fork, 0));
            expand_rng(side_stream, nullptr);
            LPRT_HIP(hipEventRecord(ev_join, side_stream));
        }
        if (fs.rebuild_blas) {
            stage(LPRT_STAGE_BUILD_BLAS, [&] { geo.build_blas(stream, builder_kind(), opts[LPRT_OPT_BLAS_LOCAL] != 0); });
            objrecs_dirty = true;
            ensure_objrecs();
        }
        SceneDev<T> sc = frame_scene();
        stage(LPRT_STAGE_PRIMARY, [&] { LPRT_TRACE(launch_fill_gbuffer)(stream, sc, fb, stage_counters(LPRT_STAGE_PRIMARY)); });
        const bool wide_walk = wide_anyhit(), streams = ray_streams(), wide_closest = opts[LPRT_OPT_STREAM_WIDE_CLOSEST] != 0;
        // round-0 shadow rays start on the camera-visible surface and aim at the same few lights: coherent, and the fused
        // per-pixel kernel serves them better (atrium: 7.5 vs 12.2 ms per frame) unless streams are forced on
        const bool stream_round0 = streams && opts[LPRT_OPT_RAY_STREAM] == 1;
        if (streams) {
            const size_t need = sizeof(uint32_t) * LPRT_STREAM_COUNTER_WORDS * (3 * (size_t)((spp + batch_max - 1) / batch_max) + 1);
            d_stream_counters.reserve(need);
            LPRT_HIP(hipMemsetAsync(d_stream_counters.ptr, 0, need, stream));
            stream_launches = 0;
        }
        if (rng_beside_build)
            LPRT_HIP(hipStreamWaitEvent(stream, ev_join, 0));
        else if (have_seeds)
            stage(LPRT_STAGE_RNG, [&] { expand_rng(stream, rng_all_runs ? nullptr : fb.g_color); });
        jitter_d = first ? 0u : seed_d[0];
        // opt-in: round 0's light commands and their visibility do not depend on the sample -> cast them for sample 0 only
        const bool reuse = fs.reuse_direct_light && fs.gi_on;
        sv.rng_stride = 5 * s_npix;
        for (uint32_t s0 = 0; s0 < spp; s0 += batch_max) {
            SampleBatch sb{};
            sb.first = s0;
            sb.count = std::min(batch_max, spp - s0);
            sb.reuse_direct = reuse ? 1u : 0u;
            if (fs.gi_on) {
                sv.rng_state = d_rng_planes.as<uint32_t>() + b_planes_at + (size_t)s0 * 5 * s_npix;
                for (uint32_t z = 0; z < sb.count; z++)
                    sb.rng_d[z] = seed_d[seeds_a.size() + s0 + z];
            }
            const uint32_t n = sb.count;
            stage(LPRT_STAGE_SHADE1, [&] { launch_shade<T>(stream, sc, sv, !fs.gi_on, sb); });
            // round 0's shadow rays: every slot, or (reuse) slot 0 of the batch that holds sample 0
            const uint32_t n_round0 = !reuse ? n : (s0 == 0 ? 1u : 0u);
            RaySortBufs sorted[3] = {};
            if (n_round0 && stream_round0)
                stage(LPRT_STAGE_RAY_SORT, [&] { sorted[1] = sort_stream(sc, sv, n_round0, 1); });
            if (n_round0)
                stage(LPRT_STAGE_TRACE_DI1, [&] {
                    if (stream_round0) {
                        const RaySortBufs so = sorted[1];
                        LPRT_TRACE(launch_stream_di)(stream, sc, sv, true, n_round0, next_stream_counter(), stage_counters(LPRT_STAGE_TRACE_DI1), wide_walk,
                                                     so.order, so.n_sorted);
                        launch_accum_di<T>(stream, sv, true, fs.gi_on != 0, fs.add_direct_out != 0, n_round0, reuse);
                    } else
                        LPRT_TRACE(launch_trace_di)(stream, sc, sv, true, fs.gi_on != 0, fs.add_direct_out != 0, n_round0, reuse,
                                           stage_counters(LPRT_STAGE_TRACE_DI1), wide_walk);
                });
            if (fs.gi_on) {
                // trace_gi + shade #2 of the hit in one kernel; trace_di #2 + write_clean_color in the next
                // (ray streams: each of the two as a traversal kernel + a per-pixel kernel, timed as stages of their own)
                if (streams) {
                    stage(LPRT_STAGE_RAY_SORT, [&] { sorted[0] = sort_stream(sc, sv, n, 0); });
                    stage(LPRT_STAGE_TRACE_GI, [&] {
                        const RaySortBufs so = sorted[0];
                        LPRT_TRACE(launch_stream_gi)(stream, sc, sv, n, next_stream_counter(), stage_counters(LPRT_STAGE_TRACE_GI), wide_closest, so.order,
                                                     so.n_sorted);
                    });
                    stage(LPRT_STAGE_SHADE2, [&] { launch_shade2<T>(stream, sc, sv, n); });
                    stage(LPRT_STAGE_RAY_SORT, [&] { sorted[2] = sort_stream(sc, sv, n, 2); });
                    stage(LPRT_STAGE_TRACE_DI2, [&] {
                        const RaySortBufs so = sorted[2];
                        LPRT_TRACE(launch_stream_di)(stream, sc, sv, false, n, next_stream_counter(), stage_counters(LPRT_STAGE_TRACE_DI2), wide_walk,
                                                     so.order, so.n_sorted);
                    });
                    stage(LPRT_STAGE_COMPOSE, [&] { launch_accum_di<T>(stream, sv, false, true, fs.add_direct_out != 0, n, reuse); });
                } else {
                    stage(LPRT_STAGE_TRACE_GI, [&] { LPRT_TRACE(launch_trace_gi)(stream, sc, sv, n, stage_counters(LPRT_STAGE_TRACE_GI)); });
                    stage(LPRT_STAGE_TRACE_DI2, [&] {
                        LPRT_TRACE(launch_trace_di)(stream, sc, sv, false, true, fs.add_direct_out != 0, n, reuse, stage_counters(LPRT_STAGE_TRACE_DI2), wide_walk);
                    });
                }
            }
            stage(LPRT_STAGE_COMPOSE, [&] { launch_spp_mean<T>(stream, sv, sb.first, sb.count, spp); });
            last_slot = sb.count - 1;
        }
        last_i0_slot = reuse ? 0u : last_slot;
        LPRT_HIP(hipGetLastError());
    }
    // temporal maps, SVGF, composition with the denoised lobes, TAA, output over the window
    void resolve_phase(const lprt_frame_settings &fs) {
        SceneDev<T> sc = frame_scene();
        const bool first = frame_index == 0;
        fb.rng_state = d_rng_planes.as<uint32_t>(); // seed A's planes (frames >= 1)
        stage(LPRT_STAGE_TEMPORAL_MAP, [&] { launch_temporal_map<T>(stream, sc, fb, first, jitter_d, opts[LPRT_OPT_FLOOR_REPROJECTION] != 0); });
        const V4<T> *res0 = fb.lobe[0].cv_a, *res1 = fb.lobe[1].cv_a;
        svgf_ran = fs.svgf != 0;
        if (fs.svgf) {
            // svgf_denoise (cuda.hpp:1636-1711), both lobes per launch
            stage(LPRT_STAGE_SVGF_GRADIENT, [&] { launch_depth_gradient<T>(stream, fb); });
            const bool always_test = opts[LPRT_OPT_SVGF_ALWAYS_TEST] != 0;
            stage(LPRT_STAGE_SVGF_TEMPORAL, [&] { launch_temporal_color<T>(stream, fb, fs.svgf_color_mix_weight, fs.svgf_moments_mix_weight, always_test); });
            stage(LPRT_STAGE_SVGF_MOMENTS, [&] { launch_moments<T>(stream, fb, fs.svgf_moments_mix_weight); });
            // the rows the frame's consumers read: the strip this context writes out (lprt_set_output_target), else its window
            uint32_t used_lo = 0, used_hi = win.rows();
            if (ext_out && ext_rows != 0) {
                used_lo = std::max(ext_first_row, win.row_begin) - win.row_begin;
                used_hi = std::min(ext_first_row + ext_rows, win.row_end) - win.row_begin;
                if (used_hi <= used_lo) {
                    used_lo = 0;
                    used_hi = win.rows();
                }
            }
            for (int level = 0; level < 5; level++)
                stage(LPRT_STAGE_SVGF_WAVELET, [&] { launch_wavelet<T>(stream, fb, level, always_test, used_lo, used_hi); });
            // moments history for the next frame: swap instead of the reference's (short) memcpy (DEVIATION Q8)
            for (int l = 0; l < 2; l++) {
                std::swap(fb.lobe[l].miu1_hist, fb.lobe[l].miu1_cur);
                std::swap(fb.lobe[l].miu2_hist, fb.lobe[l].miu2_cur);
            }
            res0 = fb.lobe[0].cv_c;
            res1 = fb.lobe[1].cv_c;
        }
        V4<T> *taa_next = (fb.taa_history == taa_a) ? taa_b : taa_a;
        stage(LPRT_STAGE_FINALIZE, [&] {
            launch_finalize<T>(stream, sc, fb, res0, res1, taa_next, fs.add_gi_colored != 0, fs.add_gi_white != 0, fs.demodulate != 0,
                               fs.taa_mix_weight, static_cast<float4 *>(ext_out), ext_first_row, ext_rows);
        });
        stage(LPRT_STAGE_BACKUP, [&] { launch_copy_last_transform<T>(stream, sc, fb.last_frame_L2W, n_objects); });
        fb.taa_history = taa_next;
        for (int k = 0; k < 16; k++)
            fb.last_frame_W2C.m[k] = (T)w2c[k]; // buffer->last_frame_transform_W2C = render_data->transform_W2C (imgui_window.cu:277)
        LPRT_HIP(hipGetLastError());
        frame_index++;
#if defined(LPRT_BOUNDS) || defined(LPRT_BOUNDS_A)
        LPRT_HIP(hipStreamSynchronize(stream));
        check_debug();
#endif
    }
    void read_buffer(lprt_buffer_id id, float *dst, size_t n_floats) override {
        ensure_frame_buffers();
        const uint32_t n_px = win.npix();
        uint32_t planes = 1;
        int ch = 3;
        const V4<T> *v4 = nullptr;
        const uint32_t *u32 = nullptr;
        const uint8_t *u8 = nullptr;
        bool rgba8 = false;
        switch (id) {
        case LPRT_BUF_COLOR_INPROGRESS: v4 = fb.color_inprogress; break;
        case LPRT_BUF_MUL_GI_COLORED: v4 = svgf_ran ? fb.lobe[0].cv_c : fb.lobe[0].cv_a; break;
        case LPRT_BUF_MUL_GI_WHITE: v4 = svgf_ran ? fb.lobe[1].cv_c : fb.lobe[1].cv_a; break;
        case LPRT_BUF_INTENSITY0: v4 = at_sample(fb, last_i0_slot).intensity0; break; // of the frame's last sample
        case LPRT_BUF_INTENSITY1: v4 = at_sample(fb, last_slot).intensity1; break;
        case LPRT_BUF_ALBEDO: v4 = fb.albedo; break;
        case LPRT_BUF_RAW_GI_COLORED: v4 = fb.raw_gi_colored; break;
        case LPRT_BUF_RAW_GI_WHITE: v4 = fb.raw_gi_white; break;
        case LPRT_BUF_RAW_COLOR: v4 = fb.raw_color; break;
        case LPRT_BUF_G_POSITION: v4 = fb.g_position; ch = 4; break;
        case LPRT_BUF_G_NORMAL_DEPTH: v4 = fb.g_normal_depth; ch = 4; break;
        case LPRT_BUF_G_TANGENT: v4 = fb.g_tangent; ch = 4; break;
        case LPRT_BUF_G_UV0UV1: v4 = fb.g_uv; ch = 4; break;
        case LPRT_BUF_G_COLOR: u32 = fb.g_color; ch = 4; rgba8 = true; break;
        case LPRT_BUF_G_OBJID: u32 = fb.g_objid; ch = 1; break;
        case LPRT_BUF_G_OFFSET: u32 = fb.g_offset; ch = 1; break;
        case LPRT_BUF_OUTPUT: ch = 4; break;
        case LPRT_BUF_GI_DIRECTION: v4 = at_sample(fb, last_slot).gi_direction; ch = 4; break;
        case LPRT_BUF_LIGHT_DIR_MAXT: v4 = at_sample(fb, last_slot).lc_dir_maxt; ch = 4; planes = LPRT_MAX_DIRECT_LIGHT; break;
        case LPRT_BUF_LIGHT_MASK: u8 = at_sample(fb, last_slot).lc_mask; ch = 1; break;
        case LPRT_BUF_SHADE_POSITION: v4 = at_sample(fb, last_slot).sc_position; break;
        case LPRT_BUF_SKIP: u32 = reinterpret_cast<const uint32_t *>(at_sample(fb, last_slot).skips); ch = 1; planes = 2; break; // uint2 = two words per pixel
        default: throw std::string("unknown buffer id");
        }
        if ((id == LPRT_BUF_LIGHT_DIR_MAXT || id == LPRT_BUF_LIGHT_MASK) && sample_rows != 0 && sample_rows != win.rows())
            throw std::string("light commands are indexed by sample-row pixel: readable in whole-window contexts only");
        const uint32_t n = n_px * planes;
        if (n_floats != (size_t)n * ch)
            throw std::string("read_buffer: size mismatch (expected rows*W*channels floats)");
        if (id == LPRT_BUF_OUTPUT) {
            LPRT_HIP(hipMemcpyAsync(dst, fb.output, sizeof(float) * n_floats, hipMemcpyDeviceToHost, stream));
        } else {
            d_readback.reserve(sizeof(float) * n_floats);
            if (v4)
                launch_read_v4<T>(stream, v4, d_readback.as<float>(), n, ch);
            else if (u8)
                launch_read_u8(stream, u8, d_readback.as<float>(), n);
            else
                launch_read_u32(stream, u32, d_readback.as<float>(), n, rgba8);
            LPRT_HIP(hipMemcpyAsync(dst, d_readback.ptr, sizeof(float) * n_floats, hipMemcpyDeviceToHost, stream));
        }
        LPRT_HIP(hipStreamSynchronize(stream));
    }
    // what frame n+1 reads of frame n beyond per-pixel G-buffer data: SVGF colour history (level-1 output) and moment
    // history of both lobes, the TAA history and the frame counters
    std::vector<HistPlane> history_planes() override {
        ensure_frame_buffers();
        std::vector<HistPlane> v;
        for (int l = 0; l < 2; l++)
            v.push_back({fb.lobe[l].cv_b, sizeof(V4<T>), LPRT_HIST_FILTER});
        for (int l = 0; l < 2; l++) {
            v.push_back({fb.lobe[l].miu1_hist, sizeof(T), LPRT_HIST_FILTER});
            v.push_back({fb.lobe[l].miu2_hist, sizeof(T), LPRT_HIST_FILTER});
        }
        v.push_back({fb.prev_frame_count, 1, LPRT_HIST_FILTER});
        v.push_back({fb.taa_history, sizeof(V4<T>), LPRT_HIST_TAA});
        for (int l = 0; l < 2; l++) // this frame's sample means (write_clean_color), the denoiser's input
            v.push_back({fb.lobe[l].cv_a, sizeof(V4<T>), LPRT_HIST_SAMPLES});
        return v;
    }
    void output_ptr(void **ptr, size_t *bytes) override {
        ensure_frame_buffers();
        *ptr = fb.output;
        *bytes = sizeof(float4) * (size_t)win.npix();
    }

    void upload_geometry(const lprt_vertex *vbo, uint32_t nv, const uint32_t *ebo, uint32_t ni, const lprt_mesh_vao *vaos,
                         uint32_t nvaos) override {
        geo.upload(stream, vbo, nv, ebo, ni, vaos, nvaos);
    }
    void upload_materials(const lprt_material *m, uint32_t n) override {
        for (uint32_t i = 0; i < n; i++)
            if (m[i].uv_color < 2u && m[i].tex_color >= tex.host.size())
                throw std::string("material ") + std::to_string(i) + " names texture handle " + std::to_string(m[i].tex_color) +
                    " which lprt_upload_texture has not returned (upload textures before materials)";
        n_materials = n;
        d_materials32.reserve(sizeof(lprt_material) * (size_t)std::max(n, 1u));
        d_materials.reserve(sizeof(MaterialT<T>) * (size_t)std::max(n, 1u));
        if (n) {
            LPRT_HIP(hipMemcpyAsync(d_materials32.ptr, m, sizeof(lprt_material) * (size_t)n, hipMemcpyHostToDevice, stream));
            LPRT_HIP(hipStreamSynchronize(stream));
            launch_convert_materials<T>(stream, d_materials32.as<lprt_material>(), d_materials.as<MaterialT<T>>(), n);
        }
    }
    void build_blas() override {
        stage(LPRT_STAGE_BUILD_BLAS, [&] { geo.build_blas(stream, builder_kind(), opts[LPRT_OPT_BLAS_LOCAL] != 0); });
        objrecs_dirty = true;
    }
    void ensure_objrecs() {
        if (!objrecs_dirty)
            return;
        d_objrecs.reserve(sizeof(ObjRec<T>) * (size_t)std::max(n_objects, 1u));
        launch_make_objrecs<T>(stream, tlas.d_slots.template as<ObjSlot<T>>(), d_objects.template as<ObjConst<T>>(), geo.d_blas.template as<BlasInfo>(),
                               geo.d_roots.template as<T>(), n_objects, geo.n_vaos, d_objrecs.as<ObjRec<T>>());
        objrecs_dirty = false;
    }
    void set_scene(const lprt_object *obj, const lprt_aabb *aabbs, uint32_t n, const lprt_light *lights, uint32_t nl,
                   const lprt_camera *cam, const float *W2C, const lprt_skybox *sky) override {
        n_objects = n;
        n_lights = nl;
        // The caller's arrays may go away after return, and a per-frame call must not stall the stream: objects, lights and
        // world boxes are copied into one of two pinned staging buffers (reused only after the copy issued from it has
        // completed), go to the device in ONE transfer, and one kernel builds the TLAS, converts lights and instance
        // constants to DataT and gathers the per-slot object records.
        const size_t b_obj = sizeof(lprt_object) * (size_t)n, b_light = sizeof(lprt_light) * (size_t)nl, b_box = sizeof(lprt_aabb) * (size_t)n;
        const size_t total = b_obj + b_light + b_box;
        d_scene32.reserve(std::max<size_t>(total, 16));
        d_objects.reserve(sizeof(ObjConst<T>) * (size_t)std::max(n, 1u));
        d_lights.reserve(sizeof(LightT<T>) * (size_t)std::max(nl, 1u));
        d_objrecs.reserve(sizeof(ObjRec<T>) * (size_t)std::max(n, 1u));
        PinnedStage &hs = h_scene[h_scene_next];
        h_scene_next ^= 1;
        hs.acquire(total + 16);
        char *hp = static_cast<char *>(hs.ptr);
        if (n) {
            std::memcpy(hp, obj, b_obj);
            std::memcpy(hp + b_obj + b_light, aabbs, b_box);
        }
        if (nl)
            std::memcpy(hp + b_obj, lights, b_light);
        if (total)
            LPRT_HIP(hipMemcpyAsync(d_scene32.ptr, hp, total, hipMemcpyHostToDevice, stream));
        hs.release(stream);
        char *dp = static_cast<char *>(d_scene32.ptr);
        SceneSetup<T> su{};
        su.objects32 = reinterpret_cast<const lprt_object *>(dp);
        su.objects = d_objects.as<ObjConst<T>>();
        su.n_objects = n;
        su.lights32 = reinterpret_cast<const lprt_light *>(dp + b_obj);
        su.lights = d_lights.as<LightT<T>>();
        su.n_lights = nl;
        su.blas = geo.built ? geo.d_blas.template as<BlasInfo>() : nullptr;
        su.roots = geo.d_roots.template as<T>();
        su.n_vaos = geo.n_vaos;
        su.objrecs = d_objrecs.as<ObjRec<T>>();
        stage(LPRT_STAGE_BUILD_TLAS, [&] { tlas.build(stream, reinterpret_cast<const lprt_aabb *>(dp + b_obj + b_light), n, su); });
        camera = *cam;
        std::memcpy(w2c, W2C, sizeof(w2c));
        if (sky && sky->valid) {
            if (sky->texture >= tex.host.size())
                throw std::string("skybox names a texture handle lprt_upload_texture has not returned");
            skybox = *sky;
        } else {
            skybox = lprt_skybox{};
        }
        have_scene = true;
        objrecs_dirty = !geo.built; // made by the set-up kernel unless the BLAS set does not exist yet
    }
    SceneDev<T> scene_dev() const {
        SceneDev<T> sc{};
        sc.tlas_inner = tlas.d_inner.template as<InnerNode<T>>();
        sc.tlas_leaf = tlas.d_leaf.template as<uint32_t>();
        sc.tlas_root = tlas.d_root.template as<T>();
        sc.tlas_slots = tlas.d_slots.template as<ObjSlot<T>>();
        sc.objrecs = d_objrecs.template as<ObjRec<T>>();
        sc.tlas_depth = tlas.depth;
        sc.n_objects = tlas.n_objects;
        sc.tlas_mode = tlas.n_objects <= 32u ? 1u : 0u;
        if (opts[LPRT_OPT_TLAS_MODE] == 0 || (opts[LPRT_OPT_TLAS_MODE] == 1 && tlas.n_objects <= 32u)) // experiment knob: 0 tree, 1 flat scan
            sc.tlas_mode = (uint32_t)opts[LPRT_OPT_TLAS_MODE];
        sc.axis_reject = opts[LPRT_OPT_AXIS_REJECT] ? 1u : 0u; // A/B switch (results must not depend on it)
        sc.stream_refill_min = (uint32_t)opts[LPRT_OPT_STREAM_REFILL_MIN];
        sc.stream_node_min = (uint32_t)opts[LPRT_OPT_STREAM_NODE_MIN];
        sc.stream_sched = (uint32_t)opts[LPRT_OPT_STREAM_SCHED];
        sc.stream_xcd = opts[LPRT_OPT_STREAM_XCD] ? 1u : 0u;
        sc.objects = d_objects.template as<ObjConst<T>>();
        sc.blas = geo.d_blas.template as<BlasInfo>();
        sc.blas_roots = geo.d_roots.template as<T>();
        sc.blas_inner = geo.d_inner.template as<BlasNode<T>>();
        sc.blas_leaf = geo.d_leaf.template as<uint32_t>();
        sc.tris = geo.d_tris.template as<TriRec<T>>();
        sc.tris_f32 = geo.d_tris_f32.template as<TriRecF32>();
        sc.vbo = geo.d_vbo.template as<Vertex<T>>();
        sc.ebo = geo.d_ebo.template as<uint32_t>();
        sc.vaos = geo.d_vaos.template as<lprt_mesh_vao>();
        sc.materials = d_materials.template as<MaterialT<T>>();
        sc.lights = d_lights.template as<LightT<T>>();
        sc.textures = tex.table.template as<Texture>();
        sc.srgb_lut = tex.lut.template as<float>();
        sc.n_lights = n_lights;
        sc.n_vaos = geo.n_vaos;
        sc.lim_vertices = geo.n_vertices;
        sc.lim_indices = geo.n_indices;
        sc.lim_tris = geo.n_tris;
        sc.lim_inner = geo.n_inner;
        sc.lim_leaves = geo.n_leaves;
        sc.lim_objects = n_objects;
        sc.lim_materials = n_materials;
        sc.debug = d_debug.template as<uint32_t>();
        return sc;
    }
    void require_ready() {
        if (!geo.built)
            throw std::string("BLAS not built: call lprt_build_blas after lprt_upload_geometry");
        if (!have_scene)
            throw std::string("no scene: call lprt_set_scene first");
        ensure_objrecs();
    }
    void trace_rays(const lprt_ray *rays, uint32_t n, int find_any, lprt_hit *hits) override {
        require_ready();
        if (!n)
            return;
        d_rays.reserve(sizeof(lprt_ray) * (size_t)n);
        d_hits.reserve(sizeof(lprt_hit) * (size_t)n);
        LPRT_HIP(hipMemcpyAsync(d_rays.ptr, rays, sizeof(lprt_ray) * (size_t)n, hipMemcpyHostToDevice, stream));
        // the walk the frame kernels would take: any-hit batches follow wide_anyhit(); LPRT_OPT_WIDE_WALK = 1 also sends
        // closest-hit batches through the two-level walk (parity of that walk at hit-record level)
        const bool wide = find_any ? wide_anyhit() : (opts[LPRT_OPT_WIDE_WALK] == 1);
        if (ray_streams()) { // the persistent traversal the frame kernels take in large scenes
            d_stream_counters.reserve(sizeof(uint32_t) * LPRT_STREAM_COUNTER_WORDS);
            LPRT_HIP(hipMemsetAsync(d_stream_counters.ptr, 0, sizeof(uint32_t) * LPRT_STREAM_COUNTER_WORDS, stream));
            LPRT_TRACE(launch_stream_rays)(stream, scene_dev(), d_rays.as<lprt_ray>(), n, find_any != 0,
                                  find_any ? wide : (wide || opts[LPRT_OPT_STREAM_WIDE_CLOSEST] != 0), d_hits.as<lprt_hit>(),
                                  d_stream_counters.as<uint32_t>(), stage_counters(LPRT_STAGE_PRIMARY));
        } else {
            LPRT_TRACE(launch_trace_rays)(stream, scene_dev(), d_rays.as<lprt_ray>(), n, find_any != 0, wide, d_hits.as<lprt_hit>(),
                                 stage_counters(LPRT_STAGE_PRIMARY));
        }
        LPRT_HIP(hipGetLastError());
        LPRT_HIP(hipMemcpyAsync(hits, d_hits.ptr, sizeof(lprt_hit) * (size_t)n, hipMemcpyDeviceToHost, stream));
        LPRT_HIP(hipStreamSynchronize(stream));
    }
    // stage < 0: sum over all stages
    void get_counters(int stage, lprt_counters *out) override {
        Counters all[LPRT_STAGE_COUNT];
        LPRT_HIP(hipMemcpyAsync(all, d_counters.ptr, sizeof(all), hipMemcpyDeviceToHost, stream));
        LPRT_HIP(hipStreamSynchronize(stream));
        Counters c{};
        for (int i = 0; i < LPRT_STAGE_COUNT; i++) {
            if (stage >= 0 && i != stage)
                continue;
            c.rays += all[i].rays;
            c.tlas_nodes += all[i].tlas_nodes;
            c.aabbw += all[i].aabbw;
            c.objects += all[i].objects;
            c.blas_nodes += all[i].blas_nodes;
            c.tris += all[i].tris;
            c.fallbacks += all[i].fallbacks;
            for (int k = 0; k < 3; k++) {
                c.wave_steps[k] += all[i].wave_steps[k];
                c.lane_steps[k] += all[i].lane_steps[k];
            }
        }
        out->rays = c.rays;
        out->tlas_node_visits = c.tlas_nodes;
        out->aabbw_tests = c.aabbw;
        out->object_visits = c.objects;
        out->blas_node_visits = c.blas_nodes;
        out->tri_tests = c.tris;
        out->fallback_tests = c.fallbacks;
        for (int k = 0; k < 3; k++) {
            out->wave_steps[k] = c.wave_steps[k];
            out->lane_steps[k] = c.lane_steps[k];
        }
    }
    void reset_counters() override { LPRT_HIP(hipMemsetAsync(d_counters.ptr, 0, sizeof(Counters) * LPRT_STAGE_COUNT, stream)); }
    Counters *stage_counters(int stage) { return counters_enabled ? d_counters.as<Counters>() + stage : nullptr; }
    void rng_states(uint64_t seed, uint32_t first, uint32_t count, uint32_t *out, size_t out_words) override {
        if (out_words != 6 * (size_t)count)
            throw std::string("rng_states: the out array must hold 6 words per pixel");
        if (!rng.ready)
            rng.init(stream);
        if ((uint64_t)first + count > (uint64_t)win.W * win.H || count == 0)
            throw std::string("rng_states: pixel range outside the frame");
        DevBuf planes;
        planes.reserve(sizeof(uint32_t) * 5 * (size_t)count);
        uint32_t d = 0;
        rng.generate(stream, &seed, 1, win.W * win.H, first, count, planes.as<uint32_t>(), &d);
        std::vector<uint32_t> host(5 * (size_t)count);
        LPRT_HIP(hipMemcpyAsync(host.data(), planes.ptr, sizeof(uint32_t) * host.size(), hipMemcpyDeviceToHost, stream));
        LPRT_HIP(hipStreamSynchronize(stream));
        for (uint32_t i = 0; i < count; i++) {
            for (int k = 0; k < 5; k++)
                out[6 * (size_t)i + k] = host[(size_t)k * count + i];
            out[6 * (size_t)i + 5] = d;
        }
    }
    void blas_info(uint32_t vao, uint32_t *n_tris, uint32_t *n_nodes, uint32_t *depth) override {
        if (!n_tris || !n_nodes || !depth)
            throw std::string("null out pointer");
        if (vao >= geo.n_vaos)
            throw std::string("vao index out of range");
        const BlasInfo &bi = geo.host_blas[vao];
        *n_tris = bi.n_tris;
        *depth = bi.depth;
        *n_nodes = bi.n_tris ? (1u << bi.depth) - 1 : 0; // heap slots of the node records (depths 0 .. D-1)
    }
    void read_blas_order(uint32_t vao, uint32_t *order, uint32_t n) override {
        if (vao >= geo.n_vaos || !geo.built)
            throw std::string("bad vao index or BLAS not built");
        const BlasInfo &bi = geo.host_blas[vao];
        if (n != bi.n_tris)
            throw std::string("size mismatch");
        if (!n)
            return;
        if (!order)
            throw std::string("null out pointer");
        // sorted values are global triangle ids; report primitive-local triangle indices
        LPRT_HIP(hipMemcpyAsync(order, geo.leaf_order() + bi.tri_base, sizeof(uint32_t) * (size_t)n,
                                hipMemcpyDeviceToHost, stream));
        LPRT_HIP(hipStreamSynchronize(stream));
        for (uint32_t i = 0; i < n; i++)
            order[i] -= bi.tri_base;
    }
    // node records in heap order (slot i at [i - 1], depths 0 .. D-1): both child boxes and both leaf words; root6 = the root box
    void read_blas_nodes(uint32_t vao, float *root6, float *boxes, uint32_t *leaf_words, uint32_t n_slots) override {
        if (vao >= geo.n_vaos || !geo.built)
            throw std::string("bad vao index or BLAS not built");
        const BlasInfo &bi = geo.host_blas[vao];
        const uint32_t expect = bi.n_tris ? (1u << bi.depth) - 1 : 0;
        if (n_slots != expect)
            throw std::string("size mismatch");
        if (!root6 || (n_slots && (!boxes || !leaf_words)))
            throw std::string("null out pointer");
        std::vector<BlasNode<T>> inner(n_slots);
        T root[6] = {};
        if (n_slots)
            LPRT_HIP(hipMemcpyAsync(inner.data(), geo.d_inner.template as<BlasNode<T>>() + bi.inner_base, sizeof(BlasNode<T>) * n_slots,
                                    hipMemcpyDeviceToHost, stream));
        if (bi.n_tris)
            LPRT_HIP(hipMemcpyAsync(root, geo.d_roots.template as<T>() + 6 * bi.root_base, sizeof(root), hipMemcpyDeviceToHost, stream));
        LPRT_HIP(hipStreamSynchronize(stream));
        for (int k = 0; k < 6; k++)
            root6[k] = (float)root[k];
        for (uint32_t i = 0; i < n_slots; i++) {
            const BlasNode<T> &nd = inner[i];
            float *c0 = boxes + 12 * (size_t)i, *c1 = c0 + 6;
            for (int k = 0; k < 3; k++) {
                c0[k] = (float)nd.lo0[k];
                c0[3 + k] = (float)nd.hi0[k];
                c1[k] = (float)nd.lo1[k];
                c1[3 + k] = (float)nd.hi1[k];
            }
            leaf_words[2 * (size_t)i] = nd.leaf[0];
            leaf_words[2 * (size_t)i + 1] = nd.leaf[1];
        }
    }
};

} // namespace lprt

using namespace lprt;

struct lprt_ctx {
    std::unique_ptr<CtxBase> impl;
};

static thread_local std::string g_create_error;

// makes the context's device current for the duration of one call and puts the caller's device back afterwards (a torch
// process with several devices must not find its current device changed by a call into this library)
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int device) {
        if (hipGetDevice(&prev) != hipSuccess)
            prev = -1;
        if (prev != device) {
            (void)hipSetDevice(device);
            switched = true;
        }
    }
    ~DeviceScope() {
        if (switched && prev >= 0)
            (void)hipSetDevice(prev);
    }
};

// no exception crosses the C boundary: whatever is thrown becomes a status code + lprt_last_error text
template <typename F> static int guarded(lprt_ctx *ctx, F &&f) {
    if (!ctx || !ctx->impl)
        return LPRT_ERR_INVALID;
    CtxBase &c = *ctx->impl;
    try {
        DeviceScope dev(c.device);
        f(c);
        return LPRT_OK;
    } catch (const HipError &e) {
        char buf[512];
        std::snprintf(buf, sizeof(buf), "%s failed: %s (%s:%d)", e.what, hipGetErrorString(e.code), e.file, e.line);
        c.error = buf;
        return LPRT_ERR_DEVICE;
    } catch (const std::string &s) {
        c.error = s;
        return LPRT_ERR_INVALID;
    } catch (const std::bad_alloc &) {
        c.error = "host allocation failed";
        return LPRT_ERR_DEVICE;
    } catch (const std::exception &e) {
        c.error = std::string("internal error: ") + e.what();
        return LPRT_ERR_INVALID;
    } catch (...) {
        c.error = "internal error: unknown exception";
        return LPRT_ERR_INVALID;
    }
}

extern "C" {

const char *lprt_version(void) { return "lprt-hip 0.3 (gfx950)"; }

void lprt_get_abi_info(lprt_abi_info *out, uint32_t info_bytes) {
    if (!out || info_bytes < sizeof(uint32_t))
        return;
    lprt_abi_info me{};
    me.info_bytes = (uint32_t)sizeof(lprt_abi_info);
    me.stage_count = LPRT_STAGE_COUNT;
    me.buffer_count = LPRT_BUF_COUNT;
    me.option_count = LPRT_OPT_COUNT;
    me.sizeof_vertex = (uint32_t)sizeof(lprt_vertex);
    me.sizeof_mesh_vao = (uint32_t)sizeof(lprt_mesh_vao);
    me.sizeof_material = (uint32_t)sizeof(lprt_material);
    me.sizeof_object = (uint32_t)sizeof(lprt_object);
    me.sizeof_aabb = (uint32_t)sizeof(lprt_aabb);
    me.sizeof_light = (uint32_t)sizeof(lprt_light);
    me.sizeof_camera = (uint32_t)sizeof(lprt_camera);
    me.sizeof_skybox = (uint32_t)sizeof(lprt_skybox);
    me.sizeof_frame_settings = (uint32_t)sizeof(lprt_frame_settings);
    me.sizeof_ray = (uint32_t)sizeof(lprt_ray);
    me.sizeof_hit = (uint32_t)sizeof(lprt_hit);
    me.sizeof_counters = (uint32_t)sizeof(lprt_counters);
    me.rng_state_words = 6;
    std::memcpy(out, &me, std::min<size_t>(info_bytes, sizeof(me)));
}

int lprt_create(lprt_ctx **out, int device, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end,
                lprt_precision precision) {
    if (!out)
        return LPRT_ERR_INVALID;
    *out = nullptr;
    if (width == 0 || height == 0 || width > 65535 || height > 65535 || row_begin >= row_end || row_end > height ||
        (precision != LPRT_F32 && precision != LPRT_F16)) {
        g_create_error = "invalid size, row window or precision";
        return LPRT_ERR_INVALID;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || device < 0 || device >= n_dev) {
        g_create_error = "no usable HIP device (this library has no CPU fallback)";
        return LPRT_ERR_NO_DEVICE;
    }
    try {
        DeviceScope dev(device);
        std::unique_ptr<CtxBase> impl;
        if (precision == LPRT_F32)
            impl.reset(new Impl<float>());
        else
            impl.reset(new Impl<h16>());
        impl->device = device;
        impl->precision = precision;
        impl->win = Window{width, height, row_begin, row_end};
        impl->options_from_environment(); // the only place the environment is read
        LPRT_HIP(hipStreamCreateWithFlags(&impl->stream, hipStreamNonBlocking));
        impl->own_stream = true;
        auto *ctx = new lprt_ctx();
        ctx->impl = std::move(impl);
        ctx->impl->reset_counters();
        *out = ctx;
        return LPRT_OK;
    } catch (const HipError &e) {
        g_create_error = std::string(e.what) + " failed: " + hipGetErrorString(e.code);
        return LPRT_ERR_DEVICE;
    } catch (const std::string &s) {
        g_create_error = s;
        return LPRT_ERR_INVALID;
    } catch (const std::exception &e) {
        g_create_error = std::string("internal error: ") + e.what();
        return LPRT_ERR_DEVICE;
    } catch (...) {
        g_create_error = "internal error: unknown exception";
        return LPRT_ERR_DEVICE;
    }
}

void lprt_destroy(lprt_ctx *ctx) {
    if (!ctx)
        return;
    if (ctx->impl) {
        DeviceScope dev(ctx->impl->device);
        // Nothing of this context may be in flight while its memory goes away.  A stream the caller handed over
        // (lprt_set_stream) is NOT touched here -- the caller may have destroyed it already; the device-wide
        // synchronisation covers whatever was enqueued on it.
        if (ctx->impl->own_stream && ctx->impl->stream)
            (void)hipStreamSynchronize(ctx->impl->stream);
        (void)hipDeviceSynchronize();
        for (auto &p : ctx->impl->timer.pending) {
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
        for (hipEvent_t e : ctx->impl->timer.pool)
            (void)hipEventDestroy(e);
        ctx->impl->timer.pending.clear();
        ctx->impl->timer.pool.clear();
        hipStream_t s = ctx->impl->own_stream ? ctx->impl->stream : nullptr;
        if (ctx->impl->side_stream) {
            (void)hipStreamDestroy(ctx->impl->side_stream);
            (void)hipEventDestroy(ctx->impl->ev_fork);
            (void)hipEventDestroy(ctx->impl->ev_join);
        }
        ctx->impl.reset();
        if (s)
            (void)hipStreamDestroy(s);
    }
    delete ctx;
}

const char *lprt_last_error(const lprt_ctx *ctx) {
    if (!ctx || !ctx->impl)
        return g_create_error.c_str();
    return ctx->impl->error.c_str();
}

int lprt_set_stream(lprt_ctx *ctx, void *hip_stream) {
    return guarded(ctx, [&](CtxBase &c) {
        LPRT_HIP(hipStreamSynchronize(c.stream));
        if (c.own_stream)
            (void)hipStreamDestroy(c.stream);
        c.stream = static_cast<hipStream_t>(hip_stream);
        c.own_stream = false;
    });
}
int lprt_synchronize(lprt_ctx *ctx) {
    return guarded(ctx, [&](CtxBase &c) { LPRT_HIP(hipStreamSynchronize(c.stream)); });
}

int lprt_upload_geometry(lprt_ctx *ctx, const lprt_vertex *vbo, uint32_t n_vertices, const uint32_t *ebo, uint32_t n_indices,
                         const lprt_mesh_vao *vaos, uint32_t n_vaos) {
    return guarded(ctx, [&](CtxBase &c) {
        if ((n_vertices && !vbo) || (n_indices && !ebo) || (n_vaos && !vaos) || n_indices % 3)
            throw std::string("null pointer or index count not a multiple of 3");
        for (uint32_t i = 0; i < n_vaos; i++) {
            if (vaos[i].ebo_index || vaos[i].vbo_index)
                throw std::string("only one shared VBO/EBO is supported (ebo_index = vbo_index = 0)");
            if (vaos[i].ebo_length % 3 || vaos[i].ebo_offset % 3 || (uint64_t)vaos[i].ebo_offset + vaos[i].ebo_length > n_indices)
                throw std::string("primitive range outside the EBO or not whole triangles");
            if (i && vaos[i].ebo_offset < vaos[i - 1].ebo_offset + vaos[i - 1].ebo_length)
                throw std::string("primitives must be disjoint and ordered by ebo_offset");
            if (vaos[i].ebo_length / 3 > (1u << 30))
                throw std::string("primitive too large");
        }
        for (uint32_t i = 0; i < n_indices; i++)
            if (ebo[i] >= n_vertices)
                throw std::string("index out of range");
        c.upload_geometry(vbo, n_vertices, ebo, n_indices, vaos, n_vaos);
    });
}
int lprt_upload_materials(lprt_ctx *ctx, const lprt_material *materials, uint32_t n) {
    return guarded(ctx, [&](CtxBase &c) {
        if (n && !materials)
            throw std::string("null materials");
        c.upload_materials(materials, n);
    });
}
int lprt_upload_texture(lprt_ctx *ctx, const uint8_t *rgba8, uint32_t width, uint32_t height, uint32_t srgb, uint32_t *handle) {
    return guarded(ctx, [&](CtxBase &c) {
        if (!rgba8 || !handle)
            throw std::string("null texture pointer");
        if (!width || !height || width > 32768u || height > 32768u)
            throw std::string("texture extent must be 1..32768");
        *handle = c.tex.add(c.stream, rgba8, width, height, srgb != 0);
    });
}
int lprt_debug_sample_texture(lprt_ctx *ctx, uint32_t handle, const float *uv, uint32_t n, float *rgba) {
    return guarded(ctx, [&](CtxBase &c) {
        if (handle >= c.tex.host.size())
            throw std::string("unknown texture handle");
        if (n && (!uv || !rgba))
            throw std::string("null uv/rgba pointer");
        if (!n)
            return;
        DevBuf d_uv, d_out;
        d_uv.reserve(sizeof(float) * 2 * (size_t)n);
        d_out.reserve(sizeof(float) * 4 * (size_t)n);
        LPRT_HIP(hipMemcpyAsync(d_uv.ptr, uv, sizeof(float) * 2 * (size_t)n, hipMemcpyHostToDevice, c.stream));
        launch_sample_texture(c.stream, c.tex.table.as<Texture>(), c.tex.lut.as<float>(), handle, d_uv.as<float>(), n, d_out.as<float>());
        LPRT_HIP(hipGetLastError());
        LPRT_HIP(hipMemcpyAsync(rgba, d_out.ptr, sizeof(float) * 4 * (size_t)n, hipMemcpyDeviceToHost, c.stream));
        LPRT_HIP(hipStreamSynchronize(c.stream));
    });
}
int lprt_debug_fmath(lprt_ctx *ctx, int op, const float *in, uint32_t n, float *out) {
    return guarded(ctx, [&](CtxBase &c) {
        if (op < 0 || op > 3)
            throw std::string("unknown fmath op");
        if (n && (!in || !out))
            throw std::string("null in/out pointer");
        if (!n)
            return;
        DevBuf d_in, d_out;
        d_in.reserve(sizeof(float) * (size_t)n);
        d_out.reserve(sizeof(float) * (size_t)n);
        LPRT_HIP(hipMemcpyAsync(d_in.ptr, in, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, c.stream));
        launch_fmath(c.stream, op, d_in.as<float>(), n, d_out.as<float>());
        LPRT_HIP(hipGetLastError());
        LPRT_HIP(hipMemcpyAsync(out, d_out.ptr, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c.stream));
        LPRT_HIP(hipStreamSynchronize(c.stream));
    });
}
int lprt_build_blas(lprt_ctx *ctx) {
    return guarded(ctx, [&](CtxBase &c) { c.build_blas(); });
}
int lprt_set_scene(lprt_ctx *ctx, const lprt_object *objects, const lprt_aabb *aabbs, uint32_t n_objects, const lprt_light *lights,
                   uint32_t n_lights, const lprt_camera *camera, const float transform_W2C[16], const lprt_skybox *skybox) {
    return guarded(ctx, [&](CtxBase &c) {
        if ((n_objects && (!objects || !aabbs)) || (n_lights && !lights) || !camera || !transform_W2C)
            throw std::string("null scene pointer");
        if (n_objects > 1024)
            throw std::string("more than 1024 object instances are not supported by the single-workgroup TLAS build");
        c.set_scene(objects, aabbs, n_objects, lights, n_lights, camera, transform_W2C, skybox);
    });
}
int lprt_trace_rays(lprt_ctx *ctx, const lprt_ray *rays, uint32_t n_rays, int find_any, lprt_hit *hits) {
    return guarded(ctx, [&](CtxBase &c) {
        if (n_rays && (!rays || !hits))
            throw std::string("null ray/hit pointer");
        c.trace_rays(rays, n_rays, find_any, hits);
    });
}
int lprt_enable_counters(lprt_ctx *ctx, int enable) {
    return guarded(ctx, [&](CtxBase &c) { c.counters_enabled = enable != 0; });
}
int lprt_get_counters(lprt_ctx *ctx, lprt_counters *out) {
    return guarded(ctx, [&](CtxBase &c) {
        if (!out)
            throw std::string("null out pointer");
        c.get_counters(-1, out);
    });
}
int lprt_get_stage_counters(lprt_ctx *ctx, lprt_stage_id stage, lprt_counters *out) {
    return guarded(ctx, [&](CtxBase &c) {
        if (!out || (int)stage < 0 || (int)stage >= LPRT_STAGE_COUNT)
            throw std::string("bad stage id or null out pointer");
        c.get_counters((int)stage, out);
    });
}
int lprt_set_output_target(lprt_ctx *ctx, void *dst, uint32_t first_row, uint32_t n_rows) {
    return guarded(ctx, [&](CtxBase &c) {
        if (dst && (first_row < c.win.row_begin || (uint64_t)first_row + n_rows > c.win.row_end))
            throw std::string("output target rows must lie inside this context's row window");
        c.ext_out = dst;
        c.ext_first_row = first_row;
        c.ext_rows = dst ? n_rows : 0;
    });
}
int lprt_history_row_bytes(lprt_ctx *ctx, uint32_t groups, size_t *bytes) {
    return guarded(ctx, [&](CtxBase &c) {
        if (!bytes)
            throw std::string("null out pointer");
        *bytes = c.history_row_bytes(groups);
    });
}
int lprt_export_history_rows(lprt_ctx *ctx, uint32_t groups, uint32_t first_row, uint32_t n_rows, void *dst_device) {
    return guarded(ctx, [&](CtxBase &c) {
        if (n_rows && !dst_device)
            throw std::string("null history buffer");
        c.copy_history_rows(groups, first_row, n_rows, dst_device, true);
    });
}
int lprt_import_history_rows(lprt_ctx *ctx, uint32_t groups, uint32_t first_row, uint32_t n_rows, const void *src_device) {
    return guarded(ctx, [&](CtxBase &c) {
        if (n_rows && !src_device)
            throw std::string("null history buffer");
        c.copy_history_rows(groups, first_row, n_rows, const_cast<void *>(src_device), false);
    });
}
int lprt_debug_rng_states(lprt_ctx *ctx, uint64_t seed, uint32_t first, uint32_t count, uint32_t *out, size_t out_words) {
    return guarded(ctx, [&](CtxBase &c) {
        if (!out)
            throw std::string("null out pointer");
        c.rng_states(seed, first, count, out, out_words);
    });
}
int lprt_set_option(lprt_ctx *ctx, lprt_option option, int value) {
    return guarded(ctx, [&](CtxBase &c) {
        if ((int)option < 0 || (int)option >= LPRT_OPT_COUNT)
            throw std::string("unknown option");
        const bool tri = option == LPRT_OPT_WIDE_WALK || option == LPRT_OPT_TLAS_MODE || option == LPRT_OPT_RAY_STREAM; // -1 = automatic
        if (option == LPRT_OPT_BLAS_BUILDER) {
            if (value < -1 || value > 2)
                throw std::string("option value out of range");
            c.opts[option] = value;
            return;
        }
        if (option == LPRT_OPT_RAY_SORT) { // a bit set
            if (value < -1 || value > 3)
                throw std::string("option value out of range");
            c.opts[option] = value;
            return;
        }
        const bool lanes = option == LPRT_OPT_STREAM_REFILL_MIN || option == LPRT_OPT_STREAM_NODE_MIN;       // a lane count
        if (option == LPRT_OPT_SAMPLE_BATCH ? (value < 1 || value > (int)LPRT_MAX_SAMPLE_BATCH)
                                            : lanes ? (value < 1 || value > 64) : (value > 1 || value < (tri ? -1 : 0)))
            throw std::string("option value out of range");
        c.opts[option] = value;
    });
}
int lprt_get_option(lprt_ctx *ctx, lprt_option option, int *value) {
    return guarded(ctx, [&](CtxBase &c) {
        if ((int)option < 0 || (int)option >= LPRT_OPT_COUNT || !value)
            throw std::string("unknown option or null out pointer");
        *value = c.opts[option];
    });
}
int lprt_reset_counters(lprt_ctx *ctx) {
    return guarded(ctx, [&](CtxBase &c) { c.reset_counters(); });
}
int lprt_get_blas_info(lprt_ctx *ctx, uint32_t vao, uint32_t *n_tris, uint32_t *n_nodes, uint32_t *depth) {
    return guarded(ctx, [&](CtxBase &c) { c.blas_info(vao, n_tris, n_nodes, depth); });
}
int lprt_read_blas_order(lprt_ctx *ctx, uint32_t vao, uint32_t *order, uint32_t n) {
    return guarded(ctx, [&](CtxBase &c) { c.read_blas_order(vao, order, n); });
}
int lprt_read_blas_nodes(lprt_ctx *ctx, uint32_t vao, float *root6, float *boxes, uint32_t *leaf_words, uint32_t n_nodes) {
    return guarded(ctx, [&](CtxBase &c) { c.read_blas_nodes(vao, root6, boxes, leaf_words, n_nodes); });
}

} // extern "C"

// ---- frame pipeline -------------------------------------------------------------------------------------
extern "C" {
int lprt_render_frame(lprt_ctx *ctx, const lprt_frame_settings *fs) {
    return guarded(ctx, [&](CtxBase &c) {
        if (!fs)
            throw std::string("null settings");
        if (fs->spp > 4096)
            throw std::string("spp too large");
        c.render_frame(*fs);
    });
}
int lprt_render_phase(lprt_ctx *ctx, const lprt_frame_settings *fs, int phase) {
    return guarded(ctx, [&](CtxBase &c) {
        if (!fs)
            throw std::string("null settings");
        if (fs->spp > 4096)
            throw std::string("spp too large");
        if (phase != LPRT_PHASE_SAMPLE && phase != LPRT_PHASE_RESOLVE)
            throw std::string("unknown render phase");
        c.render_phase(*fs, phase);
    });
}
int lprt_set_sample_rows(lprt_ctx *ctx, uint32_t first_row, uint32_t n_rows) {
    return guarded(ctx, [&](CtxBase &c) { c.set_sample_rows(first_row, n_rows); });
}
int lprt_reset_history(lprt_ctx *ctx) {
    return guarded(ctx, [&](CtxBase &c) { c.reset_history(); });
}
int lprt_read_buffer(lprt_ctx *ctx, lprt_buffer_id id, float *dst, size_t n_floats) {
    return guarded(ctx, [&](CtxBase &c) {
        if (!dst)
            throw std::string("null destination");
        c.read_buffer(id, dst, n_floats);
    });
}
int lprt_output_device_ptr(lprt_ctx *ctx, void **ptr, size_t *bytes) {
    return guarded(ctx, [&](CtxBase &c) {
        if (!ptr || !bytes)
            throw std::string("null out pointer");
        c.output_ptr(ptr, bytes);
    });
}
int lprt_enable_stage_timing(lprt_ctx *ctx, int enable) {
    return guarded(ctx, [&](CtxBase &c) {
        c.timer.enabled = enable != 0;
        c.timer.mask = 0xffffffffu;
    });
}
int lprt_set_stage_timing_mask(lprt_ctx *ctx, uint32_t stage_mask) {
    return guarded(ctx, [&](CtxBase &c) {
        c.timer.enabled = stage_mask != 0;
        c.timer.mask = stage_mask;
    });
}
int lprt_get_stage_times(lprt_ctx *ctx, float *ms, uint32_t *launches, uint32_t n) {
    return guarded(ctx, [&](CtxBase &c) {
        if (n != LPRT_STAGE_COUNT)
            throw std::string("stage arrays must hold LPRT_STAGE_COUNT entries (lprt_get_abi_info().stage_count)");
        c.collect_times();
        for (int i = 0; i < LPRT_STAGE_COUNT; i++) {
            if (ms)
                ms[i] = c.timer.ms[i];
            if (launches)
                launches[i] = c.timer.launches[i];
        }
    });
}
int lprt_reset_stage_times(lprt_ctx *ctx) {
    return guarded(ctx, [&](CtxBase &c) {
        c.collect_times();
        for (int i = 0; i < LPRT_STAGE_COUNT; i++) {
            c.timer.ms[i] = 0;
            c.timer.launches[i] = 0;
        }
    });
}
}
