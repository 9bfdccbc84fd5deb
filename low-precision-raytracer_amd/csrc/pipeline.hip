// Per-frame wavefront pipeline kernels: primary rays -> G-buffer, temporal reprojection maps, shade,
// shadow/GI tracing, composition.  One thread per pixel of the window; all per-pixel planes are SoA or
// 8/16-byte vectors so that a wave's accesses are contiguous.
//
// Reference stage order: Renderer::render_gbuffer + render_cuda (rt/gui/imgui_window.cu:121-329).
// Differences that do not change results: the five per-frame memsets (imgui_window.cu:136-140) are gone
// because every kernel writes all of its outputs; the 4N-thread trace_di_light + accumulate_di_light pair
// is one kernel that walks a pixel's active light slots in slot order; TraceLightCommand /
// TraceGICommand / PixelShaderInput AoS records are split into planes.
#include "kernels.hpp"
#include "shade.hpp"
#include "trace.hpp"
#include "stream.hpp"
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>

// Waves per SIMD the traversal kernels are compiled for (VGPR budget = 512 / waves).  Traversal is latency-bound, so
// the closest-hit kernels take a fifth wave at the price of a few spilled registers (measured on MI355X, Cornell 1080p:
// k_trace_gi 235 -> 213 us per launch); the shadow-ray kernel loses more to its spills than it gains (129 -> 154 us).
#ifndef LPRT_CLOSEST_WAVES
#define LPRT_CLOSEST_WAVES 5
#endif
#ifndef LPRT_CLOSEST_WAVES_F16
#define LPRT_CLOSEST_WAVES_F16 6 // half: 80 registers with 10-11 spilled (single precision would spill 34-83): Cornell fp16 GI 0.65 -> 0.62 ms
#endif
template <typename T> struct ClosestWaves { static constexpr int value = sizeof(T) == 2 ? LPRT_CLOSEST_WAVES_F16 : LPRT_CLOSEST_WAVES; };
#ifndef LPRT_ANYHIT_WAVES
#define LPRT_ANYHIT_WAVES 4
#endif
#ifndef LPRT_ANYHIT_WAVES_F16
#define LPRT_ANYHIT_WAVES_F16 5 // the half instantiation fits 96 registers with 4-10 spilled (single precision: 33-44)
#endif
template <typename T> struct AnyhitWaves { static constexpr int value = sizeof(T) == 2 ? LPRT_ANYHIT_WAVES_F16 : LPRT_ANYHIT_WAVES; };

namespace lprt {

// A workgroup of the per-pixel kernels covers a 32x8-pixel tile and each wave an 8x8 block of it (instead of 256 / 64
// pixels of one row): the rays of a wave start close together in both directions, so they walk the same part of the
// BVH (atrium shadow rays 15.7 -> 11 ms per frame, Cornell frame 3.45 -> 3.35 ms).  Rows stay the unit of memory
// layout: a wave touches 8 row segments of 8 pixels, a full 128-B line for the 16-B planes.
#ifndef LPRT_TILE_W
#define LPRT_TILE_W 32u
#define LPRT_TILE_H 8u
#define LPRT_WAVE_W 8u
#define LPRT_WAVE_H 8u
#endif
static_assert(LPRT_WAVE_W * LPRT_WAVE_H == 64u && (LPRT_TILE_W / LPRT_WAVE_W) * (LPRT_TILE_H / LPRT_WAVE_H) == 4u, "a workgroup is four waves");
template <typename T> LPRT_DEV uint32_t tile_pixel(const FrameDev<T> &fb) {
    const uint32_t W = fb.win.W, R = fb.win.rows();
    const uint32_t tiles_x = (W + LPRT_TILE_W - 1u) / LPRT_TILE_W;
    const uint32_t tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t wx = wave % (LPRT_TILE_W / LPRT_WAVE_W), wy = wave / (LPRT_TILE_W / LPRT_WAVE_W);
    const uint32_t x = tx * LPRT_TILE_W + wx * LPRT_WAVE_W + lane % LPRT_WAVE_W, y = ty * LPRT_TILE_H + wy * LPRT_WAVE_H + lane / LPRT_WAVE_W;
    return (x < W && y < R) ? y * W + x : 0xffffffffu;
}

// Stable counting sort of the 256 work items of a workgroup by a small key (< 16 bins): returns the index of the thread
// whose item this thread should process.  Items with equal keys end up in the same waves, so a wave walks rays that point
// into the same octant (or start in the same part of the scene), and the items without work (key = n_bins - 1) gather in
// the last waves, which then leave at once.  Per-pixel results do not depend on which thread computes them.
// `payload` travels with the item: on return it is the payload of the item this thread got (what the thread already knows about
// its own pixel -- a light mask, a validity flag -- need not be fetched again for the pixel it is handed)
LPRT_DEV uint32_t block_sort_order(uint32_t key, uint32_t n_bins, uint32_t &payload) {
    __shared__ uint16_t s_cnt[16][4];
    __shared__ uint16_t s_order[256];
    __shared__ uint8_t s_pay[256];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    uint32_t below_in_wave = 0;
    for (uint32_t b = 0; b < n_bins; b++) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(key == b);
        if (lane == 0)
            s_cnt[b][wave] = (uint16_t)__builtin_popcountll(m);
        if (key == b)
            below_in_wave = (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t b = 0; b < n_bins; b++)
        for (uint32_t w = 0; w < 4; w++)
            if (b < key || (b == key && w < wave))
                base += s_cnt[b][w];
    s_order[base + below_in_wave] = (uint16_t)threadIdx.x;
    s_pay[base + below_in_wave] = (uint8_t)payload;
    __syncthreads();
    payload = s_pay[threadIdx.x];
    return s_order[threadIdx.x];
}
LPRT_DEV uint32_t block_sort_order(uint32_t key, uint32_t n_bins) {
    uint32_t none = 0u;
    return block_sort_order(key, n_bins, none);
}
// pixel of work item `item` (0..255) of this workgroup's 32x8 tile
template <typename T> LPRT_DEV uint32_t tile_pixel_of(const FrameDev<T> &fb, uint32_t item) {
    const uint32_t W = fb.win.W, R = fb.win.rows();
    const uint32_t tiles_x = (W + LPRT_TILE_W - 1u) / LPRT_TILE_W;
    const uint32_t tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const uint32_t wave = item >> 6, lane = item & 63u;
    const uint32_t wx = wave % (LPRT_TILE_W / LPRT_WAVE_W), wy = wave / (LPRT_TILE_W / LPRT_WAVE_W);
    const uint32_t x = tx * LPRT_TILE_W + wx * LPRT_WAVE_W + lane % LPRT_WAVE_W, y = ty * LPRT_TILE_H + wy * LPRT_WAVE_H + lane / LPRT_WAVE_W;
    return (x < W && y < R) ? y * W + x : 0xffffffffu;
}
template <typename T> static inline dim3 tile_grid(const FrameDev<T> &fb) { return dim3(((fb.win.W + LPRT_TILE_W - 1u) / LPRT_TILE_W) * ((fb.win.rows() + LPRT_TILE_H - 1u) / LPRT_TILE_H)); }

template <typename T> LPRT_DEV V3<T> ld3(const T *p) { return mk3(p[0], p[1], p[2]); }
LPRT_DEV uint32_t pack_rgba8(uint32_t r, uint32_t g, uint32_t b, uint32_t a) { return r | (g << 8) | (b << 16) | (a << 24); }

// attribute interpolation shared by fill_gbuffer (cuda.hpp:967-987) and trace_gi (cuda.hpp:2107-2128)
template <typename T> struct HitAttr {
    V3<T> position, normal, tangent, color;
    V2<T> uv0, uv1;
};
template <typename T> LPRT_DEV HitAttr<T> interpolate_hit(const SceneDev<T> &sc, const Isect<T> &is) {
    HitAttr<T> h{};
    LPRT_CHECKA_RET(sc, is.inscene_objid < sc.lim_objects, 101, is.inscene_objid, sc.lim_objects, h);
    const ObjConst<T> *oc = sc.objects + is.inscene_objid;
    LPRT_CHECKA_RET(sc, oc->objectid < sc.n_vaos, 102, oc->objectid, sc.n_vaos, h);
    const lprt_mesh_vao vao = sc.vaos[oc->objectid];
    LPRT_CHECKA_RET(sc, vao.ebo_offset + is.inobject_offset + 2 < sc.lim_indices, 103, vao.ebo_offset, is.inobject_offset, h);
    const uint32_t *e = sc.ebo + vao.ebo_offset + is.inobject_offset;
    LPRT_CHECKA_RET(sc, e[0] < sc.lim_vertices && e[1] < sc.lim_vertices && e[2] < sc.lim_vertices, 104, e[0], e[2], h);
    const Vertex<T> a = sc.vbo[e[0]], b = sc.vbo[e[1]], c = sc.vbo[e[2]];
    V3<T> position = lerp3(is.bary, ld3(a.position), ld3(b.position), ld3(c.position));
    V3<T> normal = normalized(lerp3(is.bary, ld3(a.normal), ld3(b.normal), ld3(c.normal)));
    h.color = lerp3(is.bary, ld3(a.color), ld3(b.color), ld3(c.color));
    V3<T> tangent = normalized(lerp3(is.bary, ld3(a.tangent), ld3(b.tangent), ld3(c.tangent)));
    h.uv0 = lerp2(is.bary, V2<T>{a.uv0[0], a.uv0[1]}, V2<T>{b.uv0[0], b.uv0[1]}, V2<T>{c.uv0[0], c.uv0[1]});
    h.uv1 = lerp2(is.bary, V2<T>{a.uv1[0], a.uv1[1]}, V2<T>{b.uv1[0], b.uv1[1]}, V2<T>{c.uv1[0], c.uv1[1]});
    h.normal = normalized(xform_dir(oc->L2W, normal));
    h.tangent = normalized(xform_dir(oc->L2W, tangent));
    h.position = xform_pos(oc->L2W, position);
    return h;
}

// ---- G2: fill_gbuffer_in_cuda (cuda.hpp:950-1012) ---------------------------------------------------------
LPRT_VAR_BEGIN
template <typename T, bool COUNT>
__global__ void __launch_bounds__(256, ClosestWaves<T>::value) k_fill_gbuffer(SceneDev<T> sc, FrameDev<T> fb, Counters *counters) {
    const uint32_t p = tile_pixel(fb);
    Cnt<COUNT> cnt;
    if (p < fb.win.npix()) {
        const uint32_t x = p % fb.win.W, y = fb.win.row_begin + p / fb.win.W;
        RayT<T> ray = primary_ray(sc, fb.win.W, fb.win.H, x, y);
        Isect<T> is = cast_ray<T, false, COUNT>(sc, ray, 0xffffffffu, 0u, (T)0, cvt<T>(1e5f), cnt);
        // All seven planes are written by ONE unconditional store sequence below.  (An if/else with stores in
        // both arms made hipcc 7.2 sink the stores into a block that indexed them with a register only the hit
        // arm had defined: the miss lanes of a mixed wave stored through garbage addresses.)
        const T one = (T)1, zero = (T)0;
        V4<T> o_pos = mk4(zero, zero, zero, zero), o_nd = o_pos, o_tan = o_pos, o_uv = o_pos;
        uint32_t o_col = 0u, o_obj = 0u, o_off = 0u;
        if (is.hit) {
            HitAttr<T> h = interpolate_hit(sc, is);
            o_pos = mk4(h.position.x, h.position.y, h.position.z, one);
            o_nd = mk4(h.normal.x, h.normal.y, h.normal.z, one); // Q11: traced depth = 1
            o_tan = mk4(h.tangent.x, h.tangent.y, h.tangent.z, one);
            // to_color_uchar (cuda.hpp:898-902): int(0.5f + c * DataT(255)), clamped
            int r = clip_both<int>(0, 255, f2i(0.5f + (float)(h.color.x * cvt<T>(255.f))));
            int g = clip_both<int>(0, 255, f2i(0.5f + (float)(h.color.y * cvt<T>(255.f))));
            int b = clip_both<int>(0, 255, f2i(0.5f + (float)(h.color.z * cvt<T>(255.f))));
            o_col = pack_rgba8((uint32_t)r, (uint32_t)g, (uint32_t)b, 255u);
            o_uv = mk4(h.uv0.x, h.uv0.y, h.uv1.x, h.uv1.y);
            o_obj = is.inscene_objid;
            o_off = is.inobject_offset;
        }
        fb.g_position[p] = o_pos;
        fb.g_normal_depth[p] = o_nd;
        fb.g_tangent[p] = o_tan;
        fb.g_color[p] = o_col;
        fb.g_uv[p] = o_uv;
        fb.g_objid[p] = o_obj;
        fb.g_offset[p] = o_off;
    }
    flush_counters<COUNT>(cnt, counters);
}
LPRT_VAR_END

// ---- V1: generate_temporal_map_step1/2 (cuda.hpp:2265-2416) ------------------------------------------------
// `first` = no history yet (DEVIATION Q7: frame_count = 0 everywhere).  The neighbours' previous frame counts
// come from a snapshot plane (prev_frame_count), which replaces the reference's racy two-step scheme.
#ifndef LPRT_REFSLAB_TU
template <typename T>
__global__ void __launch_bounds__(256) k_temporal_map(SceneDev<T> sc, FrameDev<T> fb, int first, uint32_t rng_d, int floor_taps) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t npix = fb.win.npix();
    if (p >= npix)
        return;
    const T zero = (T)0;
    TemporalMapT<T> sm, tm;
    for (int i = 0; i < 4; i++) {
        sm.weights[i] = zero;
        tm.weights[i] = zero;
        sm.pos[i][0] = sm.pos[i][1] = tm.pos[i][0] = tm.pos[i][1] = 0;
    }
    sm.frame_count = 0;
    tm.frame_count = 0;
    const bool empty = (fb.g_color[p] >> 24) == 0;
    if (!empty && !first) {
        const int W = (int)fb.win.W, H = (int)fb.win.H, r0 = (int)fb.win.row_begin, r1 = (int)fb.win.row_end;
        const ObjConst<T> *oc = sc.objects + fb.g_objid[p];
        const uint32_t objectid = oc->objectid;
        V4<T> cW = fb.g_position[p];
        cW.w = (T)1;
        V4<T> cL = mul(oc->W2L, cW);
        V4<T> cWlf = mul(fb.last_frame_L2W[objectid], cL);
        V4<T> cC = mul(fb.last_frame_W2C, cWlf);
        const T one = (T)1, two = (T)2;
        T g_fx = tdiv(one + tdiv(cC.x, cC.w), two) * cvt<T>((float)W);
        T g_fy = tdiv(one + tdiv(cC.y, cC.w), two) * cvt<T>((float)H);
        // rows outside this GPU's window count as outside the image (sharded frames)
        { // svgf 2x2, pixel centre
            T fx = g_fx - cvt<T>(0.5f), fy = g_fy - cvt<T>(0.5f);
            // f2i truncates toward zero as the reference's int() does: a position in (-1, 0) keeps lx = 0 and extrapolates
            // (LPRT_OPT_FLOOR_REPROJECTION = 1: floor)
            int lx = floor_taps ? (int)__builtin_floorf((float)fx) : f2i((float)fx), ly = floor_taps ? (int)__builtin_floorf((float)fy) : f2i((float)fy);
            int ux = lx + 1, uy = ly + 1;
            int nx[4] = {lx, lx, ux, ux}, ny[4] = {ly, uy, ly, uy};
            T tux = cvt<T>((float)ux), tuy = cvt<T>((float)uy), tlx = cvt<T>((float)lx), tly = cvt<T>((float)ly);
            T wts[4] = {(tux - fx) * (tuy - fy), (tux - fx) * (fy - tly), (fx - tlx) * (tuy - fy), (fx - tlx) * (fy - tly)};
            T total = zero;
            int count = 0;
            for (int i = 0; i < 4; i++)
                if (nx[i] >= 0 && nx[i] < W && ny[i] >= r0 && ny[i] < r1) {
                    uint32_t q = (uint32_t)(ny[i] - r0) * (uint32_t)W + (uint32_t)nx[i];
                    if (fb.last_frame_id[q].x == objectid) {
                        count = max(count, (int)fb.prev_frame_count[q]);
                        sm.weights[i] = wts[i];
                        total = total + wts[i];
                        sm.pos[i][0] = (uint16_t)nx[i];
                        sm.pos[i][1] = (uint16_t)(ny[i] - r0);
                    }
                }
            if (!(total == zero)) {
                for (int i = 0; i < 4; i++) {
                    if (count < 255)
                        count += 1; // Q6: +4 per frame
                    sm.weights[i] = tdiv(sm.weights[i], total);
                }
                sm.frame_count = (uint32_t)count;
            }
        }
        { // taa 2x2, jittered with two uniforms of curand_init(seed A, pixel, 0)
            Xorwow st;
            const uint32_t *rs = fb.rng_state;
            st.v0 = rs[p];
            st.v1 = rs[npix + p];
            st.v2 = rs[2 * npix + p];
            st.v3 = rs[3 * npix + p];
            st.v4 = rs[4 * npix + p];
            st.d = rng_d;
            st.bm_flag = 0;
            st.bm_extra = 0.f;
            T fx = g_fx - cvt<T>(xorwow_uniform(st));
            T fy = g_fy - cvt<T>(xorwow_uniform(st));
            // f2i truncates toward zero as the reference's int() does: a position in (-1, 0) keeps lx = 0 and extrapolates
            // (LPRT_OPT_FLOOR_REPROJECTION = 1: floor)
            int lx = floor_taps ? (int)__builtin_floorf((float)fx) : f2i((float)fx), ly = floor_taps ? (int)__builtin_floorf((float)fy) : f2i((float)fy);
            int ux = lx + 1, uy = ly + 1;
            int nx[4] = {lx, lx, ux, ux}, ny[4] = {ly, uy, ly, uy};
            T tux = cvt<T>((float)ux), tuy = cvt<T>((float)uy), tlx = cvt<T>((float)lx), tly = cvt<T>((float)ly);
            T wts[4] = {(tux - fx) * (tuy - fy), (tux - fx) * (fy - tly), (fx - tlx) * (tuy - fy), (fx - tlx) * (fy - tly)};
            T total = zero;
            bool same = false;
            for (int i = 0; i < 4; i++)
                if (nx[i] >= 0 && nx[i] < W && ny[i] >= r0 && ny[i] < r1) {
                    uint32_t q = (uint32_t)(ny[i] - r0) * (uint32_t)W + (uint32_t)nx[i];
                    tm.weights[i] = wts[i];
                    total = total + wts[i];
                    tm.pos[i][0] = (uint16_t)nx[i];
                    tm.pos[i][1] = (uint16_t)(ny[i] - r0);
                    if (fb.last_frame_id[q].x == objectid)
                        same = true;
                }
            if (same) {
                for (int i = 0; i < 4; i++)
                    tm.weights[i] = tdiv(tm.weights[i], total);
                tm.frame_count = 1;
            }
        }
    }
    fb.svgf_map[p] = sm;
    fb.taa_map[p] = tm;
    fb.valid_frame_count[p] = (uint8_t)sm.frame_count;
}

// ---- S: shade<T,4,FIRST_ROUND,NO_GI> (cuda.hpp:1751-2056) ----------------------------------------------------
// The surface point shade() works on: a G-buffer texel (round 0, loaded by k_shade) or the interpolated GI hit
// (round 1, handed over in registers by k_trace_gi: the PixelShaderInput record never goes through memory).
#endif // LPRT_REFSLAB_TU
template <typename T> struct ShadeIn {
    V3<T> position, raw_normal, raw_tangent, vertex_color;
    V4<T> uv4;
    MaterialT<T> material;
    uint32_t inobject_offset, objectid;
    bool live, sky_pixel;
};

template <typename T, bool FIRST, bool NO_GI>
LPRT_DEV void shade_and_store(const SceneDev<T> &sc, const FrameDev<T> &fb, uint32_t p, uint32_t rng_d, const ShadeIn<T> &in, bool first_sample,
                              bool skip_direct = false) {
    const uint32_t npix = fb.win.npix();
    const uint32_t x = p % fb.win.W, y = fb.win.row_begin + p / fb.win.W;
    const T zero = (T)0, one = (T)1;
    const V4<T> zero4 = mk4(zero, zero, zero, zero);
    V4<T> *intensity = FIRST ? fb.intensity0 : fb.intensity1;
    const V3<T> position = in.position, raw_normal = in.raw_normal, raw_tangent = in.raw_tangent, vertex_color = in.vertex_color;
    const MaterialT<T> &material = in.material;
    const uint32_t inobject_offset = in.inobject_offset, objectid = in.objectid;
    const V4<T> uv4 = in.uv4;
    bool live = in.live;
    const bool sky_pixel = in.sky_pixel;
    V3<T> color, normal, view_dir;
    if (live) {
        color = ld3(material.color);
        if (material.uv_color < 2u) { // cuda.hpp:1850-1855
            const float4 tc = material.uv_color == 0u ? tex2d(sc.textures, sc.srgb_lut, material.tex_color, (float)uv4.x, (float)uv4.y)
                                                      : tex2d(sc.textures, sc.srgb_lut, material.tex_color, (float)uv4.z, (float)uv4.w);
            color = mk3(cvt<T>(tc.x), cvt<T>(tc.y), cvt<T>(tc.z));
        }
        color = elemul(color, vertex_color);
        normal = raw_normal;
        if (FIRST)
            view_dir = neg(primary_ray_direction(sc, fb.win.W, fb.win.H, x, y));
        else
            view_dir = neg(xyz(fb.gi_direction[p])); // view_direction of round 1 = -(GI direction) (cuda.hpp:1985)
        if (dot(view_dir, normal) < zero) {
            if (material.double_sided)
                normal = neg(normal);
            else
                live = false; // cuda.hpp:1873-1877
        }
    }
    // Outputs are collected in registers and written by one unconditional store sequence at the end (see the
    // note in k_fill_gbuffer about stores in both arms of a branch).
    V4<T> o_intensity = zero4, o_albedo = zero4, o_mult = zero4, o_gidir = zero4;
    uint2 o_skip = make_uint2(0xffffffffu, 0u);
    uint32_t o_gi_valid = 0, mask = 0;
    if (live) {
        V3<T> bitangent = normalized(cross(raw_normal, raw_tangent));
        V3<T> tangent = normalized(cross(bitangent, raw_normal));
        o_skip = make_uint2(objectid, inobject_offset);
        V3<T> emission = ld3(material.emission);
        o_albedo = mk4(color.x, color.y, color.z, zero);

        if (!NO_GI) {
            Xorwow rs;
            const uint32_t *st = fb.rng_state;
            rs.v0 = st[p];
            rs.v1 = st[npix + p];
            rs.v2 = st[2 * npix + p];
            rs.v3 = st[3 * npix + p];
            rs.v4 = st[4 * npix + p];
            rs.d = rng_d;
            rs.bm_flag = 0;
            rs.bm_extra = 0.f;
            V3<T> dir, mult;
            // NaN in .z tags "split lobes for demodulated SVGF" (cuda.hpp:1919)
            const T nanT = cvt<T>(__builtin_nanf(""));
            const T metallic = material.metallic, roughness = material.roughness;
            if (roughness < cvt<T>(0.1f)) {
                if (cvt<T>(xorwow_uniform(rs)) < metallic) {
                    dir = normalized(reflect_about(view_dir, normal));
                    mult = vdiv(get_brdf(glassy_brdf(metallic, view_dir, dir, normal), color), metallic);
                } else if (cvt<T>(xorwow_uniform(rs)) < cvt<T>(0.6f)) {
                    dir = normalized(reflect_about(view_dir, normal));
                    Brdf<T> b = glassy_brdf(metallic, view_dir, dir, normal);
                    T pdf = (one - metallic) * cvt<T>(0.6f);
                    if (FIRST)
                        mult = mk3(tdiv(b.colored, pdf), tdiv(b.white, pdf), nanT);
                    else
                        mult = vdiv(get_brdf(b, color), pdf);
                } else { // Q10: uniform direction from three normals
                    T n0 = cvt<T>(xorwow_normal(rs)), n1 = cvt<T>(xorwow_normal(rs)), n2 = cvt<T>(xorwow_normal(rs));
                    V3<T> rd = normalized(mk3(n0, n1, n2));
                    T cosine = dot(rd, normal);
                    if (cosine < zero) {
                        rd = neg(rd);
                        cosine = tneg(cosine);
                    }
                    dir = rd;
                    Brdf<T> b = material_brdf(metallic, roughness, view_dir, rd, normal);
                    T m = cosine * cvt<T>((float)(LPRT_PI_D * 2));
                    if (FIRST)
                        mult = mk3(b.colored * m, b.white * m, nanT);
                    else
                        mult = get_brdf(b, color) * m;
                }
            } else {
                T pdf, cosine;
                V3<T> rd;
                if (cvt<T>(xorwow_uniform(rs)) < metallic) {
                    T a = roughness * roughness;
                    T a2 = a * a;
                    V3<T> ht = sample_ggx(a2, rs);
                    pdf = pdf_ggx_reflect(ht.z, a2) * metallic;
                    V3<T> hw = normalized(tangent * ht.x + bitangent * ht.y + normal * ht.z); // tangent_to_world (cuda.hpp:1737-1740)
                    rd = reflect_about(view_dir, hw);
                    cosine = tmax(cvt<T>(0.05f), dot(rd, hw));
                    pdf = tdiv(pdf, cvt<T>(4.0f * (float)cosine)); // `4 * cosine`: int * DataT is formed in float
                } else {
                    rd.x = cvt<T>(xorwow_normal(rs));
                    rd.y = cvt<T>(xorwow_normal(rs));
                    rd.z = cvt<T>(xorwow_normal(rs));
                    cosine = dot(rd, normal);
                    if (cosine < zero) {
                        rd = neg(rd);
                        cosine = tneg(cosine);
                    }
                    pdf = cvt<T>((float)(0.5 / LPRT_PI_D)) * (one - metallic);
                }
                rd = normalized(rd);
                dir = rd;
                Brdf<T> b = material_brdf(metallic, roughness, view_dir, rd, normal);
                T m = tdiv(cosine, pdf);
                if (FIRST)
                    mult = mk3(b.colored * m, b.white * m, nanT);
                else
                    mult = m * get_brdf(b, color);
            }
            o_gi_valid = 1;
            o_gidir = mk4(dir.x, dir.y, dir.z, one); // w = 1: a ray is cast (read-back only)
            o_mult = mk4(mult.x, mult.y, mult.z, zero);
            o_intensity = mk4(emission.x, emission.y, emission.z, zero);
        } else {
            // Q4: without GI the emission is overwritten by SKYBOX_COLOR (0) * color * 0.5 (cuda.hpp:2001-2006)
            V3<T> sky = elemul(zero3<T>(), color) * cvt<T>(0.5f);
            o_intensity = mk4(sky.x, sky.y, sky.z, zero);
        }

        // direct light commands (cuda.hpp:2009-2055).  skip_direct (opt-in, samples >= 1 of round 0): the commands and the
        // visibility they lead to depend on the G-buffer and the lights only, so sample 0's radiance plane is kept as is
        const T rough_l = tmax(material.roughness, cvt<T>(0.10f));
        for (uint32_t i = 0; i < LPRT_MAX_DIRECT_LIGHT; i++) {
            if (i >= sc.n_lights || skip_direct)
                break;
            const LightT<T> light = sc.lights[i];
            V3<T> light_dir, mm = zero3<T>();
            T max_t = zero;
            bool active = false;
            if (light.type == LPRT_LIGHT_POINT || light.type == LPRT_LIGHT_SPOT) {
                V3<T> ldv = ld3(light.position) - position;
                light_dir = normalized(ldv);
                T cosine = dot(light_dir, normal);
                if (!(cosine < zero)) {
                    T d2 = norm2_squared(ldv);
                    V3<T> brdf = get_brdf(material_brdf(material.metallic, rough_l, view_dir, light_dir, normal), color);
                    mm = tdiv(tdiv(cosine, d2), cvt<T>(10.f)) * elemul(brdf, ld3(light.intensity));
                    max_t = tsqrt(d2);
                    active = true;
                }
            } else {
                light_dir = neg(normalized(ld3(light.direction)));
                T cosine = dot(light_dir, normal);
                if (!(cosine < zero)) {
                    V3<T> brdf = get_brdf(material_brdf(material.metallic, rough_l, view_dir, light_dir, normal), color);
                    mm = cosine * elemul(brdf, ld3(light.intensity));
                    max_t = cvt<T>(1000.f);
                    active = true;
                }
            }
            if (active) {
                mask |= 1u << i;
                fb.lc_dir_maxt[i * npix + p] = mk4(light_dir.x, light_dir.y, light_dir.z, max_t);
                fb.lc_multiplier[i * npix + p] = mk4(mm.x, mm.y, mm.z, zero);
            }
        }
    }
    if (sky_pixel && sc.sky_valid) {
        // the reference parks this in di_intensity[4p + 0]; accumulate_di_light then adds 0 + sky + 0 + 0 + 0 to a cleared pixel
        V3<T> dir = FIRST ? primary_ray_direction(sc, fb.win.W, fb.win.H, x, y) : neg(normalized(neg(xyz(fb.gi_direction[p]))));
        V3<T> sky = sample_skybox(sc, dir);
        o_intensity = mk4(sky.x, sky.y, sky.z, zero);
    }
    if (!skip_direct) {
        fb.lc_mask[p] = (uint8_t)mask;
        intensity[p] = o_intensity;
    }
    fb.skips[p] = o_skip;
    if (!FIRST) // the shadow rays of round 1 start at the GI hit
        fb.sc_position[p] = live ? mk4(position.x, position.y, position.z, zero) : zero4;
    if (FIRST) {
        if (first_sample) // the albedo does not depend on the sample
            fb.albedo[p] = o_albedo;
        fb.gi_valid[p] = (uint8_t)o_gi_valid; // DEVIATION Q13: untouched GI commands are invalid
        fb.gi_multiplier[p] = o_mult;
        fb.gi_direction[p] = o_gidir;
    }
}

#ifndef LPRT_REFSLAB_TU
template <typename T, bool NO_GI>
__global__ void __launch_bounds__(256) k_shade(SceneDev<T> sc, FrameDev<T> fb0, SampleBatch sa) {
    const FrameDev<T> fb = at_sample(fb0, blockIdx.z);
    const uint32_t sample = sa.first + blockIdx.z, rng_d = sa.rng_d[blockIdx.z];
    const uint32_t skip_direct = (sa.reuse_direct && sample > 0u) ? 1u : 0u;
    const uint32_t p = tile_pixel(fb);
    if (p >= fb.win.npix())
        return;
    const T zero = (T)0;
    ShadeIn<T> in;
    in.uv4 = mk4(zero, zero, zero, zero);
    in.live = true;
    in.sky_pixel = false;
    in.inobject_offset = 0u;
    in.objectid = 0u;
    const uint32_t c4 = fb.g_color[p];
    // to_color_float (cuda.hpp:890-892): DataT(c / 255.f)
    in.vertex_color = mk3(cvt<T>((float)(c4 & 255u) / 255.f), cvt<T>((float)((c4 >> 8) & 255u) / 255.f), cvt<T>((float)((c4 >> 16) & 255u) / 255.f));
    if ((c4 >> 24) == 0) {
        in.live = false; // empty pixel (cuda.hpp:1781-1796)
        in.sky_pixel = true;
    } else {
        in.uv4 = fb.g_uv[p];
        in.inobject_offset = fb.g_offset[p];
        in.position = xyz(fb.g_position[p]);
        in.raw_normal = xyz(fb.g_normal_depth[p]);
        in.raw_tangent = xyz(fb.g_tangent[p]);
        const ObjConst<T> *oc = sc.objects + fb.g_objid[p];
        in.material = sc.materials[oc->materialid];
        in.objectid = oc->objectid;
    }
    shade_and_store<T, true, NO_GI>(sc, fb, p, rng_d, in, sample == 0u, skip_direct != 0u);
}
#endif // LPRT_REFSLAB_TU

// ---- TD + A (+ W): trace_di_light + accumulate_di_light (cuda.hpp:2058-2081, 2143-2159); the LAST round of a sample
// also runs write_clean_color (cuda.hpp:2170-2199) and the build-side spp mean on the radiance it just completed:
// sample s of n accumulates in f32 in sample order; the last sample divides once and rounds once to DataT.

struct ComposeArgs {
    int gi_on, add_direct_out;
    uint32_t reuse_direct; // round 0 ran for sample 0 only (lprt_frame_settings.reuse_direct_light): its radiance is slot 0's
};
// write_clean_color (cuda.hpp:2170-2199) for pixel p of one sample slot, whose radiance of this round (`cur`) has just been
// completed: the sample's three contributions go to the slot's smp_* planes (DataT, as computed) and k_spp_mean sums the slots in f32 in sample
// order.  slot0: the frame planes of slot 0 (round 0's radiance under reuse_direct_light).
template <typename T, bool FIRST>
LPRT_DEV void compose_tail(const FrameDev<T> &fb, const FrameDev<T> &slot0, const ComposeArgs &ca, uint32_t p, V4<T> cur) {
    V3<T> fc = zero3<T>(), colored = zero3<T>(), white = zero3<T>();
    const V4<T> m0 = fb.gi_multiplier[p];
    if (ca.add_direct_out)
        fc = FIRST ? xyz(cur) : xyz((ca.reuse_direct ? slot0.intensity0 : fb.intensity0)[p]);
    const V3<T> i1 = FIRST ? zero3<T>() : xyz(cur); // gi off: there is no second round
    if (tisnan(m0.z)) {
        colored = m0.x * i1;
        white = m0.y * i1;
    } else if (ca.add_direct_out) {
        fc = fc + elemul(i1, xyz(m0));
    }
    const T zero = (T)0;
    fb.smp_color[p] = mk4(fc.x, fc.y, fc.z, zero);
    fb.smp_gi_colored[p] = mk4(colored.x, colored.y, colored.z, zero);
    fb.smp_gi_white[p] = mk4(white.x, white.y, white.z, zero);
}
// The build-side spp mean: sample s of n accumulates in f32 in sample order; the last sample divides once and rounds once to
// DataT (spp = 1 is exactly a reference frame).  One launch per batch of samples, after the batch's last stage.
#ifndef LPRT_REFSLAB_TU
template <typename T> __global__ void __launch_bounds__(256) k_spp_mean(FrameDev<T> fb, uint32_t first, uint32_t count, uint32_t spp) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= fb.win.npix())
        return;
    const T zero = (T)0;
    float4 ac = make_float4(0.f, 0.f, 0.f, 0.f), ag = ac, aw = ac;
    for (uint32_t z = 0; z < count; z++) {
        const size_t q = (size_t)z * fb.smp_stride + p;
        const V4<T> ct = fb.smp_color[q], gt = fb.smp_gi_colored[q], wt = fb.smp_gi_white[q];
        const float4 c = make_float4((float)ct.x, (float)ct.y, (float)ct.z, 0.f), g = make_float4((float)gt.x, (float)gt.y, (float)gt.z, 0.f),
                     w = make_float4((float)wt.x, (float)wt.y, (float)wt.z, 0.f);
        if (first + z == 0u) {
            ac = c;
            ag = g;
            aw = w;
        } else {
            if (z == 0u) {
                ac = fb.acc_color[p];
                ag = fb.acc_gi_colored[p];
                aw = fb.acc_gi_white[p];
            }
            ac = make_float4(ac.x + c.x, ac.y + c.y, ac.z + c.z, 0.f);
            ag = make_float4(ag.x + g.x, ag.y + g.y, ag.z + g.z, 0.f);
            aw = make_float4(aw.x + w.x, aw.y + w.y, aw.z + w.z, 0.f);
        }
    }
    if (first + count < spp) {
        fb.acc_color[p] = ac;
        fb.acc_gi_colored[p] = ag;
        fb.acc_gi_white[p] = aw;
    } else {
        const float n = (float)spp;
        const V4<T> c = mk4(cvt<T>(ac.x / n), cvt<T>(ac.y / n), cvt<T>(ac.z / n), zero);
        const V4<T> g = mk4(cvt<T>(ag.x / n), cvt<T>(ag.y / n), cvt<T>(ag.z / n), zero);
        const V4<T> w = mk4(cvt<T>(aw.x / n), cvt<T>(aw.y / n), cvt<T>(aw.z / n), zero);
        fb.color_inprogress[p] = c;
        fb.lobe[0].cv_a[p] = g;
        fb.lobe[1].cv_a[p] = w;
        fb.raw_color[p] = c;
        fb.raw_gi_colored[p] = g;
        fb.raw_gi_white[p] = w;
    }
}
#endif // LPRT_REFSLAB_TU
LPRT_VAR_BEGIN
template <typename T, bool FIRST, bool COMPOSE, bool COUNT, bool WIDE>
__global__ void __launch_bounds__(256, AnyhitWaves<T>::value) k_trace_di(SceneDev<T> sc, FrameDev<T> fb0, ComposeArgs ca, Counters *counters) {
    const FrameDev<T> fb = at_sample(fb0, blockIdx.z);
    uint32_t p = tile_pixel(fb);
    const uint32_t npix = fb.win.npix();
    uint32_t sorted_mask = 0u;
    if (!FIRST) {
        // Only pixels whose GI ray hit something cast shadow rays in round 1 (41 % of the Cornell frame): the tile's pixels with
        // rays first, so that they share waves and the waves without work leave at once (as k_trace_gi does by octant).
        // Half precision 0.36 -> 0.33 ms, single 0.39 -> 0.37; ordering round 0 too, or either round by the number of
        // light slots, costs more in scattered pixels than the fuller waves return (atrium round 0: 3.17 -> 3.36 ms).
        // The light mask travels with the pixel: the thread that gets it starts the pixel's other reads at once instead of
        // fetching the mask a second time first (one dependent round trip to memory less in front of the traversal).
        sorted_mask = p < npix ? fb.lc_mask[p] : 0u;
        const uint32_t key = sorted_mask ? 0u : 1u;
        p = tile_pixel_of(fb, block_sort_order(key, 2u, sorted_mask));
    }
    Cnt<COUNT> cnt;
    if (p < npix) {
        const T zero = (T)0, one = (T)1;
        V4<T> *intensity = FIRST ? fb.intensity0 : fb.intensity1;
        const uint32_t mask = FIRST ? (uint32_t)fb.lc_mask[p] : sorted_mask;
        V4<T> cur = intensity[p];
        if (mask) {
            const V3<T> source = xyz(FIRST ? fb.g_position[p] : fb.sc_position[p]);
            const uint2 skip = fb.skips[p];
            V3<T> fc = zero3<T>(); // accumulate_di_light: final_color = {} ; += di_intensity[4p+i]
            // Each lane walks ITS OWN active slots in slot order (the j-th pass serves every lane's j-th light) instead of all
            // lanes stepping through slots 0..3 together: a pass is as full as the lanes' light COUNTS allow, not as the
            // agreement of their masks (atrium: shadow rays of round 0 6.66 -> 6.53 ms, 4.99 -> 4.94 in half).  The per-pixel
            // sum keeps its order; the slots a pixel does not use contributed +0, and x + (+0) == x for every x this sum can hold
            // (it starts at +0, so it is never -0).
            for (uint32_t m = mask; m; m &= m - 1u) {
                const uint32_t i = (uint32_t)__builtin_ctz(m);
                const V4<T> dm = fb.lc_dir_maxt[i * npix + p];
                const V3<T> mm = xyz(fb.lc_multiplier[i * npix + p]);
                RayT<T> ray{source, xyz(dm)};
                Isect<T> is = cast_ray<T, true, COUNT, WIDE>(sc, ray, skip.x, skip.y, cvt<T>(ray_moveforward_t<T>()), dm.w, cnt);
                fc = fc + (is.hit ? zero : one) * mm;
            }
            cur = mk4(cur.x + fc.x, cur.y + fc.y, cur.z + fc.z, zero);
            intensity[p] = cur;
        }
        if (COMPOSE)
            compose_tail<T, FIRST>(fb, fb0, ca, p, cur);
    }
    flush_counters<COUNT>(cnt, counters);
}
LPRT_VAR_END

// ---- TG + S2: trace_gi (cuda.hpp:2083-2141) followed, in the same thread, by shade #2 of the hit ---------------------
// The reference stores a PixelShaderInput record per pixel and launches shade<false,true> over it; nothing but the
// traversal result is live here when the traversal ends, so the record stays in registers.
// shade #2 of pixel p on what its GI ray found (`cast`: a GI ray was cast at all)
template <typename T> LPRT_DEV void shade_gi_hit(const SceneDev<T> &sc, const FrameDev<T> &fb, uint32_t p, bool cast, const Isect<T> &is) {
    const T zero = (T)0;
    ShadeIn<T> in; // type INVALID unless the ray says otherwise
    in.uv4 = mk4(zero, zero, zero, zero);
    in.live = false;
    in.sky_pixel = false;
    in.inobject_offset = 0u;
    in.objectid = 0u;
    if (cast) {
        if (is.hit) {
            HitAttr<T> h = interpolate_hit(sc, is);
            in.position = h.position;
            in.raw_normal = h.normal;
            in.raw_tangent = h.tangent;
            in.vertex_color = h.color;
            in.uv4 = mk4(h.uv0.x, h.uv0.y, h.uv1.x, h.uv1.y);
            const ObjConst<T> *oc = sc.objects + is.inscene_objid;
            in.material = sc.materials[oc->materialid];
            in.objectid = oc->objectid;
            in.inobject_offset = is.inobject_offset;
            in.live = true; // COMMON
        } else {
            in.sky_pixel = true; // SKYBOX
        }
    }
    shade_and_store<T, false, true>(sc, fb, p, 0u, in, false);
}

LPRT_VAR_BEGIN
template <typename T, bool COUNT>
__global__ void __launch_bounds__(256, ClosestWaves<T>::value) k_trace_gi(SceneDev<T> sc, FrameDev<T> fb0, Counters *counters) {
    const FrameDev<T> fb = at_sample(fb0, blockIdx.z);
    uint32_t p = tile_pixel(fb);
    uint32_t sorted_valid = 0u;
    {
        // flag and direction are requested together (a pixel without a GI ray reads 16 bytes it does not need) and the flag
        // travels with the pixel: two dependent round trips to memory less in front of the traversal
        uint32_t key = 8u; // no GI ray
        if (p < fb.win.npix()) {
            sorted_valid = fb.gi_valid[p];
            const V4<T> d = fb.gi_direction[p];
            if (sorted_valid)
                key = ((float)d.x < 0.f ? 1u : 0u) | ((float)d.y < 0.f ? 2u : 0u) | ((float)d.z < 0.f ? 4u : 0u);
        }
        p = tile_pixel_of(fb, block_sort_order(key, 9u, sorted_valid));
    }
    Cnt<COUNT> cnt;
    if (p < fb.win.npix()) {
        const bool cast = sorted_valid != 0u;
        Isect<T> is;
        is.hit = false;
        if (cast) {
            const uint2 skip = fb.skips[p];
            RayT<T> ray{xyz(fb.g_position[p]), xyz(fb.gi_direction[p])};
            is = cast_ray<T, false, COUNT>(sc, ray, skip.x, skip.y, cvt<T>(ray_moveforward_t<T>()), cvt<T>(1e5f), cnt);
        }
        shade_gi_hit(sc, fb, p, cast, is);
    }
    flush_counters<COUNT>(cnt, counters);
}
LPRT_VAR_END

static inline dim3 grid_for(uint32_t n) { return dim3((n + 255) / 256); }

// ---- ray streams (stream.hpp): persistent traversal kernels + the per-pixel stages that follow them ----------------------
#ifndef LPRT_STREAM_DI_WAVES
#define LPRT_STREAM_DI_WAVES LPRT_STREAM_WAVES // the any-hit stream (no hit record, no barycentrics) needs fewer registers
#endif
#ifndef LPRT_STREAM_WAVES
#define LPRT_STREAM_WAVES 4 // waves per SIMD the stream kernels are compiled for
#endif
// items are numbered so that 64 consecutive ones are the pixels of one 8x8 block (row-major blocks): a refill hands
// neighbouring pixels to a wave
template <typename T> LPRT_DEV uint32_t block_pixel(const FrameDev<T> &fb, uint32_t blk, uint32_t l) {
    const uint32_t n_bx = (fb.win.W + 7u) / 8u;
    const uint32_t x = (blk % n_bx) * 8u + (l & 7u), y = (blk / n_bx) * 8u + (l >> 3);
    return (x < fb.win.W && y < fb.win.rows()) ? y * fb.win.W + x : 0xffffffffu;
}
template <typename T> static inline uint32_t n_blocks8(const FrameDev<T> &fb) { return ((fb.win.W + 7u) / 8u) * ((fb.win.rows() + 7u) / 8u); }

template <typename T> struct GiSource { // trace_gi's rays (cuda.hpp:2083-2105): one item per pixel and sample slot
    FrameDev<T> fb0;
    uint32_t slots; // sample slots of the batch: the runs of 64 items of one 8x8 block are adjacent, one per slot (the
                    // samples of a pixel start at the same point: consecutive batches of a wave walk the same part of the scene)
    LPRT_DEV bool load(uint32_t item, RayT<T> &ray, T &tmin, T &tmax, uint32_t &skip_vao, uint32_t &skip_off) const {
        const uint32_t run = item >> 6, z = run % slots;
        const FrameDev<T> fb = at_sample(fb0, z);
        const uint32_t p = block_pixel(fb, run / slots, item & 63u);
        if (p == 0xffffffffu || !fb.gi_valid[p])
            return false;
        const uint2 skip = fb.skips[p];
        ray.o = xyz(fb.g_position[p]);
        ray.d = xyz(fb.gi_direction[p]);
        tmin = cvt<T>(ray_moveforward_t<T>());
        tmax = cvt<T>(1e5f);
        skip_vao = skip.x;
        skip_off = skip.y;
        return true;
    }
    LPRT_DEV void store(uint32_t item, const Isect<T> &is) const {
        const uint32_t run = item >> 6, z = run % slots;
        const FrameDev<T> fb = at_sample(fb0, z);
        const uint32_t p = block_pixel(fb, run / slots, item & 63u);
        GiHit<T> h;
        h.objid_hit = (is.hit ? 0x80000000u : 0u) | is.inscene_objid;
        h.inobject_offset = is.inobject_offset;
        h.bx = is.bary.x;
        h.by = is.bary.y;
        h.bz = is.bary.z;
        h.t = is.t;
        fb.gi_hit[p] = h;
    }
};
template <typename T, bool FIRST> struct DiSource { // trace_di_light's rays (cuda.hpp:2058-2081): one item per pixel, light slot and sample slot
    FrameDev<T> fb0;
    uint32_t n_slots; // min(n_lights, 4)
    uint32_t slots;   // sample slots of the batch.  Items of one 8x8 block: slots x n_slots runs of 64, sample slot outer, light
                      // slot inner (a run's rays aim at one light)
    LPRT_DEV FrameDev<T> where(uint32_t item, uint32_t &p, uint32_t &slot) const {
        const uint32_t run = item >> 6;
        slot = run % n_slots;
        const uint32_t r2 = run / n_slots, z = r2 % slots;
        const FrameDev<T> fb = at_sample(fb0, z);
        p = block_pixel(fb, r2 / slots, item & 63u);
        return fb;
    }
    LPRT_DEV bool load(uint32_t item, RayT<T> &ray, T &tmin, T &tmax, uint32_t &skip_vao, uint32_t &skip_off) const {
        uint32_t p, slot;
        const FrameDev<T> fb = where(item, p, slot);
        if (p == 0xffffffffu || !((fb.lc_mask[p] >> slot) & 1u))
            return false;
        const uint32_t npix = fb.win.npix();
        const V4<T> dm = fb.lc_dir_maxt[slot * npix + p];
        const uint2 skip = fb.skips[p];
        ray.o = xyz(FIRST ? fb.g_position[p] : fb.sc_position[p]);
        ray.d = xyz(dm);
        tmin = cvt<T>(ray_moveforward_t<T>());
        tmax = dm.w;
        skip_vao = skip.x;
        skip_off = skip.y;
        return true;
    }
    LPRT_DEV void store(uint32_t item, const Isect<T> &is) const {
        uint32_t p, slot;
        const FrameDev<T> fb = where(item, p, slot);
        fb.vis[slot * fb.win.npix() + p] = is.hit ? 1u : 0u;
    }
};
template <typename T> struct HostRaySource { // lprt_trace_rays
    const lprt_ray *rays;
    lprt_hit *hits;
    LPRT_DEV bool load(uint32_t item, RayT<T> &ray, T &tmin, T &tmax, uint32_t &skip_vao, uint32_t &skip_off) const {
        const lprt_ray in = rays[item];
        ray.o = mk3(cvt<T>(in.source[0]), cvt<T>(in.source[1]), cvt<T>(in.source[2]));
        ray.d = mk3(cvt<T>(in.direction[0]), cvt<T>(in.direction[1]), cvt<T>(in.direction[2]));
        tmin = cvt<T>(in.min_distance);
        tmax = cvt<T>(in.max_distance);
        skip_vao = in.skip_vao;
        skip_off = in.skip_inobject_offset;
        return true;
    }
    LPRT_DEV void store(uint32_t item, const Isect<T> &is) const {
        lprt_hit h;
        h.hit = is.hit ? 1u : 0u;
        h.inscene_objid = is.inscene_objid;
        h.inobject_offset = is.inobject_offset;
        h.t = (float)is.t;
        h.barycentric[0] = (float)is.bary.x;
        h.barycentric[1] = (float)is.bary.y;
        h.barycentric[2] = (float)is.bary.z;
        hits[item] = h;
    }
};


#ifndef LPRT_REFSLAB_TU
// ---- ray sort: the order a stream hands its rays out in -------------------------------------------------------------------
// A stream's items are numbered by pixel block, so 64 consecutive GI rays start next to each other but head wherever their
// BRDF samples sent them, and the shadow rays of round 1 start wherever those GI rays ended.  On the atrium that leaves 6 of
// 16 lanes active per vector cycle and an L2 hit rate of 0.85 (profiles/r02k_atrium_f16_sq_summary.json): the lanes of a wave
// walk different parts of a 12-MB hierarchy.  The sort puts the items that carry a ray into the order
//     key = [light slot (shadow rays)] | Morton code of the origin's cell on a 32^3 grid over the scene's bounds | direction octant
// by a counting sort over 2^18 bins -- three passes over the items: keys + histogram, scan of the bins, scatter -- with the
// atomics of a wave aggregated per distinct key.  The order WITHIN a bin is whatever the scatter's atomics make it; results do
// not depend on the order at all (every ray reports to its own item), so frames stay bit-identical with the sort on or off
// (tests/test_gpu_frame.py::test_launch_structure_never_changes_a_frame).  Items without a ray drop out: the stream is
// *n_sorted long.
// MEASURED (round 3, MI355X, atrium 1080p 8 spp fp16, gpurun_out/ab_sort_*.json): GI rays 7.93 -> 7.86 ms per frame -- nothing:
// their origins were already neighbours (8x8 pixel blocks) and an octant is half of the hemisphere they are drawn from;
// shadow rays of round 1 7.42 -> 6.61 ms (-11 %: these start wherever the GI rays ended).  The sort itself costs 1.6 / 4.9 ms
// per frame in this form (device-scope atomics), several times what it buys, so LPRT_OPT_RAY_SORT defaults to OFF; a sort
// worth having here would have to cost < 0.5 ms per 50 M items (LDS-privatised radix passes) for a net gain of ~2 %.
LPRT_DEV uint32_t spread5(uint32_t v) { // 5 bits -> every third bit
    v &= 31u;
    v = (v | (v << 8)) & 0x100fu;
    v = (v | (v << 4)) & 0x10c3u;
    v = (v | (v << 2)) & 0x1249u;
    return v;
}
template <typename T> LPRT_DEV uint32_t ray_cell(const SceneDev<T> &sc, V3<T> o) {
    const float lo[3] = {(float)sc.tlas_root[0], (float)sc.tlas_root[1], (float)sc.tlas_root[2]};
    const float hi[3] = {(float)sc.tlas_root[3], (float)sc.tlas_root[4], (float)sc.tlas_root[5]};
    const float p[3] = {(float)o.x, (float)o.y, (float)o.z};
    uint32_t q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float e = hi[k] - lo[k];
        float f = e > 0.f ? (p[k] - lo[k]) / e * 32.f : 0.f;
        f = f > 0.f ? f : 0.f; // NaN -> 0
        q[k] = f < 31.f ? (uint32_t)f : 31u;
    }
    return spread5(q[0]) | (spread5(q[1]) << 1) | (spread5(q[2]) << 2);
}
template <typename T> LPRT_DEV uint32_t ray_octant(V3<T> d) {
    return ((float)d.x < 0.f ? 1u : 0u) | ((float)d.y < 0.f ? 2u : 0u) | ((float)d.z < 0.f ? 4u : 0u);
}
// the stream item's key (0xffffffff: the item carries no ray); the item numbering is GiSource's / DiSource's
template <typename T, int WHICH> LPRT_DEV uint32_t ray_sort_key(const SceneDev<T> &sc, const FrameDev<T> &fb0, uint32_t n_slots, uint32_t slots, uint32_t item) {
    const uint32_t run = item >> 6, l = item & 63u;
    if (WHICH == 0) {
        const uint32_t z = run % slots;
        const FrameDev<T> fb = at_sample(fb0, z);
        const uint32_t p = block_pixel(fb, run / slots, l);
        if (p == 0xffffffffu || !fb.gi_valid[p])
            return 0xffffffffu;
        return (ray_cell(sc, xyz(fb.g_position[p])) << 3) | ray_octant(xyz(fb.gi_direction[p]));
    }
    const uint32_t slot = run % n_slots, r2 = run / n_slots, z = r2 % slots;
    const FrameDev<T> fb = at_sample(fb0, z);
    const uint32_t p = block_pixel(fb, r2 / slots, l);
    if (p == 0xffffffffu || !((fb.lc_mask[p] >> slot) & 1u))
        return 0xffffffffu;
    const V3<T> o = xyz(WHICH == 1 ? fb.g_position[p] : fb.sc_position[p]);
    return (slot << 16) | (ray_cell(sc, o) << 1);
}
// one atomic per DISTINCT key of the wave; returns the value before this lane's own contribution (base + rank among equals).
// (One plain atomic per lane measured far slower: GI rays 7.8 ms instead of 1.6 ms per frame -- neighbouring pixels hit the same
// bin, and device-scope atomics on one address serialise.)
LPRT_DEV uint32_t wave_bin_add(uint32_t *bins, uint32_t key, bool valid) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(valid);
    uint32_t out = 0;
    while (todo) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        const uint32_t k = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)leader);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(valid && key == k);
        uint32_t base = 0;
        if (lane == leader)
            base = atomicAdd(bins + k, (uint32_t)__builtin_popcountll(same));
        base = (uint32_t)__builtin_amdgcn_readlane((int)base, (int)leader);
        if (valid && key == k)
            out = base + (uint32_t)__builtin_popcountll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    return out;
}
template <typename T, int WHICH>
__global__ void __launch_bounds__(256) k_ray_keys(SceneDev<T> sc, FrameDev<T> fb, uint32_t n_slots, uint32_t slots, uint32_t n_items, uint32_t *keys, uint32_t *bins) {
    const uint32_t item = blockIdx.x * 256u + threadIdx.x;
    uint32_t key = 0xffffffffu;
    if (item < n_items) {
        key = ray_sort_key<T, WHICH>(sc, fb, n_slots, slots, item);
        keys[item] = key;
    }
    wave_bin_add(bins, key, key != 0xffffffffu);
}
// exclusive scan of the bins in place (one workgroup: 2^18 counters are 16 passes of 1024 threads x 16); the total goes to *n_sorted
__global__ void __launch_bounds__(1024) k_ray_bins_scan(uint32_t *bins, uint32_t n_bins, uint32_t *n_sorted) {
    __shared__ uint32_t wsum[16];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < n_bins; base += 1024u * 16u) {
        const uint32_t i0 = base + 16u * threadIdx.x;
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
            const uint4 q = *reinterpret_cast<const uint4 *>(bins + i0 + j); // n_bins is a multiple of 16384
            v[j] = q.x, v[j + 1] = q.y, v[j + 2] = q.z, v[j + 3] = q.w;
        }
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const uint32_t x = v[j];
            v[j] = sum;
            sum += x;
        }
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= (uint32_t)off)
                incl += up;
        }
        if (lane == 63u)
            wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0, tile = 0;
#pragma unroll
        for (uint32_t w = 0; w < 16u; w++) {
            const uint32_t x = wsum[w];
            before += w < wave ? x : 0u;
            tile += x;
        }
        const uint32_t excl = carry + before + incl - sum;
#pragma unroll
        for (int j = 0; j < 16; j += 4)
            *reinterpret_cast<uint4 *>(bins + i0 + j) = make_uint4(excl + v[j], excl + v[j + 1], excl + v[j + 2], excl + v[j + 3]);
        carry += tile;
        __syncthreads();
    }
    if (threadIdx.x == 0)
        *n_sorted = carry;
}
__global__ void __launch_bounds__(256) k_ray_scatter(const uint32_t *keys, uint32_t n_items, uint32_t *bins, uint32_t *order) {
    const uint32_t item = blockIdx.x * 256u + threadIdx.x;
    const uint32_t key = item < n_items ? keys[item] : 0xffffffffu;
    const bool valid = key != 0xffffffffu;
    const uint32_t pos = wave_bin_add(bins, key, valid);
    if (valid)
        order[pos] = item;
}
template <typename T> uint32_t ray_sort_items(const FrameDev<T> &fb, uint32_t n_lights, uint32_t slots, int which) {
    const uint32_t n_slots = std::min<uint32_t>(n_lights, LPRT_MAX_DIRECT_LIGHT);
    return n_blocks8(fb) * 64u * slots * (which == 0 ? 1u : n_slots);
}
template <typename T> void launch_ray_sort(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, uint32_t slots, int which, const RaySortBufs &b) {
    const uint32_t n_slots = std::min<uint32_t>(sc.n_lights, LPRT_MAX_DIRECT_LIGHT);
    const uint32_t n_items = ray_sort_items(fb, sc.n_lights, slots, which);
    (void)hipMemsetAsync(b.bins, 0, sizeof(uint32_t) * LPRT_RAYSORT_BINS, st);
    if (!n_items) {
        (void)hipMemsetAsync(b.n_sorted, 0, sizeof(uint32_t), st);
        return;
    }
    const dim3 g((n_items + 255u) / 256u), blk(256);
    if (which == 0)
        hipLaunchKernelGGL((k_ray_keys<T, 0>), g, blk, 0, st, sc, fb, n_slots, slots, n_items, b.keys, b.bins);
    else if (which == 1)
        hipLaunchKernelGGL((k_ray_keys<T, 1>), g, blk, 0, st, sc, fb, n_slots, slots, n_items, b.keys, b.bins);
    else
        hipLaunchKernelGGL((k_ray_keys<T, 2>), g, blk, 0, st, sc, fb, n_slots, slots, n_items, b.keys, b.bins);
    hipLaunchKernelGGL(k_ray_bins_scan, dim3(1), dim3(1024), 0, st, b.bins, LPRT_RAYSORT_BINS, b.n_sorted);
    hipLaunchKernelGGL(k_ray_scatter, g, blk, 0, st, b.keys, n_items, b.bins, b.order);
}
#endif // LPRT_REFSLAB_TU

LPRT_VAR_BEGIN
template <typename T, bool COUNT, bool WIDE>
__global__ void __launch_bounds__(256, LPRT_STREAM_WAVES) k_stream_gi(SceneDev<T> sc, FrameDev<T> fb, uint32_t slots, uint32_t n_items, StreamOrder so,
                                                                       uint32_t *counter, Counters *counters) {
    Cnt<COUNT> cnt;
    GiSource<T> src{fb, slots};
    trace_stream<T, false, COUNT, WIDE>(sc, src, n_items, so, counter, cnt);
    flush_counters<COUNT>(cnt, counters);
}
template <typename T, bool FIRST, bool COUNT, bool WIDE>
__global__ void __launch_bounds__(256, LPRT_STREAM_DI_WAVES) k_stream_di(SceneDev<T> sc, FrameDev<T> fb, uint32_t n_slots, uint32_t slots, uint32_t n_items,
                                                                       StreamOrder so, uint32_t *counter, Counters *counters) {
    Cnt<COUNT> cnt;
    DiSource<T, FIRST> src{fb, n_slots, slots};
    trace_stream<T, true, COUNT, WIDE>(sc, src, n_items, so, counter, cnt);
    flush_counters<COUNT>(cnt, counters);
}
template <typename T, bool ANY, bool COUNT, bool WIDE>
__global__ void __launch_bounds__(256, LPRT_STREAM_WAVES) k_stream_rays(SceneDev<T> sc, const lprt_ray *rays, uint32_t n, lprt_hit *hits, uint32_t *counter,
                                                                         Counters *counters) {
    Cnt<COUNT> cnt;
    HostRaySource<T> src{rays, hits};
    trace_stream<T, ANY, COUNT, WIDE>(sc, src, n, StreamOrder{nullptr, nullptr}, counter, cnt);
    flush_counters<COUNT>(cnt, counters);
}
LPRT_VAR_END

#ifndef LPRT_REFSLAB_TU
// shade #2 of every pixel on the hit record its GI ray left (the tail of k_trace_gi as a kernel of its own)
template <typename T> __global__ void __launch_bounds__(256) k_shade2(SceneDev<T> sc, FrameDev<T> fb0) {
    const FrameDev<T> fb = at_sample(fb0, blockIdx.z);
    const uint32_t p = tile_pixel(fb);
    if (p >= fb.win.npix())
        return;
    const bool cast = fb.gi_valid[p] != 0;
    Isect<T> is;
    is.hit = false;
    if (cast) {
        const GiHit<T> h = fb.gi_hit[p];
        is.hit = (h.objid_hit >> 31) != 0u;
        is.inscene_objid = h.objid_hit & 0x7fffffffu;
        is.inobject_offset = h.inobject_offset;
        is.bary = mk3(h.bx, h.by, h.bz);
        is.t = h.t;
    }
    shade_gi_hit(sc, fb, p, cast, is);
}
// accumulate_di_light over the visibility bytes the shadow-ray stream left (the tail of k_trace_di as a kernel of its own)
template <typename T, bool FIRST, bool COMPOSE>
__global__ void __launch_bounds__(256) k_accum_di(FrameDev<T> fb0, ComposeArgs ca) {
    const FrameDev<T> fb = at_sample(fb0, blockIdx.z);
    const uint32_t p = tile_pixel(fb);
    const uint32_t npix = fb.win.npix();
    if (p >= npix)
        return;
    const T zero = (T)0, one = (T)1;
    V4<T> *intensity = FIRST ? fb.intensity0 : fb.intensity1;
    const uint32_t mask = fb.lc_mask[p];
    V4<T> cur = intensity[p];
    if (mask) {
        V3<T> fc = zero3<T>(); // accumulate_di_light: final_color = {} ; += di_intensity[4p+i]
        for (uint32_t i = 0; i < LPRT_MAX_DIRECT_LIGHT; i++) {
            V3<T> di = zero3<T>();
            if (mask & (1u << i))
                di = (fb.vis[i * npix + p] ? zero : one) * xyz(fb.lc_multiplier[i * npix + p]);
            fc = fc + di;
        }
        cur = mk4(cur.x + fc.x, cur.y + fc.y, cur.z + fc.z, zero);
        intensity[p] = cur;
    }
    if (COMPOSE)
        compose_tail<T, FIRST>(fb, fb0, ca, p, cur);
}
#endif // LPRT_REFSLAB_TU

// persistent grid: as many workgroups as the chip holds at once (occupancy x CUs), never more than the stream has work for
// (the occupancy query is cached per (device, kernel): instantiations that differ only in their template arguments share a
// function-pointer TYPE, so the cache cannot be a static of this template alone)
template <typename K> static dim3 stream_grid(K kernel, uint32_t n_items) {
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, uint32_t> resident; // workgroups the device holds at once
    int dev = 0;
    (void)hipGetDevice(&dev);
    uint32_t have = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        const auto key = std::make_pair(dev, reinterpret_cast<const void *>(kernel));
        auto it = resident.find(key);
        if (it == resident.end()) {
            int per_cu = 0, cus = 0;
            (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, 0) != hipSuccess || per_cu <= 0)
                per_cu = LPRT_STREAM_WAVES;
            it = resident.emplace(key, (uint32_t)(per_cu * std::max(cus, 1))).first;
        }
        have = it->second;
    }
    const uint32_t want = (n_items + 255u) / 256u;
    return dim3(std::max(1u, std::min(want, have)));
}

// ---- launchers ---------------------------------------------------------------------------------------------------

LPRT_VAR_BEGIN
template <typename T> void launch_fill_gbuffer(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, Counters *c) {
    if (c)
        hipLaunchKernelGGL((k_fill_gbuffer<T, true>), tile_grid(fb), dim3(256), 0, st, sc, fb, c);
    else
        hipLaunchKernelGGL((k_fill_gbuffer<T, false>), tile_grid(fb), dim3(256), 0, st, sc, fb, c);
}
LPRT_VAR_END
#ifndef LPRT_REFSLAB_TU
template <typename T> void launch_temporal_map(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, bool first, uint32_t rng_d, bool floor_taps) {
    hipLaunchKernelGGL(k_temporal_map<T>, grid_for(fb.win.npix()), dim3(256), 0, st, sc, fb, first ? 1 : 0, rng_d, floor_taps ? 1 : 0);
}
#endif // LPRT_REFSLAB_TU
// tiles x sample slots of the batch
template <typename T> static inline dim3 tile_grid_z(const FrameDev<T> &fb, uint32_t slots) { return dim3(tile_grid(fb).x, 1, slots); }

#ifndef LPRT_REFSLAB_TU
template <typename T> void launch_shade(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, bool no_gi, const SampleBatch &sb) {
    const dim3 g = tile_grid_z(fb, sb.count), b(256);
    if (!no_gi)
        hipLaunchKernelGGL((k_shade<T, false>), g, b, 0, st, sc, fb, sb);
    else
        hipLaunchKernelGGL((k_shade<T, true>), g, b, 0, st, sc, fb, sb);
}
#endif // LPRT_REFSLAB_TU
LPRT_VAR_BEGIN
// wide: the BLAS set holds deep trees (GeometryStore::max_depth >= 6): the shadow rays walk them two levels per step
template <typename T, bool WIDE>
static void launch_trace_di_w(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, bool first, bool gi_on, bool add_direct, uint32_t slots,
                              uint32_t reuse, Counters *c) {
    const dim3 g = tile_grid_z(fb, slots), b(256);
    const ComposeArgs ca{gi_on ? 1 : 0, add_direct ? 1 : 0, reuse};
    if (first && gi_on) {
        if (c)
            hipLaunchKernelGGL((k_trace_di<T, true, false, true, WIDE>), g, b, 0, st, sc, fb, ca, c);
        else
            hipLaunchKernelGGL((k_trace_di<T, true, false, false, WIDE>), g, b, 0, st, sc, fb, ca, c);
    } else if (first) {
        if (c)
            hipLaunchKernelGGL((k_trace_di<T, true, true, true, WIDE>), g, b, 0, st, sc, fb, ca, c);
        else
            hipLaunchKernelGGL((k_trace_di<T, true, true, false, WIDE>), g, b, 0, st, sc, fb, ca, c);
    } else {
        if (c)
            hipLaunchKernelGGL((k_trace_di<T, false, true, true, WIDE>), g, b, 0, st, sc, fb, ca, c);
        else
            hipLaunchKernelGGL((k_trace_di<T, false, true, false, WIDE>), g, b, 0, st, sc, fb, ca, c);
    }
}
template <typename T>
void launch_trace_di(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, bool first, bool gi_on, bool add_direct, uint32_t slots, bool reuse,
                     Counters *c, bool wide) {
    if (wide)
        launch_trace_di_w<T, true>(st, sc, fb, first, gi_on, add_direct, slots, reuse ? 1u : 0u, c);
    else
        launch_trace_di_w<T, false>(st, sc, fb, first, gi_on, add_direct, slots, reuse ? 1u : 0u, c);
}
template <typename T> void launch_trace_gi(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, uint32_t slots, Counters *c) {
    if (c)
        hipLaunchKernelGGL((k_trace_gi<T, true>), tile_grid_z(fb, slots), dim3(256), 0, st, sc, fb, c);
    else
        hipLaunchKernelGGL((k_trace_gi<T, false>), tile_grid_z(fb, slots), dim3(256), 0, st, sc, fb, c);
}
LPRT_VAR_END
#ifndef LPRT_REFSLAB_TU
template <typename T> void launch_spp_mean(hipStream_t st, const FrameDev<T> &fb, uint32_t first, uint32_t count, uint32_t spp) {
    hipLaunchKernelGGL(k_spp_mean<T>, grid_for(fb.win.npix()), dim3(256), 0, st, fb, first, count, spp);
}
#endif // LPRT_REFSLAB_TU

LPRT_VAR_BEGIN
template <typename T, bool WIDE>
static void launch_stream_gi_w(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, uint32_t slots, uint32_t *counter, Counters *c,
                               const StreamOrder &so) {
    const uint32_t per_slot = n_blocks8(fb) * 64u, n_items = per_slot * slots;
    if (c)
        hipLaunchKernelGGL((k_stream_gi<T, true, WIDE>), stream_grid(k_stream_gi<T, true, WIDE>, n_items), dim3(256), 0, st, sc, fb, slots, n_items, so, counter, c);
    else
        hipLaunchKernelGGL((k_stream_gi<T, false, WIDE>), stream_grid(k_stream_gi<T, false, WIDE>, n_items), dim3(256), 0, st, sc, fb, slots, n_items, so, counter, c);
}
LPRT_VAR_END
#ifndef LPRT_REFSLAB_TU
template <typename T> void launch_shade2(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, uint32_t slots) {
    hipLaunchKernelGGL(k_shade2<T>, tile_grid_z(fb, slots), dim3(256), 0, st, sc, fb);
}
#endif // LPRT_REFSLAB_TU
LPRT_VAR_BEGIN
template <typename T>
void launch_stream_gi(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, uint32_t slots, uint32_t *counter, Counters *c, bool wide,
                      const uint32_t *order, const uint32_t *n_sorted) {
    const StreamOrder so{order, n_sorted};
    if (wide)
        launch_stream_gi_w<T, true>(st, sc, fb, slots, counter, c, so);
    else
        launch_stream_gi_w<T, false>(st, sc, fb, slots, counter, c, so);
}
template <typename T, bool FIRST, bool WIDE>
static void launch_stream_di_w(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, uint32_t slots, uint32_t *counter, Counters *c,
                               const StreamOrder &so) {
    const uint32_t n_slots = std::min<uint32_t>(sc.n_lights, LPRT_MAX_DIRECT_LIGHT);
    const uint32_t per_slot = n_blocks8(fb) * 64u * n_slots, n_items = per_slot * slots;
    if (!n_items)
        return;
    if (c)
        hipLaunchKernelGGL((k_stream_di<T, FIRST, true, WIDE>), stream_grid(k_stream_di<T, FIRST, true, WIDE>, n_items), dim3(256), 0, st, sc, fb, n_slots,
                           slots, n_items, so, counter, c);
    else
        hipLaunchKernelGGL((k_stream_di<T, FIRST, false, WIDE>), stream_grid(k_stream_di<T, FIRST, false, WIDE>, n_items), dim3(256), 0, st, sc, fb, n_slots,
                           slots, n_items, so, counter, c);
}
template <typename T>
void launch_stream_di(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, bool first, uint32_t slots, uint32_t *counter, Counters *c, bool wide,
                      const uint32_t *order, const uint32_t *n_sorted) {
    const StreamOrder so{order, n_sorted};
    if (first) {
        if (wide)
            launch_stream_di_w<T, true, true>(st, sc, fb, slots, counter, c, so);
        else
            launch_stream_di_w<T, true, false>(st, sc, fb, slots, counter, c, so);
    } else {
        if (wide)
            launch_stream_di_w<T, false, true>(st, sc, fb, slots, counter, c, so);
        else
            launch_stream_di_w<T, false, false>(st, sc, fb, slots, counter, c, so);
    }
}
LPRT_VAR_END
#ifndef LPRT_REFSLAB_TU
template <typename T>
void launch_accum_di(hipStream_t st, const FrameDev<T> &fb, bool first, bool gi_on, bool add_direct, uint32_t slots, bool reuse) {
    const dim3 g = tile_grid_z(fb, slots), b(256);
    const ComposeArgs ca{gi_on ? 1 : 0, add_direct ? 1 : 0, reuse ? 1u : 0u};
    if (first && gi_on)
        hipLaunchKernelGGL((k_accum_di<T, true, false>), g, b, 0, st, fb, ca);
    else if (first)
        hipLaunchKernelGGL((k_accum_di<T, true, true>), g, b, 0, st, fb, ca);
    else
        hipLaunchKernelGGL((k_accum_di<T, false, true>), g, b, 0, st, fb, ca);
}
#endif // LPRT_REFSLAB_TU
LPRT_VAR_BEGIN
template <typename T, bool ANY, bool WIDE>
static void launch_stream_rays_w(hipStream_t st, const SceneDev<T> &sc, const lprt_ray *rays, uint32_t n, lprt_hit *hits, uint32_t *counter, Counters *c) {
    if (c)
        hipLaunchKernelGGL((k_stream_rays<T, ANY, true, WIDE>), stream_grid(k_stream_rays<T, ANY, true, WIDE>, n), dim3(256), 0, st, sc, rays, n, hits, counter, c);
    else
        hipLaunchKernelGGL((k_stream_rays<T, ANY, false, WIDE>), stream_grid(k_stream_rays<T, ANY, false, WIDE>, n), dim3(256), 0, st, sc, rays, n, hits, counter, c);
}
template <typename T>
void launch_stream_rays(hipStream_t st, const SceneDev<T> &sc, const lprt_ray *rays, uint32_t n, bool any, bool wide, lprt_hit *hits, uint32_t *counter,
                        Counters *counters) {
    if (n == 0)
        return;
    if (any && wide)
        launch_stream_rays_w<T, true, true>(st, sc, rays, n, hits, counter, counters);
    else if (any)
        launch_stream_rays_w<T, true, false>(st, sc, rays, n, hits, counter, counters);
    else if (wide)
        launch_stream_rays_w<T, false, true>(st, sc, rays, n, hits, counter, counters);
    else
        launch_stream_rays_w<T, false, false>(st, sc, rays, n, hits, counter, counters);
}

#define INST_TRACE(T)                                                                                                  \
    template void launch_stream_gi<T>(hipStream_t, const SceneDev<T> &, const FrameDev<T> &, uint32_t, uint32_t *, Counters *, bool, const uint32_t *, \
                                      const uint32_t *);                                                                \
    template void launch_stream_di<T>(hipStream_t, const SceneDev<T> &, const FrameDev<T> &, bool, uint32_t, uint32_t *, Counters *, bool,     \
                                      const uint32_t *, const uint32_t *);                                             \
    template void launch_stream_rays<T>(hipStream_t, const SceneDev<T> &, const lprt_ray *, uint32_t, bool, bool, lprt_hit *, uint32_t *, Counters *); \
    template void launch_fill_gbuffer<T>(hipStream_t, const SceneDev<T> &, const FrameDev<T> &, Counters *);          \
    template void launch_trace_di<T>(hipStream_t, const SceneDev<T> &, const FrameDev<T> &, bool, bool, bool, uint32_t, bool, Counters *, bool); \
    template void launch_trace_gi<T>(hipStream_t, const SceneDev<T> &, const FrameDev<T> &, uint32_t, Counters *);
#ifndef LPRT_REFSLAB_TU
INST_TRACE(float)
#endif
INST_TRACE(h16)
#undef INST_TRACE
LPRT_VAR_END

#ifndef LPRT_REFSLAB_TU
#define INST(T)                                                                                                        \
    template void launch_shade2<T>(hipStream_t, const SceneDev<T> &, const FrameDev<T> &, uint32_t);                   \
    template uint32_t ray_sort_items<T>(const FrameDev<T> &, uint32_t, uint32_t, int);                                 \
    template void launch_ray_sort<T>(hipStream_t, const SceneDev<T> &, const FrameDev<T> &, uint32_t, int, const RaySortBufs &); \
    template void launch_accum_di<T>(hipStream_t, const FrameDev<T> &, bool, bool, bool, uint32_t, bool);              \
    template void launch_spp_mean<T>(hipStream_t, const FrameDev<T> &, uint32_t, uint32_t, uint32_t);                  \
    template void launch_temporal_map<T>(hipStream_t, const SceneDev<T> &, const FrameDev<T> &, bool, uint32_t, bool);      \
    template void launch_shade<T>(hipStream_t, const SceneDev<T> &, const FrameDev<T> &, bool, const SampleBatch &);
INST(float)
INST(h16)
#undef INST
#endif // LPRT_REFSLAB_TU

} // namespace lprt
