// TEST INFRASTRUCTURE — C entry points of the CPU oracle for ctypes (tests/, smoke(), bench.py's
// cpu_baseline leg only).  Not part of the product; see oracle/half.hpp header note.
#include "render.hpp"
#include <memory>
#include <new>

using namespace orc;

namespace {
struct Handle {
    int precision;
    std::unique_ptr<Renderer<float>> f32;
    std::unique_ptr<Renderer<half>> f16;
};
template <typename T> void read_vec3(const std::vector<Vec3<T>> &v, size_t off, size_t n, float *dst) {
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++)
            dst[3 * i + size_t(k)] = float(v[off + i][k]);
}
template <typename T> void read_vec4(const std::vector<Vec4<T>> &v, size_t n, float *dst) {
    for (size_t i = 0; i < n; i++)
        for (int k = 0; k < 4; k++)
            dst[4 * i + size_t(k)] = float(v[i][k]);
}
template <typename T> int read_buffer(Renderer<T> &r, int id, float *dst, size_t n_floats) {
    size_t N = size_t(r.W) * r.H;
    auto need = [&](size_t c) { return n_floats == N * c; };
    switch (id) {
    case LPRT_BUF_COLOR_INPROGRESS: if (!need(3)) return LPRT_ERR_INVALID; read_vec3(r.color_inprogress, 0, N, dst); return 0;
    case LPRT_BUF_MUL_GI_COLORED: if (!need(3)) return LPRT_ERR_INVALID; read_vec3(r.mul_gi_colored, 0, N, dst); return 0;
    case LPRT_BUF_MUL_GI_WHITE: if (!need(3)) return LPRT_ERR_INVALID; read_vec3(r.mul_gi_white, 0, N, dst); return 0;
    case LPRT_BUF_INTENSITY0: if (!need(3)) return LPRT_ERR_INVALID; read_vec3(r.intensity, 0, N, dst); return 0;
    case LPRT_BUF_INTENSITY1: if (!need(3)) return LPRT_ERR_INVALID; read_vec3(r.intensity, N, N, dst); return 0;
    case LPRT_BUF_ALBEDO: if (!need(3)) return LPRT_ERR_INVALID; read_vec3(r.albedo, 0, N, dst); return 0;
    case LPRT_BUF_RAW_GI_COLORED: if (!need(3)) return LPRT_ERR_INVALID; read_vec3(r.raw_gi_colored, 0, N, dst); return 0;
    case LPRT_BUF_RAW_GI_WHITE: if (!need(3)) return LPRT_ERR_INVALID; read_vec3(r.raw_gi_white, 0, N, dst); return 0;
    case LPRT_BUF_RAW_COLOR: if (!need(3)) return LPRT_ERR_INVALID; read_vec3(r.raw_color, 0, N, dst); return 0;
    case LPRT_BUF_OUTPUT: if (!need(4)) return LPRT_ERR_INVALID; read_vec4(r.output, N, dst); return 0;
    case LPRT_BUF_G_POSITION: if (!need(4)) return LPRT_ERR_INVALID; read_vec4(r.g_position, N, dst); return 0;
    case LPRT_BUF_G_NORMAL_DEPTH: if (!need(4)) return LPRT_ERR_INVALID; read_vec4(r.g_normal_depth, N, dst); return 0;
    case LPRT_BUF_G_TANGENT: if (!need(4)) return LPRT_ERR_INVALID; read_vec4(r.g_tangent, N, dst); return 0;
    case LPRT_BUF_G_UV0UV1: if (!need(4)) return LPRT_ERR_INVALID; read_vec4(r.g_uv, N, dst); return 0;
    case LPRT_BUF_G_COLOR:
        if (!need(4)) return LPRT_ERR_INVALID;
        for (size_t i = 0; i < N; i++)
            for (int k = 0; k < 4; k++)
                dst[4 * i + size_t(k)] = float(r.g_color[i][size_t(k)]);
        return 0;
    case LPRT_BUF_G_OBJID:
        if (!need(1)) return LPRT_ERR_INVALID;
        for (size_t i = 0; i < N; i++) dst[i] = float(r.g_objid[i]);
        return 0;
    case LPRT_BUF_G_OFFSET:
        if (!need(1)) return LPRT_ERR_INVALID;
        for (size_t i = 0; i < N; i++) dst[i] = float(r.g_offset[i]);
        return 0;
    // the commands the last shade of the frame left (rt/rtrt/cuda.hpp:1014-1032), for the ray replay of tests/replay.py
    case LPRT_BUF_GI_DIRECTION:
        if (!need(4)) return LPRT_ERR_INVALID;
        if (r.dbg_gi_direction.size() != N) return LPRT_ERR_INVALID; // no frame rendered yet
        read_vec4(r.dbg_gi_direction, N, dst);
        return 0;
    case LPRT_BUF_LIGHT_DIR_MAXT: // [4 slots][N][4]
        if (!need(16)) return LPRT_ERR_INVALID;
        for (size_t s = 0; s < 4; s++)
            for (size_t i = 0; i < N; i++) {
                const auto &c = r.trace_light_commands[i * 4 + s];
                const bool on = c.slot != UINT8_MAX;
                for (int k = 0; k < 3; k++)
                    dst[4 * (s * N + i) + size_t(k)] = on ? float(c.direction[k]) : 0.f;
                dst[4 * (s * N + i) + 3] = on ? float(c.maximum_t) : 0.f;
            }
        return 0;
    case LPRT_BUF_LIGHT_MASK:
        if (!need(1)) return LPRT_ERR_INVALID;
        for (size_t i = 0; i < N; i++) {
            unsigned m = 0;
            for (size_t s = 0; s < 4; s++)
                m |= (r.trace_light_commands[i * 4 + s].slot != UINT8_MAX ? 1u : 0u) << s;
            dst[i] = float(m);
        }
        return 0;
    case LPRT_BUF_SHADE_POSITION: // source of the round-1 shadow rays = the commands' own source
        if (!need(3)) return LPRT_ERR_INVALID;
        for (size_t i = 0; i < N; i++) {
            const decltype(r.trace_light_commands.data()) c0 = &r.trace_light_commands[i * 4];
            int first = -1;
            for (int s = 3; s >= 0; s--)
                if (c0[s].slot != UINT8_MAX)
                    first = s;
            for (int k = 0; k < 3; k++)
                dst[3 * i + size_t(k)] = first >= 0 ? float(c0[first].source[k]) : 0.f;
        }
        return 0;
    case LPRT_BUF_SKIP:
        if (!need(2)) return LPRT_ERR_INVALID;
        for (size_t i = 0; i < N; i++) {
            dst[2 * i] = float(r.skips[i].vao_index);
            dst[2 * i + 1] = float(r.skips[i].inobject_offset);
        }
        return 0;
    default: return LPRT_ERR_INVALID;
    }
}
template <typename T> void trace_rays(Renderer<T> &r, const lprt_ray *rays, uint32_t n, int find_any, lprt_hit *hits) {
    std::vector<Counters> local(size_t(r.threads > 0 ? r.threads : 1));
#pragma omp parallel for schedule(dynamic, 16) num_threads(r.threads)
    for (int64_t i = 0; i < int64_t(n); i++) {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        const lprt_ray &in = rays[i];
        Ray<T> ray;
        for (int k = 0; k < 3; k++) {
            ray.source[k] = T(in.source[k]);
            ray.direction[k] = T(in.direction[k]);
        }
        SkipGeometry skip{in.skip_vao, in.skip_inobject_offset};
        Intersection<T> is = find_any == 2 ? r.cast_ray_exhaustive(ray, skip, T(in.min_distance), T(in.max_distance), local[size_t(tid)])
                             : find_any  ? r.template cast_ray<true>(ray, skip, T(in.min_distance), T(in.max_distance), local[size_t(tid)])
                                         : r.template cast_ray<false>(ray, skip, T(in.min_distance), T(in.max_distance), local[size_t(tid)]);
        lprt_hit &h = hits[i];
        h.hit = is.hit ? 1u : 0u;
        h.inscene_objid = is.inscene_objid;
        h.inobject_offset = is.inobject_offset;
        h.t = float(is.t);
        for (int k = 0; k < 3; k++)
            h.barycentric[k] = float(is.barycentric[k]);
    }
    for (auto &c : local)
        r.counters.add(c);
}
} // namespace

#define DISPATCH(h, expr_f32, expr_f16) ((h)->precision == LPRT_F32 ? (expr_f32) : (expr_f16))

extern "C" {

void *orc_create(int precision, uint32_t width, uint32_t height) {
    if (width == 0 || height == 0 || width > 65535 || height > 65535)
        return nullptr;
    auto *h = new (std::nothrow) Handle();
    if (!h)
        return nullptr;
    h->precision = precision;
    if (precision == LPRT_F32)
        h->f32.reset(new Renderer<float>(width, height));
    else
        h->f16.reset(new Renderer<half>(width, height));
    return h;
}
void orc_destroy(void *p) { delete static_cast<Handle *>(p); }

int orc_set_threads(void *p, int n) {
    auto *h = static_cast<Handle *>(p);
    if (n < 1)
        n = 1;
    if (h->f32) h->f32->threads = n; else h->f16->threads = n;
    return 0;
}
// see Renderer::tie_order; before orc_upload_geometry
int orc_set_tie_order(void *p, int mode) {
    auto *h = static_cast<Handle *>(p);
    if (mode < 0 || mode > 2)
        return LPRT_ERR_INVALID;
    if (h->f32) h->f32->tie_order = mode; else h->f16->tie_order = mode;
    return 0;
}
int orc_upload_geometry(void *p, const lprt_vertex *vbo, uint32_t nv, const uint32_t *ebo, uint32_t ni,
                        const lprt_mesh_vao *vaos, uint32_t nvaos) {
    auto *h = static_cast<Handle *>(p);
    for (uint32_t i = 0; i < nvaos; i++)
        if (vaos[i].ebo_index || vaos[i].vbo_index || vaos[i].ebo_offset + vaos[i].ebo_length > ni || vaos[i].ebo_length % 3)
            return LPRT_ERR_INVALID;
    for (uint32_t i = 0; i < ni; i++)
        if (ebo[i] >= nv)
            return LPRT_ERR_INVALID;
    if (h->f32) h->f32->upload_geometry(vbo, nv, ebo, ni, vaos, nvaos); else h->f16->upload_geometry(vbo, nv, ebo, ni, vaos, nvaos);
    return 0;
}
int orc_upload_materials(void *p, const lprt_material *m, uint32_t n) {
    auto *h = static_cast<Handle *>(p);
    if (h->f32) h->f32->upload_materials(m, n); else h->f16->upload_materials(m, n);
    return 0;
}
int orc_set_scene(void *p, const lprt_object *obj, const lprt_aabb *aabbs, uint32_t n, const lprt_light *lights, uint32_t nl,
                  const lprt_camera *cam, const float *w2c, const lprt_skybox *sky) {
    auto *h = static_cast<Handle *>(p);
    if (h->f32) h->f32->set_scene(obj, aabbs, n, lights, nl, cam, w2c); else h->f16->set_scene(obj, aabbs, n, lights, nl, cam, w2c);
    if (h->f32) h->f32->set_skybox(sky); else h->f16->set_skybox(sky);
    return 0;
}
int orc_upload_texture(void *p, const uint8_t *rgba8, uint32_t w, uint32_t hgt, uint32_t srgb, uint32_t *handle) {
    auto *h = static_cast<Handle *>(p);
    if (!rgba8 || !w || !hgt || !handle)
        return LPRT_ERR_INVALID;
    *handle = h->f32 ? h->f32->upload_texture(rgba8, w, hgt, srgb != 0) : h->f16->upload_texture(rgba8, w, hgt, srgb != 0);
    return 0;
}
int orc_tex2d(void *p, uint32_t handle, const float *uv, uint32_t n, float *rgba) {
    auto *h = static_cast<Handle *>(p);
    for (uint32_t i = 0; i < n; i++) {
        auto c = h->f32 ? h->f32->tex2d(handle, uv[2 * i], uv[2 * i + 1]) : h->f16->tex2d(handle, uv[2 * i], uv[2 * i + 1]);
        for (int k = 0; k < 4; k++)
            rgba[4 * i + k] = c[k];
    }
    return 0;
}
int orc_render_frame(void *p, const lprt_frame_settings *fs) {
    auto *h = static_cast<Handle *>(p);
    if (h->f32) h->f32->render_frame(*fs); else h->f16->render_frame(*fs);
    return 0;
}
int orc_reset_history(void *p) {
    auto *h = static_cast<Handle *>(p);
    if (h->f32) h->f32->reset_history(); else h->f16->reset_history();
    return 0;
}
int orc_read_buffer(void *p, int id, float *dst, size_t n_floats) {
    auto *h = static_cast<Handle *>(p);
    return h->f32 ? read_buffer(*h->f32, id, dst, n_floats) : read_buffer(*h->f16, id, dst, n_floats);
}
int orc_trace_rays(void *p, const lprt_ray *rays, uint32_t n, int find_any, lprt_hit *hits) {
    auto *h = static_cast<Handle *>(p);
    if (h->f32) trace_rays(*h->f32, rays, n, find_any, hits); else trace_rays(*h->f16, rays, n, find_any, hits);
    return 0;
}
// include/lprt_fmath.h as the HOST evaluates it (op 0 logf, 1 sinf, 2 cosf, 3 tanf): tests compare accuracy against
// double-precision libm here and bit-equality against the device's evaluation (lprt_debug_fmath)
void orc_fmath(int op, const float *in, uint32_t n, float *out) {
    for (uint32_t i = 0; i < n; i++)
        out[i] = op == 0 ? lprt_logf(in[i]) : op == 1 ? lprt_sinf(in[i]) : op == 2 ? lprt_cosf(in[i]) : lprt_tanf(in[i]);
}
int orc_get_counters(void *p, lprt_counters *out) {
    auto *h = static_cast<Handle *>(p);
    const Counters &c = h->f32 ? h->f32->counters : h->f16->counters;
    out->rays = c.rays;
    out->tlas_node_visits = c.tlas_node_visits;
    out->aabbw_tests = c.aabbw_tests;
    out->object_visits = c.object_visits;
    out->blas_node_visits = c.blas_node_visits;
    out->tri_tests = c.tri_tests;
    out->fallback_tests = c.fallback_tests;
    for (int k = 0; k < 3; k++)
        out->wave_steps[k] = out->lane_steps[k] = 0; // GPU-side occupancy counters: no meaning on the CPU
    return 0;
}
int orc_reset_counters(void *p) {
    auto *h = static_cast<Handle *>(p);
    if (h->f32) h->f32->counters = Counters(); else h->f16->counters = Counters();
    return 0;
}
int orc_get_stage_seconds(void *p, double *out) {
    auto *h = static_cast<Handle *>(p);
    const double *s = h->f32 ? h->f32->stage_seconds : h->f16->stage_seconds;
    for (int i = 0; i < LPRT_STAGE_COUNT; i++)
        out[i] = s[i];
    return 0;
}
int orc_reset_stage_seconds(void *p) {
    auto *h = static_cast<Handle *>(p);
    double *s = h->f32 ? h->f32->stage_seconds : h->f16->stage_seconds;
    for (int i = 0; i < LPRT_STAGE_COUNT; i++)
        s[i] = 0;
    return 0;
}

// ---- unit-level probes used by the CPU tests ---------------------------------------------------
uint16_t orc_f32_to_f16(float f) { return f32_to_f16_bits(f); }
float orc_f16_to_f32(uint16_t h) { return f16_bits_to_f32(h); }
// op: 0 add, 1 sub, 2 mul, 3 div, 4 sqrt(a)
uint16_t orc_half_op(int op, uint16_t a, uint16_t b) {
    half x, y;
    x.bits = a;
    y.bits = b;
    half r;
    switch (op) {
    case 0: r = x + y; break;
    case 1: r = x - y; break;
    case 2: r = x * y; break;
    case 3: r = x / y; break;
    default: r = sqrt(x); break;
    }
    return r.bits;
}
void orc_material_brdf(int precision, float metallic, float roughness, const float *V, const float *L, const float *N, float *out2) {
    if (precision == LPRT_F32) {
        auto b = material_brdf<float>(metallic, roughness, mk3(V[0], V[1], V[2]), mk3(L[0], L[1], L[2]), mk3(N[0], N[1], N[2]));
        out2[0] = b.colored;
        out2[1] = b.white;
    } else {
        auto h3 = [](const float *p) { return mk3(half(p[0]), half(p[1]), half(p[2])); };
        auto b = material_brdf<half>(half(metallic), half(roughness), h3(V), h3(L), h3(N));
        out2[0] = float(b.colored);
        out2[1] = float(b.white);
    }
}
// first `n` raw outputs, uniforms or normals of curand_init(seed, subsequence, 0)
void orc_xorwow(uint64_t seed, uint64_t subsequence, int kind, uint32_t n, void *out) {
    XorwowState st = xorwow_init(seed, subsequence);
    for (uint32_t i = 0; i < n; i++) {
        if (kind == 0)
            static_cast<uint32_t *>(out)[i] = xorwow_next(st);
        else if (kind == 1)
            static_cast<float *>(out)[i] = xorwow_uniform(st);
        else
            static_cast<float *>(out)[i] = xorwow_normal(st);
    }
}
// state words v[0..4], d after curand_init(seed, subsequence, 0)
void orc_xorwow_state(uint64_t seed, uint64_t subsequence, uint32_t *out6) {
    XorwowState st = xorwow_init(seed, subsequence);
    for (int k = 0; k < 5; k++)
        out6[k] = st.v[k];
    out6[5] = st.d;
}
// column-major dump of M^(2^k) (k-th doubling of the 2^67-step matrix), 160*5 words
void orc_xorwow_matrix(int k, uint32_t *out800) {
    const XorwowMatrix &m = xorwow_skip().power(k);
    for (int i = 0; i < 160; i++)
        for (int w = 0; w < 5; w++)
            out800[i * 5 + w] = m.col[i][w];
}
// BLAS/TLAS read-back (oracle topology) for traversal-level tests
uint32_t orc_blas_node_count(void *p, uint32_t vao) {
    auto *h = static_cast<Handle *>(p);
    return uint32_t(h->f32 ? h->f32->blas[vao].nodes.size() : h->f16->blas[vao].nodes.size());
}
uint32_t orc_tlas_node_count(void *p) {
    auto *h = static_cast<Handle *>(p);
    return uint32_t(h->f32 ? h->f32->tlas.nodes.size() : h->f16->tlas.nodes.size());
}
}
