// TEST INFRASTRUCTURE — CPU oracle (see oracle/half.hpp header note).  PARITY UNPINNED against
// the CUDA path: the reference has no tests, fixtures or CPU path and cannot be built here
// (DESIGN.md "Oracle"); the only pins are the vendored half.hpp (oracle/ref_half_check.cpp) and
// the BRDF probe values recorded in SURVEY.md §8(c).
//
// Restatement of the reference's per-frame pipeline, templated on DataT T in {float, orc::half}.
// Function-level citations give the reference file:line each block follows.  Deviations from the
// reference are marked "DEVIATION Qn".
#ifndef ORC_RENDER_HPP
#define ORC_RENDER_HPP

#include "../include/lprt.h"
#include "core.hpp"
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace orc {

// ---- POD layouts (rt/rtrt/memory.hpp:26-51, 279-332, 382-405; rt/rtrt/cuda.hpp:1015-1065) ----
template <typename T> struct RDVertex {
    Vec3<T> position, normal, tangent, color;
    Vec2<T> uv0, uv1;
};
template <typename T> struct Material {
    Vec3<T> color, emission;
    T metallic, roughness;
    bool double_sided;
    uint32_t uv_color, tex_color;
};
// RDTexture (memory.hpp:334-369): RGBA8 pitch2D, cudaReadModeNormalizedFloat, cudaFilterModeLinear, normalised
// coordinates, address mode left at 0 = wrap, optional sRGB decode.
struct TextureData {
    std::vector<uint8_t> rgba;
    uint32_t width = 0, height = 0;
    bool srgb = false;
};
template <typename T> struct ObjectConstant {
    Mat4<T> transform_L2W, transform_W2L;
    uint32_t materialid, objectid;
};
template <typename T> struct Light {
    uint32_t type;
    Vec3<T> direction, intensity, position;
};
template <typename T> struct AABB {
    Vec3<T> lower_bound, upper_bound;
};
struct BVHNode { // rt/trace/object_bvh.hpp:20-41 == rt/trace/scene_bvh.hpp:22-43
    uint32_t a = UINT32_MAX; // lc | geometry offset
    uint32_t b = UINT32_MAX; // rc | geometry length
    uint32_t parent = UINT32_MAX;
    bool leaf = false;
};
template <typename T> struct BVH {
    std::vector<BVHNode> nodes;
    std::vector<AABB<T>> boxes;      // nodes[i].aabb
    std::vector<uint32_t> geometry;  // geometry_offset (BLAS) / inscene_objid (TLAS)
    std::vector<AABB<T>> aabbs_W;    // TLAS only
};

template <typename T> struct Ray {
    Vec3<T> source, direction;
    // cuda.hpp:42-47: direction is NOT renormalised
    Ray transformed(const Mat4<T> &m) const {
        Vec4<T> s4 = m * mk4(source[0], source[1], source[2], T(1.f));
        Vec4<T> d4 = m * mk4(direction[0], direction[1], direction[2], T(0.f));
        Ray r;
        r.source = mk3(s4[0] / s4[3], s4[1] / s4[3], s4[2] / s4[3]);
        r.direction = mk3(d4[0], d4[1], d4[2]);
        return r;
    }
};
template <typename T> struct Intersection { // cuda.hpp:50-57
    uint32_t inscene_objid = 0, inobject_offset = 0;
    Vec3<T> barycentric;
    T t = T(1e5f);
    bool hit = false;
};
struct SkipGeometry { // cuda.hpp:187-190
    uint32_t vao_index = UINT32_MAX;
    uint32_t inobject_offset = 0;
};
template <typename T> struct TraceGICommand { // cuda.hpp:1015-1020
    uint16_t x = 0, y = 0;
    Vec3<T> source, direction;
};
template <typename T> struct TraceLightCommand { // cuda.hpp:1023-1032
    uint16_t x = 0, y = 0;
    Vec3<T> source, direction;
    T maximum_t = T(0.f);
    uint8_t slot = 0;
    Vec3<T> material_multiplier;
};
template <typename T> struct PixelShaderInput { // cuda.hpp:1034-1048
    enum Type { INVALID, COMMON, SKYBOX } type = INVALID;
    Vec3<T> color, normal, position, tangent;
    Vec2<T> uv0, uv1;
    uint32_t materialid = 0, inscene_objid = 0, inobject_offset = 0;
};
template <typename T> struct TemporalMap { // cuda.hpp:1055-1065
    uint8_t frame_count = 0;
    uint16_t pos[4][2] = {};
    T weights[4] = {T(0.f), T(0.f), T(0.f), T(0.f)};
};
struct LastFramePixelID {
    uint32_t object_id = 0, inobject_offset = 0;
};

template <typename T> constexpr float ray_moveforward_t() { return sizeof(T) == 4 ? 1e-4f : 1e-1f; } // cuda.hpp:934-942

template <typename T> inline Vec3<T> lerp3(const Vec3<T> &b, const Vec3<T> &v1, const Vec3<T> &v2, const Vec3<T> &v3) {
    return b[0] * v1 + b[1] * v2 + b[2] * v3; // cuda.hpp:944-948
}
template <typename T> inline Vec2<T> lerp2(const Vec3<T> &b, const Vec2<T> &v1, const Vec2<T> &v2, const Vec2<T> &v3) {
    return b[0] * v1 + b[1] * v2 + b[2] * v3;
}
template <typename T> inline void value_or(Vec3<T> &io, const Vec3<T> &dflt) { // cuda.hpp:1168-1177
    for (int i = 0; i < 3; i++)
        if (!isfinite(io[i]))
            io[i] = dflt[i];
}
template <typename V> inline V clip_both(V lo, V hi, V val) { // cuda.hpp:881-888
    if (val < lo)
        return lo;
    else if (val > hi)
        return hi;
    else
        return val;
}
inline int f2i(float f) { // (int)float as CUDA/AMD hardware does it: NaN -> 0, saturating
    if (std::isnan(f))
        return 0;
    if (f >= 2147483648.f)
        return INT32_MAX;
    if (f <= -2147483648.f)
        return INT32_MIN;
    return int(f);
}

// ---- BRDFs (rt/rtrt/bsdf.hpp) -------------------------------------------------------------------
template <typename T> struct BRDF {
    T colored, white;
    Vec3<T> get_brdf(const Vec3<T> &base) const { // bsdf.hpp:52-54
        return base * colored + mk3(T(1.f), T(1.f), T(1.f)) * white;
    }
};
template <typename T> T specularBRDF(Vec3<T> V, Vec3<T> H, Vec3<T> L, Vec3<T> N, T alpha) { // bsdf.hpp:12-33
    T HdotL = dot_product(H, L);
    T HdotV = dot_product(H, V);
    T NdotH = dot_product(N, H);
    if (NdotH <= T(0.f) || HdotL <= T(0.f) || HdotV <= T(0.f)) {
        return T(0.f);
    } else {
        T absNdotL = abs(dot_product(N, L));
        T absNdotV = abs(dot_product(N, V));
        T a2 = alpha * alpha;
        T div1 = (absNdotL + sqrt(tmax<T>(T(0.f), (a2 + (T(1.f) - a2) * absNdotL * absNdotL))));
        T div2 = (absNdotV + sqrt(tmax<T>(T(0.f), (a2 + (T(1.f) - a2) * absNdotV * absNdotV))));
        T D_val = a2 / (T(float(M_PI)) * pow(NdotH * NdotH * (a2 - T(1.f)) + T(1.f), T(2.f)));
        return D_val / div1 / div2;
    }
}
template <typename T> BRDF<T> material_brdf(T metallic, T roughness, Vec3<T> V, Vec3<T> L, Vec3<T> N) { // bsdf.hpp:57-97
    if (dot_product(L, N) < T(0.f)) {
        return {T(0.f), T(0.f)};
    } else {
        Vec3<T> H = (L + V).normalized();
        T VdotH = dot_product(V, H);
        T pow5 = pow(T(tmax(T(0.f), T(1.f) - abs(VdotH))), T(5.f));
        T alpha = roughness * roughness;
        T layer = specularBRDF(V, H, L, N, alpha);
        T f0 = T(0.04f); // `DataT dielectric_f0 = 0.04` (double literal -> float -> T)
        T fr = f0 + T(1.f - float(f0)) * pow5; // `DataT(1 - dielectric_f0)`: int - T is formed in float
        T dielectric_white = fr * layer;
        T dielectric_colored = (T(1.f) - fr) * T(float(1 / M_PI));
        T metal_white = layer * pow5;
        T metal_colored = layer * (T(1.f) - pow5);
        return {metal_colored * metallic + dielectric_colored * (T(1.f) - metallic),
                metal_white * metallic + dielectric_white * (T(1.f) - metallic)};
    }
}
template <typename T> BRDF<T> glassy_brdf(T metallic, Vec3<T> V, Vec3<T> L, Vec3<T> N) { // bsdf.hpp:106-139
    if (dot_product(L, N) < T(0.f)) {
        return {T(0.f), T(0.f)};
    } else {
        Vec3<T> H = (L + V).normalized();
        T VdotH = dot_product(V, H);
        T pow5 = pow(T(tmax(T(0.f), T(1.f) - abs(VdotH))), T(5.f));
        T f0 = T(0.04f);
        T fr = f0 + T(1.f - float(f0)) * pow5;
        T dielectric_white = fr;
        T dielectric_colored = T(0.f);
        T metal_white = pow5;
        T metal_colored = (T(1.f) - pow5);
        return {metal_colored * metallic + dielectric_colored * (T(1.f) - metallic),
                metal_white * metallic + dielectric_white * (T(1.f) - metallic)};
    }
}

// ---- sampling helpers (cuda.hpp:1713-1749) ------------------------------------------------------
template <typename T> Vec3<T> sample_ggx(T a2, XorwowState &st) { // cuda.hpp:1714-1723 (Q9: x,y are not cos/sin phi)
    T rand_x = T(xorwow_uniform(st));
    T rand_y = T(xorwow_uniform(st));
    T rand_z = T(xorwow_uniform(st));
    T z = sqrt((T(1.f) - rand_x) / (T(1.f) + a2 * rand_x - rand_x));
    T r = sqrt(tmax(T(0.f), T(1.f) - z * z));
    T x = r * rand_y;
    T y = r * rand_z;
    return mk3(x, y, z);
}
template <typename T> T D_ggx_opti(T NH, T a2) { // cuda.hpp:1726-1729: `M_PI * tmp * tmp` is double * T
    T tmp = (NH * a2 - NH) * NH + T(1.0f);
    return T(float(M_PI * double(float(tmp)) * double(float(tmp))));
}
template <typename T> T pdf_ggx_reflect(T NH, T a2) { return NH * a2 / D_ggx_opti(NH, a2); } // cuda.hpp:1732-1734

struct Counters {
    uint64_t rays = 0, tlas_node_visits = 0, aabbw_tests = 0, object_visits = 0, blas_node_visits = 0,
             tri_tests = 0, fallback_tests = 0;
    void add(const Counters &o) {
        rays += o.rays;
        tlas_node_visits += o.tlas_node_visits;
        aabbw_tests += o.aabbw_tests;
        object_visits += o.object_visits;
        blas_node_visits += o.blas_node_visits;
        tri_tests += o.tri_tests;
        fallback_tests += o.fallback_tests;
    }
};

// =================================================================================================
template <typename T> class Renderer {
  public:
    uint32_t W = 0, H = 0;
    int threads = 1;
    // Which of two triangles with EQUAL split keys goes left (object_bvh.hpp:122-125 leaves that to std::nth_element, whose
    // arrangement among equal keys is implementation-defined -- the reference was built with MSVC's).  0: whatever
    // libstdc++'s nth_element does; 1: the lower primitive index first; 2: the higher index first.  1 and 2 are two LEGAL
    // reference topologies; comparing them measures how much of a half-precision frame the topology alone decides
    // (tests: the oracle-vs-oracle baseline of the fp16 parity bounds).  Must be set before upload_geometry.
    int tie_order = 0;

    // ---- resources (RDResource, memory.hpp:568-648) ----
    std::vector<RDVertex<T>> vbo;
    std::vector<uint32_t> ebo;
    std::vector<lprt_mesh_vao> vaos;
    std::vector<Mat3<T>> M_shift;
    std::vector<Mat3<float>> M_shift_f32;
    std::vector<Vec3<float>> position_f32;
    std::vector<Vec3<float>> positions_src; // fp32 positions for the BLAS build
    std::vector<Material<T>> materials;
    std::vector<BVH<T>> blas;
    // ---- per-frame scene (RDScene, memory.hpp:694-708) ----
    std::vector<ObjectConstant<T>> objects;
    std::vector<Light<T>> lights;
    BVH<T> tlas;
    Mat4<T> cam_L2W;
    T cam_fovy = T(0.f);
    Mat4<T> transform_W2C;
    // ---- RTRTProcedureBuffer (cuda.hpp:1475-1634) + GBuffer ----
    std::vector<Vec4<T>> g_position, g_normal_depth, g_tangent, g_uv;
    std::vector<std::array<uint8_t, 4>> g_color;
    std::vector<uint32_t> g_objid, g_offset;
    std::vector<Vec3<T>> gi_material_multiplier, intensity, view_direction, di_intensity, albedo, mul_gi_colored,
        mul_gi_white, color_inprogress, taa_history_color;
    std::vector<SkipGeometry> skips;
    std::vector<PixelShaderInput<T>> shade_commands;
    std::vector<TraceLightCommand<T>> trace_light_commands;
    std::vector<TraceGICommand<T>> trace_gi_commands;
    std::vector<uint8_t> gi_valid; // DEVIATION Q13
    std::vector<TemporalMap<T>> svgf_temporal_map, taa_temporal_map;
    std::vector<uint8_t> valid_frame_count;
    std::vector<Mat4<T>> last_frame_L2W;
    std::vector<LastFramePixelID> last_frame_id;
    Mat4<T> last_frame_transform_W2C;
    struct SVGF { // cuda.hpp:1117-1166
        std::vector<T> miu1_1, miu2_1, miu1_2, miu2_2, var_1, var_2, depth, illum;
        std::vector<Vec3<T>> color_history, color_buffer, normal;
        std::vector<Vec2<T>> depth_gradient;
        void resize(size_t n) {
            for (auto *v : {&miu1_1, &miu2_1, &miu1_2, &miu2_2, &var_1, &var_2, &depth, &illum})
                v->assign(n, T(0.f));
            color_history.assign(n, Vec3<T>());
            color_buffer.assign(n, Vec3<T>());
            normal.assign(n, Vec3<T>());
            depth_gradient.assign(n, Vec2<T>());
        }
    } svgf_colored, svgf_white;
    std::vector<Vec3<T>> raw_gi_colored, raw_gi_white, raw_color; // read-back copies
    std::vector<Vec4<T>> dbg_gi_direction; // read-back copy of the last sample's TraceGICommand directions (w = 1: cast)
    std::vector<Vec4<float>> output;
    uint64_t frame_index = 0;
    Counters counters;
    double stage_seconds[LPRT_STAGE_COUNT] = {};

    Renderer(uint32_t w, uint32_t h) : W(w), H(h) {
        init_srgb_lut();
        size_t n = size_t(w) * h;
        g_position.assign(n, Vec4<T>());
        g_normal_depth.assign(n, Vec4<T>());
        g_tangent.assign(n, Vec4<T>());
        g_uv.assign(n, Vec4<T>());
        g_color.assign(n, {0, 0, 0, 0});
        g_objid.assign(n, 0);
        g_offset.assign(n, 0);
        gi_material_multiplier.assign(n, Vec3<T>());
        intensity.assign(2 * n, Vec3<T>());
        view_direction.assign(n, Vec3<T>());
        di_intensity.assign(4 * n, Vec3<T>());
        albedo.assign(n, Vec3<T>());
        mul_gi_colored.assign(n, Vec3<T>());
        mul_gi_white.assign(n, Vec3<T>());
        color_inprogress.assign(n, Vec3<T>());
        taa_history_color.assign(n, Vec3<T>());
        skips.assign(n, SkipGeometry());
        shade_commands.assign(n, PixelShaderInput<T>());
        trace_light_commands.assign(4 * n, TraceLightCommand<T>());
        trace_gi_commands.assign(n, TraceGICommand<T>());
        gi_valid.assign(n, 0);
        svgf_temporal_map.assign(n, TemporalMap<T>());
        taa_temporal_map.assign(n, TemporalMap<T>());
        valid_frame_count.assign(n, 0);
        last_frame_id.assign(n, LastFramePixelID());
        svgf_colored.resize(n);
        svgf_white.resize(n);
        raw_gi_colored.assign(n, Vec3<T>());
        raw_gi_white.assign(n, Vec3<T>());
        raw_color.assign(n, Vec3<T>());
        output.assign(n, Vec4<float>());
    }

    // ------------------------------------------------------------------------------------------
    // B3: RDResource::compute_m (memory.hpp:621-648) + vertex conversion (memory.hpp:35-40)
    // ------------------------------------------------------------------------------------------
    void upload_geometry(const lprt_vertex *v, uint32_t nv, const uint32_t *idx, uint32_t ni, const lprt_mesh_vao *va,
                         uint32_t nva) {
        vbo.resize(nv);
        positions_src.resize(nv);
        for (uint32_t i = 0; i < nv; i++) {
            for (int k = 0; k < 3; k++) {
                vbo[i].position[k] = T(v[i].position[k]);
                vbo[i].normal[k] = T(v[i].normal[k]);
                vbo[i].tangent[k] = T(v[i].tangent[k]);
                vbo[i].color[k] = T(v[i].color[k]);
                positions_src[i][k] = v[i].position[k];
            }
            for (int k = 0; k < 2; k++) {
                vbo[i].uv0[k] = T(v[i].uv0[k]);
                vbo[i].uv1[k] = T(v[i].uv1[k]);
            }
        }
        ebo.assign(idx, idx + ni);
        vaos.assign(va, va + nva);
        M_shift.assign(ni / 3, Mat3<T>());
        M_shift_f32.assign(ni / 3, Mat3<float>());
        position_f32.assign(nv, Vec3<float>());
        for (uint32_t i = 0; i + 2 < ni; i += 3) {
            Vec3<float> v0 = positions_src[idx[i]], v1 = positions_src[idx[i + 1]], v2 = positions_src[idx[i + 2]];
            position_f32[idx[i]] = v0;
            position_f32[idx[i + 1]] = v1;
            position_f32[idx[i + 2]] = v2;
            Vec3<float> col0 = v0 - v2;
            Vec3<float> col1 = v1 - v2;
            Vec3<float> col2 = cross_product_difference(v0 - v2, v1 - v2) - v2; // Q1: the "- v2" is kept
            Mat3<float> M1;
            for (int r = 0; r < 3; r++) {
                M1.d[r][0] = col0[r];
                M1.d[r][1] = col1[r];
                M1.d[r][2] = col2[r];
            }
            Mat3<float> M2 = inversed_3_3(M1);
            M_shift_f32[i / 3] = M2;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++)
                    M_shift[i / 3].d[r][c] = T(M2.d[r][c]);
        }
        // B1: one BLAS per primitive from primitive-local positions/indices (loader.cu:330-336, 434)
        blas.assign(nva, BVH<T>());
        for (uint32_t p = 0; p < nva; p++)
            build_object_bvh(p);
    }

    void upload_materials(const lprt_material *m, uint32_t n) {
        materials.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            for (int k = 0; k < 3; k++) {
                materials[i].color[k] = T(m[i].color[k]);
                materials[i].emission[k] = T(m[i].emission[k]);
            }
            materials[i].metallic = T(m[i].metallic);
            materials[i].roughness = T(m[i].roughness);
            materials[i].double_sided = m[i].double_sided != 0;
            materials[i].uv_color = m[i].uv_color;
            materials[i].tex_color = m[i].tex_color;
        }
    }
    // ---- textures: the documented behaviour of the texture unit the reference samples through ----------------
    // (CUDA C Programming Guide, "Texture Fetching": wrap -> frac(u)*N, linear filtering at xB = x - 0.5 with the
    // blend weights held in 1.8 fixed point; sRGB textures are decoded per texel before filtering, alpha is not.)
    // PARITY UNPINNED: no reference image to pin the hardware's sRGB table or weight rounding against.
    std::vector<TextureData> textures;
    float srgb_lut[256];
    void init_srgb_lut() {
        for (int i = 0; i < 256; i++) {
            double c = i / 255.0;
            srgb_lut[i] = float(c <= 0.04045 ? c / 12.92 : std::pow((c + 0.055) / 1.055, 2.4));
        }
    }
    uint32_t upload_texture(const uint8_t *rgba8, uint32_t w, uint32_t h, bool srgb) {
        TextureData t;
        t.rgba.assign(rgba8, rgba8 + size_t(w) * h * 4);
        t.width = w;
        t.height = h;
        t.srgb = srgb;
        textures.push_back(std::move(t));
        return uint32_t(textures.size() - 1);
    }
    std::array<float, 4> texel(const TextureData &t, int i, int j) const {
        const uint8_t *p = &t.rgba[(size_t(j) * t.width + size_t(i)) * 4];
        if (t.srgb)
            return {srgb_lut[p[0]], srgb_lut[p[1]], srgb_lut[p[2]], p[3] / 255.f};
        return {p[0] / 255.f, p[1] / 255.f, p[2] / 255.f, p[3] / 255.f};
    }
    static void wrap_coord(float u, uint32_t n, int &i0, int &i1, float &alpha) {
        float f = u - std::floor(u);
        if (!(f == f) || !(f < 1.f)) // NaN / inf coordinates sample texel 0 (defined here; undefined in the reference)
            f = 0.f;
        float xb = f * float(n) - 0.5f;
        float fl = std::floor(xb);
        alpha = std::nearbyint((xb - fl) * 256.f) * (1.f / 256.f);
        i0 = int(fl);
        i1 = i0 + 1;
        if (i0 < 0)
            i0 += int(n);
        if (i1 >= int(n))
            i1 -= int(n);
    }
    std::array<float, 4> tex2d(uint32_t handle, float u, float v) const {
        if (handle >= textures.size())
            return {0.f, 0.f, 0.f, 0.f};
        const TextureData &t = textures[handle];
        int i0, i1, j0, j1;
        float a, b;
        wrap_coord(u, t.width, i0, i1, a);
        wrap_coord(v, t.height, j0, j1, b);
        auto t00 = texel(t, i0, j0), t10 = texel(t, i1, j0), t01 = texel(t, i0, j1), t11 = texel(t, i1, j1);
        std::array<float, 4> o;
        for (int k = 0; k < 4; k++)
            o[k] = (1.f - a) * (1.f - b) * t00[k] + a * (1.f - b) * t10[k] + (1.f - a) * b * t01[k] + a * b * t11[k];
        return o;
    }
    // RDSkybox<DataT>::Ref (memory.hpp:671-690) and direction_to_spherical (cuda.hpp:1742-1749)
    bool sky_valid = false;
    uint32_t sky_texture = 0;
    T sky_dx = T(0.f), sky_dy = T(0.f), sky_exposure = T(1.f);
    void set_skybox(const lprt_skybox *s) {
        sky_valid = s && s->valid && s->texture < textures.size();
        if (sky_valid) {
            sky_texture = s->texture;
            sky_dx = T(s->delta_x);
            sky_dy = T(s->delta_y);
            sky_exposure = T(s->exposure);
        }
    }
    Vec3<T> sample_skybox(const Vec3<T> &dir) const {
        float u = 0.1591f * std::atan2(float(dir[1]), float(dir[0])) + 0.5f + float(sky_dx);
        float v = 0.3183f * std::asin(float(dir[2])) + 0.5f + float(sky_dy);
        u = std::fmod(u, 1.f);
        v = 1.f - std::fmod(v, 1.f);
        auto c = tex2d(sky_texture, u, v);
        return mk3(T(c[0]), T(c[1]), T(c[2])) * sky_exposure;
    }

    // ------------------------------------------------------------------------------------------
    // B1: ObjectBVH (object_bvh.hpp:49-134).  std::nth_element is libstdc++'s here (the reference
    // was built with MSVC's); the partition it leaves is implementation-defined, so the exact
    // topology among equal keys is unpinned.
    // ------------------------------------------------------------------------------------------
    void build_object_bvh(uint32_t p) {
        const lprt_mesh_vao &vao = vaos[p];
        std::vector<uint32_t> geometries(vao.ebo_length / 3);
        for (size_t i = 0; i < geometries.size(); i++)
            geometries[i] = uint32_t(i * 3);
        BVH<T> &out = blas[p];
        out = BVH<T>();
        const uint32_t *indices = ebo.data() + vao.ebo_offset; // global vertex ids; same positions as the local ones
        build_object_rec(out, geometries.begin(), geometries.end(), indices, UINT32_MAX);
    }
    uint32_t build_object_rec(BVH<T> &out, std::vector<uint32_t>::iterator begin, std::vector<uint32_t>::iterator end,
                              const uint32_t *indices, uint32_t parent) {
        size_t size = size_t(end - begin);
        if (size == 0)
            return UINT32_MAX;
        Vec3<float> mn = positions_src[indices[*begin]], mx = mn;
        for (auto it = begin; it < end; it++)
            for (int j = 0; j < 3; j++) {
                const Vec3<float> &q = positions_src[indices[j + *it]];
                for (int k = 0; k < 3; k++) {
                    mn[k] = tmin(mn[k], q[k]);
                    mx[k] = tmax(mx[k], q[k]);
                }
            }
        out.nodes.push_back({});
        out.boxes.push_back({});
        uint32_t index = uint32_t(out.nodes.size() - 1);
        out.nodes[index].parent = parent;
        for (int k = 0; k < 3; k++) { // `.as<DataT>()`: round to nearest, not outward (Q2)
            out.boxes[index].lower_bound[k] = T(mn[k]);
            out.boxes[index].upper_bound[k] = T(mx[k]);
        }
        if (size <= 1) {
            out.nodes[index].a = uint32_t(out.geometry.size());
            for (auto it = begin; it < end; ++it)
                out.geometry.push_back(*it);
            out.nodes[index].b = uint32_t(size);
            out.nodes[index].leaf = true;
        } else {
            Vec3<float> width = mx - mn;
            size_t axis = 0;
            if (width[1] > width[0] && width[1] > width[2])
                axis = 1;
            else if (width[2] > width[0] && width[2] > width[1])
                axis = 2;
            size_t split = size / 2;
            const int tie = tie_order;
            std::nth_element(begin, begin + split, end, [&](uint32_t x, uint32_t y) {
                const float kx = positions_src[indices[x]][int(axis)], ky = positions_src[indices[y]][int(axis)];
                if (tie == 0 || kx != ky)
                    return kx < ky;
                return tie == 1 ? x < y : x > y;
            });
            uint32_t lc = build_object_rec(out, begin, begin + split, indices, index);
            out.nodes[index].a = lc;
            uint32_t rc = build_object_rec(out, begin + split, end, indices, index);
            out.nodes[index].b = rc;
            out.nodes[index].leaf = false;
        }
        return index;
    }

    // ------------------------------------------------------------------------------------------
    // B2: SceneBVH::update (scene_bvh.hpp:50-133) + per-frame uploads (cuda.hpp:662-710)
    // ------------------------------------------------------------------------------------------
    void set_scene(const lprt_object *obj, const lprt_aabb *aabbs, uint32_t n, const lprt_light *lg, uint32_t nl,
                   const lprt_camera *cam, const float *w2c) {
        objects.resize(n);
        for (uint32_t i = 0; i < n; i++) {
            objects[i].transform_L2W = mat4_from<T>(obj[i].transform_L2W);
            objects[i].transform_W2L = mat4_from<T>(obj[i].transform_W2L);
            objects[i].materialid = obj[i].materialid;
            objects[i].objectid = obj[i].objectid;
        }
        lights.resize(nl);
        for (uint32_t i = 0; i < nl; i++) {
            lights[i].type = lg[i].type;
            for (int k = 0; k < 3; k++) {
                lights[i].direction[k] = T(lg[i].direction[k]);
                lights[i].intensity[k] = T(lg[i].intensity[k]);
                lights[i].position[k] = T(lg[i].position[k]);
            }
        }
        cam_L2W = mat4_from<T>(cam->transform_L2W);
        cam_fovy = T(cam->field_of_view_y);
        transform_W2C = mat4_from<T>(w2c);
        if (last_frame_L2W.size() < vaos.size())
            last_frame_L2W.assign(vaos.size(), Mat4<T>());
        std::vector<uint32_t> order(n);
        for (uint32_t i = 0; i < n; i++)
            order[i] = i;
        tlas = BVH<T>();
        build_scene_rec(aabbs, order.begin(), order.end(), UINT32_MAX);
        tlas.aabbs_W.resize(n);
        for (uint32_t i = 0; i < n; i++) { // aabbs in the permuted order left by nth_element (scene_bvh.hpp:60-65)
            uint32_t id = order[i];
            for (int k = 0; k < 3; k++) {
                tlas.aabbs_W[i].lower_bound[k] = T(aabbs[id].lower[k]);
                tlas.aabbs_W[i].upper_bound[k] = T(aabbs[id].upper[k]);
            }
        }
    }
    uint32_t build_scene_rec(const lprt_aabb *aabbs, std::vector<uint32_t>::iterator begin,
                             std::vector<uint32_t>::iterator end, uint32_t parent) {
        size_t size = size_t(end - begin);
        if (size == 0)
            return UINT32_MAX;
        float mn[3], mx[3];
        for (int k = 0; k < 3; k++) {
            mn[k] = aabbs[*begin].lower[k];
            mx[k] = aabbs[*begin].upper[k];
        }
        for (auto it = begin + 1; it < end; it++)
            for (int k = 0; k < 3; k++) {
                mn[k] = std::min(mn[k], aabbs[*it].lower[k]);
                mx[k] = std::max(mx[k], aabbs[*it].upper[k]);
            }
        tlas.nodes.push_back({});
        tlas.boxes.push_back({});
        uint32_t index = uint32_t(tlas.nodes.size() - 1);
        tlas.nodes[index].parent = parent;
        for (int k = 0; k < 3; k++) {
            tlas.boxes[index].lower_bound[k] = T(mn[k]);
            tlas.boxes[index].upper_bound[k] = T(mx[k]);
        }
        if (size <= 1) {
            tlas.nodes[index].leaf = true;
            tlas.nodes[index].a = uint32_t(tlas.geometry.size());
            for (auto it = begin; it < end; it++)
                tlas.geometry.push_back(*it);
            tlas.nodes[index].b = uint32_t(size);
        } else {
            float width[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
            int axis = 0;
            if (width[1] > width[0] && width[1] > width[2])
                axis = 1;
            else if (width[2] > width[0] && width[2] > width[1])
                axis = 2;
            size_t split = size / 2;
            std::nth_element(begin, begin + split, end,
                             [&](uint32_t x, uint32_t y) { return aabbs[x].lower[axis] < aabbs[y].lower[axis]; });
            uint32_t lc = build_scene_rec(aabbs, begin, begin + split, index);
            tlas.nodes[index].a = lc;
            uint32_t rc = build_scene_rec(aabbs, begin + split, end, index);
            tlas.nodes[index].b = rc;
            tlas.nodes[index].leaf = false;
        }
        return index;
    }

    // ------------------------------------------------------------------------------------------
    // T2: slab tests (cuda.hpp:70-149)
    // ------------------------------------------------------------------------------------------
    template <bool SCENE> static bool ray_aabb(const Ray<T> &ray, const AABB<T> &aabb, T *min_t, T *max_t) {
        Vec3<T> t1 = elediv(aabb.lower_bound - ray.source, ray.direction);
        Vec3<T> t2 = elediv(aabb.upper_bound - ray.source, ray.direction);
        for (int i = 0; i < 3; i++)
            if (t2[i] < t1[i])
                std::swap(t1[i], t2[i]);
        bool updated = false;
        T t1max = T(0.f), t2min = T(0.f);
        for (int i = 0; i < 3; i++) {
            if (!(isfinite(t1[i]) && isfinite(t2[i])))
                continue; // Q3
            if (!updated || (t1max < t1[i]))
                t1max = t1[i];
            if (!updated || (t2[i] < t2min))
                t2min = t2[i];
            updated = true;
        }
        if (!updated)
            return false;
        bool out;
        if (SCENE) // `0 <= t2min + ...`: int vs T compares in float
            out = t1max <= t2min + T(0.02f) && 0 <= float(t2min + T(0.02f));
        else
            out = t1max <= t2min * T(1.001953f) && 0 <= float(t2min);
        if (out) {
            *min_t = t1max;
            *max_t = t2min;
        }
        return out;
    }

    // ------------------------------------------------------------------------------------------
    // T3: ray_triangle_intersection (cuda.hpp:193-354)
    // ------------------------------------------------------------------------------------------
    bool ray_triangle(const Ray<T> &ray, uint32_t inscene_objid, uint32_t vao_index, uint32_t inobject_offset,
                      Intersection<T> &isect, T min_distance, T max_distance, Counters &cnt) const {
        cnt.tri_tests++;
        const lprt_mesh_vao &vao = vaos[vao_index];
        uint32_t offset = inobject_offset + vao.ebo_offset;
        Vec3<T> v2 = vbo[ebo[offset + 2]].position;
        Vec3<T> O = ray.source, D = ray.direction;
        const Mat3<T> &M = M_shift[offset / 3];
        O[0] = O[0] - v2[0];
        O[1] = O[1] - v2[1];
        O[2] = O[2] - v2[2];
        T Ox_0 = O[0] * M.d[0][0], Ox_1 = O[1] * M.d[0][1], Ox_2 = O[2] * M.d[0][2];
        T Dx_0 = D[0] * M.d[0][0], Dx_1 = D[1] * M.d[0][1], Dx_2 = D[2] * M.d[0][2];
        T Oy_0 = O[0] * M.d[1][0], Oy_1 = O[1] * M.d[1][1], Oy_2 = O[2] * M.d[1][2];
        T Dy_0 = D[0] * M.d[1][0], Dy_1 = D[1] * M.d[1][1], Dy_2 = D[2] * M.d[1][2];
        T Ox = Ox_0 + Ox_1 + Ox_2;
        T Dx = Dx_0 + Dx_1 + Dx_2;
        T Oy = Oy_0 + Oy_1 + Oy_2;
        T Dy = Dy_0 + Dy_1 + Dy_2;
        float Oz = float(O[0]) * float(M.d[2][0]) + float(O[1]) * float(M.d[2][1]) + float(O[2]) * float(M.d[2][2]);
        float Dz = float(D[0]) * float(M.d[2][0]) + float(D[1]) * float(M.d[2][1]) + float(D[2]) * float(M.d[2][2]);
        float invDz = float(1.0 / double(Dz));
        float t = -Oz * invDz;
        T t_Dx = T(t * float(Dx)); // float * T -> float, rounded once
        T t_Dy = T(t * float(Dy));
        T u = Ox + t_Dx;
        T v = Oy + t_Dy;
        T delta1 = T(std::pow(2.f, -10.f));
        T delta2 = T(std::pow(2.f, -8.f));
        T error_Ox = delta1 * (abs(Ox_0) + abs(Ox_1) + abs(Ox_2)) + delta2 * (abs(Ox_0) + abs(Ox_1) + abs(Ox_2));
        T error_Dx = delta1 * (abs(Dx_0) + abs(Dx_1) + abs(Dx_2)) + delta2 * (abs(Dx_0) + abs(Dx_1) + abs(Dx_2));
        T error_Oy = delta1 * (abs(Oy_0) + abs(Oy_1) + abs(Oy_2)) + delta2 * (abs(Oy_0) + abs(Oy_1) + abs(Oy_2));
        T error_Dy = delta1 * (abs(Dy_0) + abs(Dy_1) + abs(Dy_2)) + delta2 * (abs(Dy_0) + abs(Dy_1) + abs(Dy_2));
        T error_u = (error_Ox + T(t) * error_Dx + delta1 * (abs(Ox) + T(3.f) * abs(t_Dx))) * T(0.2f);
        T error_v = (error_Oy + T(t) * error_Dy + delta1 * (abs(Oy) + T(3.f) * abs(t_Dy))) * T(0.2f);

        if (t > float(min_distance) && t < float(isect.t) && t < float(max_distance)) {
            float w = 1 - float(u) - float(v); // `1 - u - v`: int - T - T is evaluated in float
            if ((u >= T(-float(error_u)) && u <= T(0.0f)) || (v >= T(-float(error_v)) && v <= T(0.0f)) ||
                (w >= float(T(-float(error_v + error_u))) && w <= float(T(0.0f)))) {
                cnt.fallback_tests++;
                const Mat3<float> &F = M_shift_f32[offset / 3];
                Vec3<float> v2f = position_f32[ebo[offset + 2]];
                float Of[3] = {float(ray.source[0]) - v2f[0], float(ray.source[1]) - v2f[1], float(ray.source[2]) - v2f[2]};
                float Df[3] = {float(ray.direction[0]), float(ray.direction[1]), float(ray.direction[2])};
                float Oxf = Of[0] * F.d[0][0] + Of[1] * F.d[0][1] + Of[2] * F.d[0][2];
                float Dxf = Df[0] * F.d[0][0] + Df[1] * F.d[0][1] + Df[2] * F.d[0][2];
                float Oyf = Of[0] * F.d[1][0] + Of[1] * F.d[1][1] + Of[2] * F.d[1][2];
                float Dyf = Df[0] * F.d[1][0] + Df[1] * F.d[1][1] + Df[2] * F.d[1][2];
                float Ozf = Of[0] * F.d[2][0] + Of[1] * F.d[2][1] + Of[2] * F.d[2][2];
                float Dzf = Df[0] * F.d[2][0] + Df[1] * F.d[2][1] + Df[2] * F.d[2][2];
                float tf = -Ozf / Dzf;
                float uf = Oxf + tf * Dxf;
                float vf = Oyf + tf * Dyf;
                if (tf > float(min_distance) && tf < float(isect.t) && tf < float(max_distance)) {
                    if (uf > 0) {
                        if (vf > 0 && uf + vf < 1) {
                            isect.hit = true;
                            isect.t = T(tf);
                            isect.barycentric[0] = T(uf);
                            isect.barycentric[1] = T(vf);
                            isect.barycentric[2] = T(1 - uf - vf);
                            isect.inobject_offset = inobject_offset;
                            isect.inscene_objid = inscene_objid;
                            return true;
                        }
                    }
                }
                return false;
            }
            if (u > T(-float(error_u))) {
                // `DataT(1.0 + error_u + error_v)`: the sum is formed in double, then double->float->T
                if (v > T(-float(error_v)) && u + v < T(float(1.0 + double(float(error_u)) + double(float(error_v))))) {
                    isect.hit = true;
                    isect.t = T(t);
                    isect.barycentric[0] = u;
                    isect.barycentric[1] = v;
                    isect.barycentric[2] = T(1 - float(u) - float(v));
                    isect.inobject_offset = inobject_offset;
                    isect.inscene_objid = inscene_objid;
                    return true;
                }
            }
        }
        return false;
    }

    // ------------------------------------------------------------------------------------------
    // T4: search_intersection_in_object (cuda.hpp:415-504), stackless parent-pointer walk
    // ------------------------------------------------------------------------------------------
    template <bool find_any>
    void search_in_object(const Ray<T> &ray, uint32_t inscene_objid, uint32_t vao_index, SkipGeometry skip,
                          Intersection<T> &isect, T min_distance, T max_distance, Counters &cnt) const {
        const BVH<T> &bvh = blas[vao_index];
        cnt.object_visits++;
        uint32_t last = UINT32_MAX, cur = 0;
        T tmn, tmx;
        while (cur < UINT32_MAX) {
            const BVHNode &node = bvh.nodes[cur];
            if (last == node.parent) {
                cnt.blas_node_visits++;
                if (ray_aabb<false>(ray, bvh.boxes[cur], &tmn, &tmx) && tmn < isect.t && tmn < max_distance &&
                    tmx > min_distance) {
                    if (node.leaf) {
                        for (uint32_t i = node.a; i < node.a + node.b; ++i) {
                            uint32_t off = bvh.geometry[i];
                            if (off == skip.inobject_offset && vao_index == skip.vao_index)
                                continue;
                            ray_triangle(ray, inscene_objid, vao_index, off, isect, min_distance, max_distance, cnt);
                            if (find_any && isect.hit)
                                return;
                        }
                        last = cur;
                        cur = node.parent;
                    } else {
                        last = cur;
                        if (node.a < UINT32_MAX)
                            cur = node.a;
                        else if (node.b < UINT32_MAX)
                            cur = node.b;
                        else
                            cur = node.parent;
                    }
                } else {
                    last = cur;
                    cur = node.parent;
                }
            } else {
                if (last == node.a) {
                    if (node.b < UINT32_MAX) {
                        last = cur;
                        cur = node.b;
                    } else {
                        last = cur;
                        cur = node.parent;
                    }
                } else {
                    last = cur;
                    cur = node.parent;
                }
            }
        }
    }

    // ------------------------------------------------------------------------------------------
    // T5: scene_bvh.traversal + cast_ray (cuda.hpp:517-631)
    // ------------------------------------------------------------------------------------------
    template <bool find_any>
    Intersection<T> cast_ray(const Ray<T> &ray, SkipGeometry skip, T min_distance, T max_distance, Counters &cnt) const {
        Intersection<T> isect;
        cnt.rays++;
        if (tlas.nodes.empty())
            return isect;
        uint32_t last = UINT32_MAX, cur = 0;
        T tmn = T(0.f), tmx = T(0.f);
        while (cur < UINT32_MAX) {
            const BVHNode &node = tlas.nodes[cur];
            if (last == node.parent) {
                cnt.tlas_node_visits++;
                if (ray_aabb<true>(ray, tlas.boxes[cur], &tmn, &tmx) && tmn < max_distance && tmx > min_distance) {
                    if (node.leaf) {
                        for (uint32_t i = node.a; i < node.a + node.b; ++i) {
                            cnt.aabbw_tests++;
                            if (ray_aabb<true>(ray, tlas.aabbs_W[i], &tmn, &tmx) && tmn < max_distance &&
                                tmx > min_distance) {
                                uint32_t inscene_objid = tlas.geometry[i];
                                const ObjectConstant<T> &oc = objects[inscene_objid];
                                search_in_object<find_any>(ray.transformed(oc.transform_W2L), inscene_objid, oc.objectid,
                                                           skip, isect, min_distance, max_distance, cnt);
                                if (find_any && isect.hit)
                                    return isect;
                            }
                        }
                        last = cur;
                        cur = node.parent;
                    } else {
                        last = cur;
                        if (node.a < UINT32_MAX)
                            cur = node.a;
                        else if (node.b < UINT32_MAX)
                            cur = node.b;
                        else
                            cur = node.parent;
                    }
                } else {
                    last = cur;
                    cur = node.parent;
                }
            } else {
                if (last == node.a) {
                    if (node.b < UINT32_MAX) {
                        last = cur;
                        cur = node.b;
                    } else {
                        last = cur;
                        cur = node.parent;
                    }
                } else {
                    last = cur;
                    cur = node.parent;
                }
            }
        }
        return isect;
    }

    // Exhaustive closest hit: the reference's triangle test (T3) against EVERY triangle of EVERY instance, no boxes.  Not a
    // reference function -- the yardstick for hits that two hierarchies report differently: a traversal can only LOSE
    // candidates (box culls), so whatever cast_ray finds over any tree has t >= this one's, and the answer that equals it
    // is the better one (adjudicate `cull` / `tie` mismatches against a brute-force intersection).
    Intersection<T> cast_ray_exhaustive(const Ray<T> &ray, SkipGeometry skip, T min_distance, T max_distance, Counters &cnt) const {
        Intersection<T> isect;
        for (uint32_t id = 0; id < objects.size(); id++) {
            const ObjectConstant<T> &oc = objects[id];
            if (oc.objectid >= vaos.size())
                continue;
            const Ray<T> lr = ray.transformed(oc.transform_W2L);
            const uint32_t n = vaos[oc.objectid].ebo_length;
            for (uint32_t off = 0; off + 2 < n; off += 3) {
                if (off == skip.inobject_offset && oc.objectid == skip.vao_index)
                    continue;
                ray_triangle(lr, id, oc.objectid, off, isect, min_distance, max_distance, cnt);
            }
        }
        return isect;
    }

    // ------------------------------------------------------------------------------------------
    // G1: CUDARenderGISettings::get_primary_ray / get_primary_ray_direction (cuda.hpp:799-832)
    // ------------------------------------------------------------------------------------------
    void primary_xy(uint32_t sx_u, uint32_t sy_u, T &x, T &y) const {
        T screen_x = T(float(sx_u)), screen_y = T(float(sy_u));
        T nx = T(2.0f) * (screen_x + T(0.5f)) / T(float(W)) - T(1.f);
        T ny = T(2.0f) * (screen_y + T(0.5f)) / T(float(H)) - T(1.f);
        T max_y = tan(cam_fovy / T(2.f));
        y = ny * max_y;
        x = nx * max_y * T(float(W)) / T(float(H)); // omit_ar = true
    }
    Ray<T> get_primary_ray(uint32_t sx, uint32_t sy) const {
        T x, y;
        primary_xy(sx, sy, x, y);
        Ray<T> ray;
        ray.source = mk3(T(0.f), T(0.f), T(0.f));
        ray.direction = mk3(x, y, T(-1.f)).normalized();
        ray = ray.transformed(cam_L2W);
        ray.direction.normalize();
        return ray;
    }
    Vec3<T> get_primary_ray_direction(uint32_t sx, uint32_t sy) const {
        T x, y;
        primary_xy(sx, sy, x, y);
        return xyz(cam_L2W * mk4(x, y, T(-1.f), T(0.f))).normalized();
    }

    static std::array<uint8_t, 4> to_color_uchar(const Vec3<T> &c) { // cuda.hpp:898-902
        std::array<uint8_t, 4> o;
        for (int k = 0; k < 3; k++)
            o[k] = uint8_t(clip_both<int>(0, 255, f2i(0.5f + float(c[k] * T(255.f)))));
        o[3] = 255;
        return o;
    }
    static Vec3<T> to_color_float(const std::array<uint8_t, 4> &c) { // cuda.hpp:890-892
        return mk3(T(c[0] / 255.f), T(c[1] / 255.f), T(c[2] / 255.f));
    }

    struct HitAttributes {
        Vec3<T> position, normal, tangent, color;
        Vec2<T> uv0, uv1;
    };
    // attribute interpolation shared by fill_gbuffer (cuda.hpp:967-987) and trace_gi (cuda.hpp:2107-2128)
    HitAttributes interpolate(const Intersection<T> &is) const {
        const ObjectConstant<T> &oc = objects[is.inscene_objid];
        const lprt_mesh_vao &vao = vaos[oc.objectid];
        const RDVertex<T> &v1 = vbo[ebo[vao.ebo_offset + is.inobject_offset + 0]];
        const RDVertex<T> &v2 = vbo[ebo[vao.ebo_offset + is.inobject_offset + 1]];
        const RDVertex<T> &v3 = vbo[ebo[vao.ebo_offset + is.inobject_offset + 2]];
        HitAttributes h;
        Vec3<T> position = lerp3(is.barycentric, v1.position, v2.position, v3.position);
        Vec3<T> normal = lerp3(is.barycentric, v1.normal, v2.normal, v3.normal).normalized();
        h.color = lerp3(is.barycentric, v1.color, v2.color, v3.color);
        Vec3<T> tangent = lerp3(is.barycentric, v1.tangent, v2.tangent, v3.tangent).normalized();
        h.uv0 = lerp2(is.barycentric, v1.uv0, v2.uv0, v3.uv0);
        h.uv1 = lerp2(is.barycentric, v1.uv1, v2.uv1, v3.uv1);
        const Mat4<T> &tr = oc.transform_L2W;
        h.normal = xyz(tr * as_dir4(normal)).normalized();
        h.tangent = xyz(tr * as_dir4(tangent)).normalized();
        h.position = to_vec3_as_pos(tr * as_pos4(position));
        return h;
    }

    template <typename F> void parallel_rows(F f) {
        std::vector<Counters> local(size_t(threads > 0 ? threads : 1));
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int64_t y = 0; y < int64_t(H); y++) {
            int tid = 0;
#ifdef _OPENMP
            tid = omp_get_thread_num();
#endif
            for (uint32_t x = 0; x < W; x++)
                f(x, uint32_t(y), local[size_t(tid)]);
        }
        for (auto &c : local)
            counters.add(c);
    }

    // ------------------------------------------------------------------------------------------
    // G2: fill_gbuffer_in_cuda (cuda.hpp:950-1012)
    // ------------------------------------------------------------------------------------------
    void fill_gbuffer() {
        parallel_rows([&](uint32_t x, uint32_t y, Counters &cnt) {
            size_t tid = size_t(y) * W + x;
            Ray<T> ray = get_primary_ray(x, y);
            Intersection<T> is = cast_ray<false>(ray, SkipGeometry(), T(0.f), T(1e5f), cnt);
            if (is.hit) {
                HitAttributes h = interpolate(is);
                g_position[tid] = mk4(h.position[0], h.position[1], h.position[2], T(1.f));
                g_normal_depth[tid] = mk4(h.normal[0], h.normal[1], h.normal[2], T(1.f)); // Q11: depth = 1
                g_tangent[tid] = mk4(h.tangent[0], h.tangent[1], h.tangent[2], T(1.f));
                g_color[tid] = to_color_uchar(h.color);
                g_uv[tid] = mk4(h.uv0[0], h.uv0[1], h.uv1[0], h.uv1[1]);
                g_objid[tid] = is.inscene_objid;
                g_offset[tid] = is.inobject_offset;
            } else {
                g_position[tid] = Vec4<T>();
                g_normal_depth[tid] = Vec4<T>();
                g_tangent[tid] = Vec4<T>();
                g_color[tid] = {0, 0, 0, 0};
                g_uv[tid] = Vec4<T>();
                g_objid[tid] = 0;
                g_offset[tid] = 0;
            }
        });
    }

    // ------------------------------------------------------------------------------------------
    // V1: generate_temporal_map_step1/2 (cuda.hpp:2265-2416), sample_linear (cuda.hpp:1085-1115)
    // ------------------------------------------------------------------------------------------
    void generate_temporal_map(uint64_t seed) {
        size_t n = size_t(W) * H;
        std::vector<uint8_t> prev_count(n);
        for (size_t i = 0; i < n; i++)
            prev_count[i] = svgf_temporal_map[i].frame_count; // snapshot: the reference races here
        bool first = frame_index == 0;
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int64_t yy = 0; yy < int64_t(H); yy++) {
            XorwowState row_state = xorwow_init(seed, uint64_t(yy) * W);
            for (uint32_t x = 0; x < W; x++) {
                uint32_t y = uint32_t(yy);
                size_t tid = size_t(y) * W + x;
                XorwowState st = row_state;
                xorwow_skip().next_subsequence(row_state.v);
                TemporalMap<T> &sm = svgf_temporal_map[tid];
                TemporalMap<T> &tm = taa_temporal_map[tid];
                bool empty = g_color[tid][3] == 0;
                if (empty) {
                    valid_frame_count[tid] = 0; // step2 then writes frame_count = 0 (== cuda.hpp:2278 without the race)
                    continue;
                }
                if (first) { // DEVIATION Q7: frame 0 has no history for any pixel
                    valid_frame_count[tid] = 0;
                    tm.frame_count = 0;
                    for (int i = 0; i < 4; i++) {
                        sm.weights[i] = T(0.f);
                        tm.weights[i] = T(0.f);
                    }
                    continue;
                }
                const ObjectConstant<T> &oc = objects[g_objid[tid]];
                Vec4<T> cW = g_position[tid];
                cW[3] = T(1.f);
                Vec4<T> cL = oc.transform_W2L * cW;
                Vec4<T> cWlf = last_frame_L2W[oc.objectid] * cL;
                Vec4<T> cC = last_frame_transform_W2C * cWlf;
                T g_fx = (T(1.f) + cC[0] / cC[3]) / T(2.f) * T(float(W));
                T g_fy = (T(1.f) + cC[1] / cC[3]) / T(2.f) * T(float(H));
                { // svgf 2x2
                    for (int i = 0; i < 4; i++)
                        sm.weights[i] = T(0.f);
                    T fx = g_fx - T(0.5f), fy = g_fy - T(0.5f);
                    int lx = f2i(float(fx)), ly = f2i(float(fy));
                    int ux = lx + 1, uy = ly + 1;
                    int count = 0;
                    int nx[4] = {lx, lx, ux, ux}, ny[4] = {ly, uy, ly, uy};
                    T wts[4] = {(T(float(ux)) - fx) * (T(float(uy)) - fy), (T(float(ux)) - fx) * (fy - T(float(ly))),
                                (fx - T(float(lx))) * (T(float(uy)) - fy), (fx - T(float(lx))) * (fy - T(float(ly)))};
                    T total = T(0.f);
                    for (int i = 0; i < 4; i++) {
                        if (nx[i] >= 0 && nx[i] < int(W) && ny[i] >= 0 && ny[i] < int(H)) {
                            size_t q = size_t(ny[i]) * W + size_t(nx[i]);
                            if (last_frame_id[q].object_id == oc.objectid) {
                                count = std::max<int>(count, prev_count[q]);
                                sm.weights[i] = wts[i];
                                total += wts[i];
                                sm.pos[i][0] = uint16_t(nx[i]);
                                sm.pos[i][1] = uint16_t(ny[i]);
                            }
                        }
                    }
                    if (total == T(0.f)) {
                        valid_frame_count[tid] = 0;
                    } else {
                        for (int i = 0; i < 4; i++) {
                            if (count < 255)
                                count += 1; // Q6: +4 per frame
                            sm.weights[i] /= total;
                        }
                        valid_frame_count[tid] = uint8_t(count);
                    }
                }
                { // taa 2x2
                    for (int i = 0; i < 4; i++)
                        tm.weights[i] = T(0.f);
                    T fx = g_fx - T(xorwow_uniform(st));
                    T fy = g_fy - T(xorwow_uniform(st));
                    int lx = f2i(float(fx)), ly = f2i(float(fy));
                    int ux = lx + 1, uy = ly + 1;
                    int nx[4] = {lx, lx, ux, ux}, ny[4] = {ly, uy, ly, uy};
                    T wts[4] = {(T(float(ux)) - fx) * (T(float(uy)) - fy), (T(float(ux)) - fx) * (fy - T(float(ly))),
                                (fx - T(float(lx))) * (T(float(uy)) - fy), (fx - T(float(lx))) * (fy - T(float(ly)))};
                    T total = T(0.f);
                    bool same = false;
                    for (int i = 0; i < 4; i++) {
                        if (nx[i] >= 0 && nx[i] < int(W) && ny[i] >= 0 && ny[i] < int(H)) {
                            size_t q = size_t(ny[i]) * W + size_t(nx[i]);
                            tm.weights[i] = wts[i];
                            total += wts[i];
                            tm.pos[i][0] = uint16_t(nx[i]);
                            tm.pos[i][1] = uint16_t(ny[i]);
                            if (last_frame_id[q].object_id == oc.objectid)
                                same = true;
                        }
                    }
                    if (!same) {
                        tm.frame_count = 0;
                    } else {
                        for (int i = 0; i < 4; i++)
                            tm.weights[i] /= total;
                        tm.frame_count = 1;
                    }
                }
            }
        }
        for (size_t i = 0; i < n; i++) // step2 (cuda.hpp:2407-2416)
            svgf_temporal_map[i].frame_count = valid_frame_count[i];
    }
    Vec3<T> sample_linear3(const TemporalMap<T> &map, const std::vector<Vec3<T>> &arr, const Vec3<T> &dflt) const {
        if (map.frame_count == 0)
            return dflt;
        T sum_weight = T(0.f);
        Vec3<T> result;
        for (int i = 0; i < 4; i++) {
            T weight = map.weights[i];
            if (weight == T(0.f))
                continue;
            sum_weight += weight;
            result += weight * arr[size_t(map.pos[i][1]) * W + map.pos[i][0]];
        }
        return result / sum_weight; // Matrix::operator/: multiply by the rounded reciprocal
    }
    T sample_linear1(const TemporalMap<T> &map, const std::vector<T> &arr, T dflt) const {
        if (map.frame_count == 0)
            return dflt;
        T sum_weight = T(0.f);
        T result = T(0.f);
        for (int i = 0; i < 4; i++) {
            T weight = map.weights[i];
            if (weight == T(0.f))
                continue;
            sum_weight += weight;
            result += weight * arr[size_t(map.pos[i][1]) * W + map.pos[i][0]];
        }
        return result / sum_weight; // scalar: a true division
    }

    // ------------------------------------------------------------------------------------------
    // S: shade<T,4,FIRST_ROUND,NO_GI> (cuda.hpp:1751-2056)
    // ------------------------------------------------------------------------------------------
    template <bool FIRST_ROUND, bool NO_GI> void shade(int round, uint64_t seed) {
        const size_t N = size_t(W) * H;
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int64_t yy = 0; yy < int64_t(H); yy++) {
            XorwowState row_state{};
            if (!NO_GI)
                row_state = xorwow_init(seed, uint64_t(yy) * W);
            for (uint32_t x = 0; x < W; x++) {
                uint32_t y = uint32_t(yy);
                size_t tid = size_t(y) * W + x;
                XorwowState rand_state = row_state; // curand_init(seed, thread_id, 0)
                if (!NO_GI)
                    xorwow_skip().next_subsequence(row_state.v);
                shade_pixel<FIRST_ROUND, NO_GI>(round, x, y, tid, N, rand_state);
            }
        }
    }
    template <bool FIRST_ROUND, bool NO_GI>
    void shade_pixel(int round, uint32_t x, uint32_t y, size_t tid, size_t N, XorwowState &rand_state) {
        Vec3<T> position, raw_normal, raw_tangent, vertex_color;
        Vec2<T> uv0, uv1;
        Material<T> material;
        uint32_t inscene_objid, inobject_offset, objectid;
        TraceLightCommand<T> *lc = &trace_light_commands[tid * 4];
        if (FIRST_ROUND) {
            const auto &c4 = g_color[tid];
            vertex_color = to_color_float(c4);
            if (c4[3] == 0) {
                for (int i = 0; i < 4; i++)
                    lc[i].slot = UINT8_MAX;
                if (sky_valid) // cuda.hpp:1786-1794
                    di_intensity[tid * 4 + 0] = sample_skybox(get_primary_ray_direction(x, y));
                gi_valid[tid] = 0; // DEVIATION Q13
                return;
            }
            inscene_objid = g_objid[tid];
            inobject_offset = g_offset[tid];
            position = xyz(g_position[tid]);
            raw_normal = xyz(g_normal_depth[tid]);
            raw_tangent = xyz(g_tangent[tid]);
            uv0 = mk2(g_uv[tid][0], g_uv[tid][1]);
            uv1 = mk2(g_uv[tid][2], g_uv[tid][3]);
            material = materials[objects[inscene_objid].materialid];
            objectid = objects[inscene_objid].objectid;
        } else {
            const PixelShaderInput<T> &sc = shade_commands[tid];
            if (sc.type == PixelShaderInput<T>::INVALID) {
                for (int i = 0; i < 4; i++)
                    lc[i].slot = UINT8_MAX;
                return;
            } else if (sc.type == PixelShaderInput<T>::SKYBOX) {
                for (int i = 0; i < 4; i++)
                    lc[i].slot = UINT8_MAX;
                // DEVIATION Q5: an invalid skybox contributes 0 (di_intensity stays cleared)
                if (sky_valid) // cuda.hpp:1828-1832
                    di_intensity[tid * 4 + 0] = sample_skybox(-view_direction[tid].normalized());
                return;
            } else {
                vertex_color = sc.color;
                inscene_objid = sc.inscene_objid;
                inobject_offset = sc.inobject_offset;
                position = sc.position;
                raw_normal = sc.normal;
                raw_tangent = sc.tangent;
                uv0 = sc.uv0;
                uv1 = sc.uv1;
                material = materials[sc.materialid];
                objectid = objects[inscene_objid].objectid;
            }
        }
        Vec3<T> color = material.color;
        if (material.uv_color < 2) { // cuda.hpp:1850-1855
            const Vec2<T> &uv = material.uv_color == 0 ? uv0 : uv1;
            auto tc = tex2d(material.tex_color, float(uv[0]), float(uv[1]));
            color = mk3(T(tc[0]), T(tc[1]), T(tc[2]));
        }
        color = elemul(color, vertex_color);
        Vec3<T> normal = raw_normal;
        Vec3<T> view_dir;
        if (FIRST_ROUND)
            view_dir = -get_primary_ray_direction(x, y);
        else
            view_dir = view_direction[tid];
        if (dot_product(view_dir, normal) < T(0.f)) {
            if (material.double_sided) {
                normal = -normal;
            } else {
                for (int i = 0; i < 4; i++)
                    lc[i].slot = UINT8_MAX;
                if (FIRST_ROUND)
                    gi_valid[tid] = 0; // DEVIATION Q13
                return;
            }
        }
        Vec3<T> raw_bitangent = cross_product(raw_normal, raw_tangent).normalized();
        raw_tangent = cross_product(raw_bitangent, raw_normal).normalized();
        Vec3<T> tangent = raw_tangent, bitangent = raw_bitangent;
        skips[tid] = SkipGeometry{objectid, inobject_offset};
        intensity[size_t(round) * N + tid] = material.emission;
        if (FIRST_ROUND)
            albedo[tid] = color;

        if (!NO_GI) {
            TraceGICommand<T> &cmd = trace_gi_commands[tid];
            gi_valid[tid] = 1;
            cmd.x = uint16_t(x);
            cmd.y = uint16_t(y);
            cmd.source = position;
            Vec3<T> &mult_out = gi_material_multiplier[size_t(round) * N + tid];
            const T nanT = T(NAN);
            if (material.roughness < T(0.1f)) {
                if (T(xorwow_uniform(rand_state)) < material.metallic) {
                    cmd.direction = symmetric_vector(view_dir, normal).normalized();
                    mult_out = glassy_brdf(material.metallic, view_dir, cmd.direction, normal).get_brdf(color) / material.metallic;
                    view_direction[tid] = -cmd.direction;
                } else {
                    if (T(xorwow_uniform(rand_state)) < T(0.6f)) {
                        cmd.direction = symmetric_vector(view_dir, normal).normalized();
                        BRDF<T> brdf = glassy_brdf(material.metallic, view_dir, cmd.direction, normal);
                        T pdf = (T(1.f) - material.metallic) * T(0.6f);
                        if (FIRST_ROUND)
                            gi_material_multiplier[tid] = mk3(brdf.colored / pdf, brdf.white / pdf, nanT);
                        else
                            mult_out = brdf.get_brdf(color) / pdf;
                        view_direction[tid] = -cmd.direction;
                    } else { // Q10: uniform direction from 3 normals
                        T n0 = T(xorwow_normal(rand_state)), n1 = T(xorwow_normal(rand_state)), n2 = T(xorwow_normal(rand_state));
                        Vec3<T> reflect_dir = mk3(n0, n1, n2);
                        reflect_dir.normalize();
                        T cosine;
                        if ((cosine = dot_product(reflect_dir, normal)) < T(0.f)) {
                            reflect_dir = -reflect_dir;
                            cosine = T(-float(cosine));
                        }
                        cmd.direction = reflect_dir;
                        view_direction[tid] = -reflect_dir;
                        BRDF<T> brdf = material_brdf(material.metallic, material.roughness, view_dir, reflect_dir, normal);
                        T multiplier = cosine * T(float(M_PI * 2));
                        if (FIRST_ROUND)
                            gi_material_multiplier[tid] = mk3(brdf.colored * multiplier, brdf.white * multiplier, nanT);
                        else
                            mult_out = brdf.get_brdf(color) * multiplier;
                    }
                }
            } else {
                T pdf;
                Vec3<T> reflect_dir;
                T cosine;
                if (T(xorwow_uniform(rand_state)) < material.metallic) {
                    T a = material.roughness * material.roughness;
                    T a2 = a * a;
                    Vec3<T> half_t = sample_ggx(a2, rand_state);
                    pdf = pdf_ggx_reflect(half_t[2], a2) * material.metallic;
                    Vec3<T> half_w = (tangent * half_t[0] + bitangent * half_t[1] + normal * half_t[2]).normalized(); // cuda.hpp:1737-1740
                    reflect_dir = symmetric_vector(view_dir, half_w);
                    cosine = tmax(T(0.05f), dot_product(reflect_dir, half_w));
                    pdf /= T(4 * float(cosine)); // `4 * cosine` is int * T -> float
                } else {
                    reflect_dir[0] = T(xorwow_normal(rand_state));
                    reflect_dir[1] = T(xorwow_normal(rand_state));
                    reflect_dir[2] = T(xorwow_normal(rand_state));
                    if ((cosine = dot_product(reflect_dir, normal)) < T(0.f)) {
                        reflect_dir = -reflect_dir;
                        cosine = T(-float(cosine));
                    }
                    pdf = T(float(0.5 / M_PI)) * (T(1.f) - material.metallic);
                }
                reflect_dir.normalize();
                cmd.direction = reflect_dir;
                view_direction[tid] = -reflect_dir;
                BRDF<T> brdf = material_brdf(material.metallic, material.roughness, view_dir, reflect_dir, normal);
                T multiplier = cosine / pdf;
                if (FIRST_ROUND)
                    gi_material_multiplier[tid] = mk3(brdf.colored * multiplier, brdf.white * multiplier, nanT);
                else
                    mult_out = multiplier * brdf.get_brdf(color);
            }
        } else {
            gi_valid[tid] = 0; // cmd.x = UINT16_MAX
        }
        if (NO_GI) // Q4: overwrites the emission with SKYBOX_COLOR (= 0) * color * 0.5
            intensity[size_t(round) * N + tid] = elemul(mk3(T(0.f), T(0.f), T(0.f)), color) * T(0.5f);

        for (int i = 0; i < 4; i++) {
            TraceLightCommand<T> &cmd = lc[i];
            cmd.x = uint16_t(x);
            cmd.y = uint16_t(y);
            cmd.source = position;
            cmd.slot = UINT8_MAX;
            if (i >= int(lights.size()))
                continue;
            const Light<T> &light = lights[size_t(i)];
            if (light.type == LPRT_LIGHT_POINT || light.type == LPRT_LIGHT_SPOT) {
                Vec3<T> ldv = light.position - position;
                Vec3<T> light_dir = ldv.normalized();
                T cosine = dot_product(light_dir, normal);
                if (cosine < T(0.f))
                    continue;
                T d2 = ldv.norm2_squared();
                Vec3<T> brdf = material_brdf(material.metallic, tmax(material.roughness, T(0.10f)), view_dir, light_dir, normal).get_brdf(color);
                cmd.direction = light_dir;
                cmd.slot = uint8_t(i);
                cmd.material_multiplier = (cosine / d2 / T(10.f)) * elemul(brdf, light.intensity);
                cmd.maximum_t = sqrt(d2);
            } else {
                Vec3<T> light_dir = -light.direction.normalized();
                T cosine = dot_product(light_dir, normal);
                if (cosine < T(0.f))
                    continue;
                Vec3<T> brdf = material_brdf(material.metallic, tmax(material.roughness, T(0.10f)), view_dir, light_dir, normal).get_brdf(color);
                cmd.direction = light_dir;
                cmd.slot = uint8_t(i);
                cmd.material_multiplier = cosine * elemul(brdf, light.intensity);
                cmd.maximum_t = T(1000.f);
            }
        }
    }

    // ------------------------------------------------------------------------------------------
    // TD / TG / A (cuda.hpp:2058-2159)
    // ------------------------------------------------------------------------------------------
    void trace_di_light() {
        parallel_rows([&](uint32_t x, uint32_t y, Counters &cnt) {
            size_t px = size_t(y) * W + x;
            for (int s = 0; s < 4; s++) {
                const TraceLightCommand<T> &c = trace_light_commands[px * 4 + size_t(s)];
                if (c.slot == UINT8_MAX)
                    continue;
                size_t shade_tid = size_t(c.y) * W + c.x;
                Ray<T> ray{c.source, c.direction};
                Intersection<T> is = cast_ray<true>(ray, skips[shade_tid], T(ray_moveforward_t<T>()), c.maximum_t, cnt);
                di_intensity[4 * shade_tid + c.slot] = (is.hit ? T(0.f) : T(1.f)) * c.material_multiplier;
            }
        });
    }
    void trace_gi() {
        parallel_rows([&](uint32_t x, uint32_t y, Counters &cnt) {
            size_t tid = size_t(y) * W + x;
            if (!gi_valid[tid])
                return;
            const TraceGICommand<T> &c = trace_gi_commands[tid];
            size_t shade_tid = size_t(c.y) * W + c.x;
            Ray<T> ray{c.source, c.direction};
            Intersection<T> is = cast_ray<false>(ray, skips[shade_tid], T(ray_moveforward_t<T>()), T(1e5f), cnt);
            PixelShaderInput<T> &sc = shade_commands[shade_tid];
            if (is.hit) {
                HitAttributes h = interpolate(is);
                sc.normal = h.normal;
                sc.tangent = h.tangent;
                sc.position = h.position;
                sc.color = h.color;
                sc.uv0 = h.uv0;
                sc.uv1 = h.uv1;
                sc.inscene_objid = is.inscene_objid;
                sc.inobject_offset = is.inobject_offset;
                sc.materialid = objects[is.inscene_objid].materialid;
                sc.type = PixelShaderInput<T>::COMMON;
            } else {
                sc.type = PixelShaderInput<T>::SKYBOX;
            }
        });
    }
    void accumulate_di_light(int round) {
        size_t N = size_t(W) * H;
        for (size_t tid = 0; tid < N; tid++) {
            Vec3<T> fc;
            for (int i = 0; i < 4; i++)
                fc += di_intensity[tid * 4 + size_t(i)];
            intensity[N * size_t(round) + tid] += fc;
        }
    }
    // write_clean_color (cuda.hpp:2170-2199)
    void write_clean_color(bool add_direct_out) {
        size_t N = size_t(W) * H;
        for (size_t tid = 0; tid < N; tid++) {
            Vec3<T> m0 = gi_material_multiplier[tid];
            Vec3<T> fc;
            if (add_direct_out)
                fc = intensity[tid];
            Vec3<T> i1 = intensity[N + tid];
            if (isnan(m0[2])) {
                mul_gi_colored[tid] = m0[0] * i1;
                mul_gi_white[tid] = m0[1] * i1;
            } else if (add_direct_out) {
                fc += elemul(i1, m0);
            }
            color_inprogress[tid] = fc;
        }
    }

    // ------------------------------------------------------------------------------------------
    // V2: SVGF (cuda.hpp:1187-1473, 1636-1711)
    // ------------------------------------------------------------------------------------------
    void preprocess_normal_depth(SVGF &s) {
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int64_t y = 0; y < int64_t(H); y++)
            for (int64_t x = 0; x < int64_t(W); x++) {
                size_t tid = size_t(y) * W + size_t(x);
                const Vec4<T> &nd = g_normal_depth[tid];
                s.normal[tid] = xyz(nd);
                T dp = nd[3];
                s.depth[tid] = dp;
                Vec2<T> g;
                // (a one-pixel-wide or one-pixel-high image has no neighbour to difference against: the reference reads past
                // the surface there; here the gradient is 0)
                if (x == 0)
                    g[0] = W > 1 ? T(g_normal_depth[tid + 1][3] - dp) : T(0);
                else
                    g[0] = dp - g_normal_depth[tid - 1][3];
                if (y == 0)
                    g[1] = H > 1 ? T(g_normal_depth[tid + W][3] - dp) : T(0);
                else
                    g[1] = dp - g_normal_depth[tid - W][3];
                s.depth_gradient[tid] = g;
            }
    }
    void temporal_accumulation_color(std::vector<Vec3<T>> &color_inout, SVGF &s, T color_taa_w) {
        std::vector<Vec3<T>> input = color_inout; // snapshot: the reference updates in place and races (Appendix C.2)
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int64_t py = 0; py < int64_t(H); py++)
            for (int64_t px = 0; px < int64_t(W); px++) {
                size_t tid = size_t(py) * W + size_t(px);
                const TemporalMap<T> &ti = svgf_temporal_map[tid];
                Vec3<T> p_color = input[tid];
                Vec3<T> miu1, miu2, weights;
                for (int i = -4; i <= 4; i++)
                    for (int j = -4; j <= 4; j++) {
                        int64_t qx = px + i, qy = py + j;
                        if (qx < 0 || qx >= int64_t(W) || qy < 0 || qy >= int64_t(H))
                            continue;
                        const Vec3<T> &q = input[size_t(qy) * W + size_t(qx)];
                        for (int k = 0; k < 3; k++)
                            if (isfinite(q[k])) {
                                weights[k] += T(1.f);
                                miu1[k] += q[k];
                                miu2[k] += q[k] * q[k];
                            }
                    }
                miu1 = elediv(miu1, weights);
                miu2 = elediv(miu2, weights);
                value_or(p_color, miu1);
                Vec3<T> var = miu2 - elemul(miu1, miu1);
                for (int i = 0; i < 3; i++) {
                    T sd = sqrt(var[i]);
                    p_color[i] = clip_both(miu1[i] - T(0.5f) * sd, miu1[i] + T(0.5f) * sd, p_color[i]);
                }
                Vec3<T> hist = sample_linear3(ti, s.color_history, p_color);
                value_or(hist, p_color);
                Vec3<T> integrated = color_taa_w * p_color + (T(1.f) - color_taa_w) * hist;
                color_inout[tid] = integrated;
                s.illum[tid] = T(0.2126f) * integrated[0] + T(0.7152f) * integrated[1] + T(0.0722f) * integrated[2];
            }
    }
    void temporal_accumulation_moments(SVGF &s, T moments_taa_w) {
        const T SIGMA_Z = T(1.f), SIGMA_N = T(128.f), EPS = T(1e-5f);
        const T wavelet_h[3] = {T(float(3. / 8)), T(float(1. / 4)), T(float(1. / 16))};
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int64_t py = 0; py < int64_t(H); py++)
            for (int64_t px = 0; px < int64_t(W); px++) {
                size_t tid = size_t(py) * W + size_t(px);
                const TemporalMap<T> &ti = svgf_temporal_map[tid];
                T illum = s.illum[tid];
                T illum2 = illum * illum;
                T miu1 = T(0.f), miu2 = T(0.f);
                if (ti.frame_count < 4) {
                    T depth_p = s.depth[tid];
                    Vec2<T> dg = s.depth_gradient[tid];
                    Vec3<T> normal_p = s.normal[tid];
                    T weight = T(0.f);
                    for (int i = -2; i <= 2; i++)
                        for (int j = -2; j <= 2; j++) {
                            int64_t qx = px + i, qy = py + j;
                            if (qx < 0 || qx >= int64_t(W) || qy < 0 || qy >= int64_t(H))
                                continue;
                            size_t q = size_t(qy) * W + size_t(qx);
                            T h_value = wavelet_h[std::abs(i)] * wavelet_h[std::abs(j)];
                            T depth_q = s.depth[q];
                            Vec2<T> dp = mk2(T(float(i)), T(float(j)));
                            // `exp(-abs(..) / (..))`: unary minus on T is formed in float, so the quotient and exp are float
                            T w_z = T(exp(-float(abs(depth_p - depth_q)) / float(SIGMA_Z * abs(dot_product(dg, dp) + EPS))));
                            T w_n = pow(tmax(T(0.f), dot_product(normal_p, s.normal[q])), SIGMA_N);
                            T w_value = w_z * w_n;
                            T hw = h_value * w_value;
                            if (isfinite(hw)) {
                                T illum_q = s.illum[q];
                                if (isfinite(illum_q)) {
                                    miu1 += hw * illum_q;
                                    miu2 += hw * illum_q * illum_q;
                                    weight += hw;
                                }
                            }
                        }
                    miu1 /= weight;
                    miu2 /= weight;
                } else {
                    miu1 = (T(1.f) - moments_taa_w) * sample_linear1(ti, s.miu1_1, miu1) + illum * moments_taa_w;
                    miu2 = (T(1.f) - moments_taa_w) * sample_linear1(ti, s.miu2_1, miu2) + illum2 * moments_taa_w;
                    if (!isfinite(miu1))
                        miu1 = illum;
                    if (!isfinite(miu2))
                        miu2 = illum2;
                }
                s.miu1_2[tid] = miu1;
                s.miu2_2[tid] = miu2;
                s.var_1[tid] = miu2 - miu1 * miu1;
            }
    }
    void wavelet_filter(int stride, const std::vector<Vec3<T>> &color_in, std::vector<Vec3<T>> &color_out,
                        const std::vector<T> &var_in, std::vector<T> &var_out, const SVGF &s) {
        const T SIGMA_Z = T(1.f), SIGMA_N = T(128.f), SIGMA_L = T(4.f), EPS = T(1e-5f);
        const T wavelet_h[3] = {T(float(3. / 8)), T(float(1. / 4)), T(float(1. / 16))};
        const T gaussian_g[2] = {T(float(1. / 2)), T(float(1. / 4))};
#pragma omp parallel for schedule(static) num_threads(threads)
        for (int64_t py = 0; py < int64_t(H); py++)
            for (int64_t px = 0; px < int64_t(W); px++) {
                size_t tid = size_t(py) * W + size_t(px);
                T sg = T(0.f), gw = T(0.f);
                for (int i = -1; i <= 1; i++)
                    for (int j = -1; j <= 1; j++) {
                        int64_t qx = px + i, qy = py + j;
                        if (qx < 0 || qx >= int64_t(W) || qy < 0 || qy >= int64_t(H))
                            continue;
                        T g = gaussian_g[std::abs(i)] * gaussian_g[std::abs(j)];
                        sg += g * var_in[size_t(qy) * W + size_t(qx)];
                        gw += g;
                    }
                sg = sqrt(sg / gw);
                Vec3<T> normal_p = s.normal[tid];
                T next_var = T(0.f);
                Vec3<T> next_color;
                T wsum_var = T(0.f), wsum_color = T(0.f);
                T depth_p = s.depth[tid], illum_p = s.illum[tid];
                Vec2<T> dg = s.depth_gradient[tid];
                for (int i = -2; i <= 2; i++)
                    for (int j = -2; j <= 2; j++) {
                        int64_t qx = px + stride * i, qy = py + stride * j;
                        if (qx < 0 || qx >= int64_t(W) || qy < 0 || qy >= int64_t(H))
                            continue;
                        size_t q = size_t(qy) * W + size_t(qx);
                        T h_value = wavelet_h[std::abs(i)] * wavelet_h[std::abs(j)];
                        T depth_q = s.depth[q];
                        Vec2<T> dp = mk2(T(float(stride * i)), T(float(stride * j)));
                        T w_z = T(exp(-float(abs(depth_p - depth_q)) / float(SIGMA_Z * abs(dot_product(dg, dp) + EPS))));
                        T w_n = pow(tmax(T(0.f), dot_product(normal_p, s.normal[q])), SIGMA_N);
                        T illum_q = s.illum[q];
                        T w_l = T(exp(-float(abs(illum_p - illum_q)) / float(SIGMA_L * sg + EPS)));
                        T w_value = w_z * w_n * w_l;
                        T hw = h_value * w_value;
                        const Vec3<T> &color_q = color_in[q];
                        T var_q = var_in[q];
                        if (isfinite(hw)) {
                            if (isfinite(var_q)) {
                                next_var += hw * hw * var_q;
                                wsum_var += hw;
                            }
                            if (isfinite(color_q[0]) && isfinite(color_q[1]) && isfinite(color_q[2])) {
                                next_color += hw * color_q;
                                wsum_color += hw;
                            }
                        }
                    }
                next_var /= (wsum_var * wsum_var);
                next_color /= wsum_color;
                value_or(next_color, color_in[tid]);
                if (!isfinite(next_var))
                    next_var = var_in[tid];
                color_out[tid] = next_color;
                var_out[tid] = next_var;
            }
    }
    void svgf_denoise(std::vector<Vec3<T>> &color_inout, SVGF &s, T taa_color_w, T taa_moments_w) {
        preprocess_normal_depth(s);
        temporal_accumulation_color(color_inout, s, taa_color_w);
        temporal_accumulation_moments(s, taa_moments_w);
        s.miu1_1 = s.miu1_2; // DEVIATION Q8: full copy (the reference copies W*H bytes only)
        s.miu2_1 = s.miu2_2;
        wavelet_filter(1, color_inout, s.color_history, s.var_1, s.var_2, s);
        wavelet_filter(2, s.color_history, s.color_buffer, s.var_2, s.var_1, s);
        wavelet_filter(4, s.color_buffer, color_inout, s.var_1, s.var_2, s);
        wavelet_filter(8, color_inout, s.color_buffer, s.var_2, s.var_1, s);
        wavelet_filter(16, s.color_buffer, color_inout, s.var_1, s.var_2, s);
    }

    // ------------------------------------------------------------------------------------------
    // F: Renderer::render = render_gbuffer + render_cuda (imgui_window.cu:121-347), plus the
    // build-side spp loop.
    // ------------------------------------------------------------------------------------------
    struct Timer {
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        double lap() {
            auto t1 = std::chrono::steady_clock::now();
            double s = std::chrono::duration<double>(t1 - t0).count();
            t0 = t1;
            return s;
        }
    };
    void render_frame(const lprt_frame_settings &fs) {
        const size_t N = size_t(W) * H;
        const uint32_t spp = fs.spp ? fs.spp : 1;
        Timer tm;
        fill_gbuffer();
        stage_seconds[LPRT_STAGE_PRIMARY] += tm.lap();
        auto seed_of = [&](uint32_t s, int which) -> uint64_t {
            if (fs.seeds)
                return fs.seeds[3 * s + uint32_t(which)];
            return 3 * (frame_index * spp + s) + 1 + uint64_t(which);
        };
        generate_temporal_map(seed_of(0, 0));
        stage_seconds[LPRT_STAGE_TEMPORAL_MAP] += tm.lap();

        std::vector<float> acc_color(3 * N, 0.f), acc_col(3 * N, 0.f), acc_wht(3 * N, 0.f);
        for (uint32_t s = 0; s < spp; s++) {
            // the five clears of imgui_window.cu:136-140
            std::fill(intensity.begin(), intensity.end(), Vec3<T>());
            std::fill(di_intensity.begin(), di_intensity.end(), Vec3<T>());
            std::fill(shade_commands.begin(), shade_commands.end(), PixelShaderInput<T>());
            std::fill(gi_valid.begin(), gi_valid.end(), uint8_t(0));
            std::fill(albedo.begin(), albedo.end(), Vec3<T>());
            std::fill(mul_gi_colored.begin(), mul_gi_colored.end(), Vec3<T>());
            std::fill(mul_gi_white.begin(), mul_gi_white.end(), Vec3<T>());
            // gi_material_multiplier is never cleared by the reference: stale values persist across frames
            tm.lap();
            if (fs.gi_on)
                shade<true, false>(0, seed_of(s, 1));
            else
                shade<true, true>(0, seed_of(s, 1));
            stage_seconds[LPRT_STAGE_SHADE1] += tm.lap();
            dbg_gi_direction.assign(N, Vec4<T>());
            for (size_t i = 0; i < N; i++)
                if (gi_valid[i]) {
                    const Vec3<T> &d = trace_gi_commands[i].direction;
                    dbg_gi_direction[i] = mk4(d[0], d[1], d[2], T(1.f));
                }
            tm.lap();
            trace_di_light();
            stage_seconds[LPRT_STAGE_TRACE_DI1] += tm.lap();
            accumulate_di_light(0);
            if (fs.gi_on) {
                std::fill(di_intensity.begin(), di_intensity.end(), Vec3<T>());
                tm.lap();
                trace_gi();
                stage_seconds[LPRT_STAGE_TRACE_GI] += tm.lap();
                shade<false, true>(1, seed_of(s, 2));
                stage_seconds[LPRT_STAGE_SHADE2] += tm.lap();
                std::fill(shade_commands.begin(), shade_commands.end(), PixelShaderInput<T>());
                std::fill(gi_valid.begin(), gi_valid.end(), uint8_t(0));
                trace_di_light();
                stage_seconds[LPRT_STAGE_TRACE_DI2] += tm.lap();
                accumulate_di_light(1);
            }
            write_clean_color(fs.add_direct_out != 0);
            for (size_t i = 0; i < N; i++)
                for (int k = 0; k < 3; k++) {
                    acc_color[3 * i + size_t(k)] += float(color_inprogress[i][k]);
                    acc_col[3 * i + size_t(k)] += float(mul_gi_colored[i][k]);
                    acc_wht[3 * i + size_t(k)] += float(mul_gi_white[i][k]);
                }
            stage_seconds[LPRT_STAGE_COMPOSE] += tm.lap();
        }
        // spp mean: float accumulation in sample order, one division, one rounding to T (spp = 1 is exact)
        for (size_t i = 0; i < N; i++)
            for (int k = 0; k < 3; k++) {
                color_inprogress[i][k] = T(acc_color[3 * i + size_t(k)] / float(spp));
                mul_gi_colored[i][k] = T(acc_col[3 * i + size_t(k)] / float(spp));
                mul_gi_white[i][k] = T(acc_wht[3 * i + size_t(k)] / float(spp));
            }
        raw_color = color_inprogress;
        raw_gi_colored = mul_gi_colored;
        raw_gi_white = mul_gi_white;
        tm.lap();
        if (fs.svgf) {
            svgf_denoise(mul_gi_colored, svgf_colored, T(fs.svgf_color_mix_weight), T(fs.svgf_moments_mix_weight));
            svgf_denoise(mul_gi_white, svgf_white, T(fs.svgf_color_mix_weight), T(fs.svgf_moments_mix_weight));
        }
        // add_denoised_color (cuda.hpp:2201-2228)
        for (size_t i = 0; i < N; i++) {
            Vec3<T> alb = fs.demodulate ? mk3(T(1.f), T(1.f), T(1.f)) : albedo[i];
            if (fs.add_gi_colored)
                color_inprogress[i] += elemul(mul_gi_colored[i], alb);
            if (fs.add_gi_white)
                color_inprogress[i] += mul_gi_white[i];
        }
        stage_seconds[LPRT_STAGE_SVGF_WAVELET] += tm.lap(); // whole SVGF + add_denoised_color
        // temporal_anti_aliasing (cuda.hpp:2418-2439) + write_to_surface2d (cuda.hpp:2441-2455)
        T taa_w = T(fs.taa_mix_weight);
        for (size_t i = 0; i < N; i++) {
            Vec3<T> in_color = color_inprogress[i];
            Vec3<T> out_color = sample_linear3(taa_temporal_map[i], taa_history_color, color_inprogress[i]);
            value_or(out_color, in_color);
            color_inprogress[i] = out_color * (T(1.f) - taa_w) + in_color * taa_w;
            for (int k = 0; k < 3; k++)
                output[i][k] = std::pow(float(color_inprogress[i][k]), 1 / 2.2f);
            output[i][3] = 1.f;
        }
        taa_history_color = color_inprogress;
        stage_seconds[LPRT_STAGE_FINALIZE] += tm.lap();
        // copy_last_frame_pixel_id / copy_last_frame_transform (cuda.hpp:2230-2263)
        for (size_t i = 0; i < N; i++) {
            if (g_color[i][3] == 0) {
                last_frame_id[i].object_id = UINT32_MAX;
            } else {
                last_frame_id[i].object_id = objects[g_objid[i]].objectid;
                last_frame_id[i].inobject_offset = g_offset[i];
            }
        }
        for (const auto &oc : objects)
            last_frame_L2W[oc.objectid] = oc.transform_L2W;
        last_frame_transform_W2C = transform_W2C;
        stage_seconds[LPRT_STAGE_BACKUP] += tm.lap();
        frame_index++;
    }

    void reset_history() {
        frame_index = 0;
        size_t n = size_t(W) * H;
        svgf_temporal_map.assign(n, TemporalMap<T>());
        taa_temporal_map.assign(n, TemporalMap<T>());
        last_frame_id.assign(n, LastFramePixelID());
        taa_history_color.assign(n, Vec3<T>());
        svgf_colored.resize(n);
        svgf_white.resize(n);
        std::fill(gi_material_multiplier.begin(), gi_material_multiplier.end(), Vec3<T>());
    }
};

} // namespace orc
#endif
