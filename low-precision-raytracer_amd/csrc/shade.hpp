// Device BRDFs, importance sampling and the cuRAND-compatible XORWOW stepping used by shade().
// Restates rt/rtrt/bsdf.hpp:12-139 and rt/rtrt/cuda.hpp:1713-1749 with explicit rounding points.
#pragma once
#include "types.hpp"
#include "../../include/lprt_fmath.h"

namespace lprt {

#define LPRT_PI_D 3.14159265358979323846

// ---- XORWOW (cuRAND device API; third party, restated from its published algorithm) -------------------
struct Xorwow {
    uint32_t v0, v1, v2, v3, v4, d;
    int bm_flag;
    float bm_extra;
};
LPRT_DEV uint32_t xorwow_next(Xorwow &s) {
    uint32_t t = s.v0 ^ (s.v0 >> 2);
    s.v0 = s.v1;
    s.v1 = s.v2;
    s.v2 = s.v3;
    s.v3 = s.v4;
    s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v4 + s.d;
}
LPRT_DEV float xorwow_uniform(Xorwow &s) { // curand_uniform: (0, 1]
    return (float)xorwow_next(s) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}
LPRT_DEV float xorwow_normal(Xorwow &s) { // curand_normal: Box-Muller pair, second value cached
    if (s.bm_flag) {
        s.bm_flag = 0;
        return s.bm_extra;
    }
    uint32_t x = xorwow_next(s), y = xorwow_next(s);
    float u = (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
    float v = (float)y * 1.46291807e-09f + (1.46291807e-09f / 2.0f);
    // logf / sinf / cosf decide GI ray directions: the shared single-precision forms of include/lprt_fmath.h, so that the
    // CPU restatement casts the bit-identical ray (libdevice vs libm differed in the last bit for ~0.3 % of the directions)
    float r = __builtin_sqrtf(-2.0f * lprt_logf(u)), sn, cs;
    lprt_sincosf(v, &sn, &cs);
    s.bm_extra = r * cs;
    s.bm_flag = 1;
    return r * sn;
}

// ---- BRDF (bsdf.hpp) -----------------------------------------------------------------------------------
template <typename T> struct Brdf {
    T colored, white;
};
template <typename T> LPRT_DEV V3<T> get_brdf(Brdf<T> b, V3<T> base) { // bsdf.hpp:52-54
    V3<T> one = mk3((T)1, (T)1, (T)1);
    return base * b.colored + one * b.white;
}
template <typename T> LPRT_DEV T specular_brdf(V3<T> V, V3<T> H, V3<T> L, V3<T> N, T alpha) { // bsdf.hpp:12-33
    T HdotL = dot(H, L), HdotV = dot(H, V), NdotH = dot(N, H);
    T zero = (T)0;
    if (NdotH <= zero || HdotL <= zero || HdotV <= zero)
        return zero;
    T aNL = tabs(dot(N, L)), aNV = tabs(dot(N, V));
    T a2 = alpha * alpha;
    T one = (T)1;
    T div1 = aNL + tsqrt(tmax(zero, a2 + (one - a2) * aNL * aNL));
    T div2 = aNV + tsqrt(tmax(zero, a2 + (one - a2) * aNV * aNV));
    T D = tdiv(a2, cvt<T>((float)LPRT_PI_D) * tpow2(NdotH * NdotH * (a2 - one) + one));
    return tdiv(tdiv(D, div1), div2);
}
template <typename T> LPRT_DEV T fresnel_pow5(V3<T> V, V3<T> H) {
    T VdotH = dot(V, H);
    return tpow5(tmax((T)0, (T)1 - tabs(VdotH)));
}
template <typename T> LPRT_DEV Brdf<T> material_brdf(T metallic, T roughness, V3<T> V, V3<T> L, V3<T> N) { // bsdf.hpp:57-97
    T zero = (T)0, one = (T)1;
    if (dot(L, N) < zero)
        return {zero, zero};
    V3<T> H = normalized(L + V);
    T pow5 = fresnel_pow5(V, H);
    T alpha = roughness * roughness;
    T layer = specular_brdf(V, H, L, N, alpha);
    T f0 = cvt<T>(0.04f);
    T fr = f0 + cvt<T>(1.0f - (float)f0) * pow5; // DataT(1 - dielectric_f0): formed in float
    T dielectric_white = fr * layer;
    T dielectric_colored = (one - fr) * cvt<T>((float)(1.0 / LPRT_PI_D));
    T metal_white = layer * pow5;
    T metal_colored = layer * (one - pow5);
    return {metal_colored * metallic + dielectric_colored * (one - metallic),
            metal_white * metallic + dielectric_white * (one - metallic)};
}
template <typename T> LPRT_DEV Brdf<T> glassy_brdf(T metallic, V3<T> V, V3<T> L, V3<T> N) { // bsdf.hpp:106-139
    T zero = (T)0, one = (T)1;
    if (dot(L, N) < zero)
        return {zero, zero};
    V3<T> H = normalized(L + V);
    T pow5 = fresnel_pow5(V, H);
    T f0 = cvt<T>(0.04f);
    T fr = f0 + cvt<T>(1.0f - (float)f0) * pow5;
    T metal_colored = one - pow5;
    return {metal_colored * metallic + zero * (one - metallic), pow5 * metallic + fr * (one - metallic)};
}

// ---- sampling (cuda.hpp:1714-1734) -----------------------------------------------------------------------
template <typename T> LPRT_DEV V3<T> sample_ggx(T a2, Xorwow &st) { // Q9: x, y are r*u2, r*u3
    T rx = cvt<T>(xorwow_uniform(st)), ry = cvt<T>(xorwow_uniform(st)), rz = cvt<T>(xorwow_uniform(st));
    T one = (T)1;
    T z = tsqrt(tdiv(one - rx, one + a2 * rx - rx));
    T r = tsqrt(tmax((T)0, one - z * z));
    return mk3(r * ry, r * rz, z);
}
template <typename T> LPRT_DEV T pdf_ggx_reflect(T NH, T a2) {
    T tmp = (NH * a2 - NH) * NH + (T)1;
    // D_ggx_opti returns `M_PI * tmp * tmp`: double arithmetic, then double -> float -> DataT
    T D = cvt<T>((float)(LPRT_PI_D * (double)(float)tmp * (double)(float)tmp));
    return tdiv(NH * a2, D);
}

// ---- camera (CUDARenderGISettings, cuda.hpp:799-832) ---------------------------------------------------------
template <typename T> LPRT_DEV void primary_xy(const SceneDev<T> &sc, uint32_t W, uint32_t H, uint32_t sx, uint32_t sy, T &x, T &y) {
    T fx = cvt<T>((float)sx), fy = cvt<T>((float)sy), fw = cvt<T>((float)W), fh = cvt<T>((float)H);
    T two = (T)2, one = (T)1, half_ = cvt<T>(0.5f);
    T nx = tdiv(two * (fx + half_), fw) - one;
    T ny = tdiv(two * (fy + half_), fh) - one;
    T max_y = ttan(tdiv(sc.cam_fovy, two));
    y = ny * max_y;
    x = tdiv(nx * max_y * fw, fh); // omit_ar = true
}
template <typename T> LPRT_DEV RayT<T> primary_ray(const SceneDev<T> &sc, uint32_t W, uint32_t H, uint32_t sx, uint32_t sy) {
    T x, y;
    primary_xy(sc, W, H, sx, sy, x, y);
    RayT<T> r;
    r.o = zero3<T>();
    r.d = normalized(mk3(x, y, (T)(-1)));
    // ray.transformed(L2W) then direction.normalize()
    V3<T> o = xform_pos(sc.cam_L2W, r.o);
    V3<T> d = xform_dir(sc.cam_L2W, r.d);
    r.o = o;
    r.d = normalized(d);
    return r;
}
template <typename T> LPRT_DEV V3<T> primary_ray_direction(const SceneDev<T> &sc, uint32_t W, uint32_t H, uint32_t sx, uint32_t sy) {
    T x, y;
    primary_xy(sc, W, H, sx, sy, x, y);
    return normalized(xform_dir(sc.cam_L2W, mk3(x, y, (T)(-1))));
}


// ---- textures: RDTexture (memory.hpp:334-369) as the texture unit is documented to fetch it -------------------------
// normalised coordinates, wrap addressing (the reference leaves addressMode at 0), linear filter at xB = x - 0.5 with
// 1.8 fixed-point blend weights, 8-bit texels read as c/255 or through the sRGB table (alpha is never decoded).
LPRT_DEV void tex_wrap(float u, uint32_t n, int &i0, int &i1, float &alpha) {
    float f = u - ::floorf(u);
    if (!(f == f) || !(f < 1.f)) // NaN / inf coordinate: texel 0
        f = 0.f;
    const float xb = f * (float)n - 0.5f;
    const float fl = ::floorf(xb);
    alpha = ::rintf((xb - fl) * 256.f) * (1.f / 256.f);
    i0 = (int)fl;
    i1 = i0 + 1;
    if (i0 < 0)
        i0 += (int)n;
    if (i1 >= (int)n)
        i1 -= (int)n;
}
LPRT_DEV float4 tex_texel(const Texture &t, const float *lut, int i, int j) {
    const uint32_t c = t.texels[(size_t)j * t.width + (size_t)i];
    const uint32_t r = c & 255u, g = (c >> 8) & 255u, b = (c >> 16) & 255u, a = c >> 24;
    if (t.srgb)
        return make_float4(lut[r], lut[g], lut[b], (float)a / 255.f);
    return make_float4((float)r / 255.f, (float)g / 255.f, (float)b / 255.f, (float)a / 255.f);
}
LPRT_DEV float4 tex2d(const Texture *textures, const float *lut, uint32_t handle, float u, float v) {
    const Texture t = textures[handle];
    int i0, i1, j0, j1;
    float a, b;
    tex_wrap(u, t.width, i0, i1, a);
    tex_wrap(v, t.height, j0, j1, b);
    const float4 t00 = tex_texel(t, lut, i0, j0), t10 = tex_texel(t, lut, i1, j0), t01 = tex_texel(t, lut, i0, j1), t11 = tex_texel(t, lut, i1, j1);
    const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
    return make_float4(w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x, w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y,
                       w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z, w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w);
}
// direction_to_spherical (cuda.hpp:1742-1749) + the skybox fetch of shade() (cuda.hpp:1786-1791, 1828-1832)
template <typename T> LPRT_DEV V3<T> sample_skybox(const SceneDev<T> &sc, const V3<T> &dir) {
    float u = 0.1591f * ::atan2f((float)dir.y, (float)dir.x) + 0.5f + (float)sc.sky_dx;
    float v = 0.3183f * ::asinf((float)dir.z) + 0.5f + (float)sc.sky_dy;
    u = ::fmodf(u, 1.f);
    v = 1.f - ::fmodf(v, 1.f);
    const float4 c = tex2d(sc.textures, sc.srgb_lut, sc.sky_texture, u, v);
    return mk3(cvt<T>(c.x), cvt<T>(c.y), cvt<T>(c.z)) * sc.sky_exposure;
}

} // namespace lprt
