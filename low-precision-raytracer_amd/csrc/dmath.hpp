// Device-side scalar/vector arithmetic for the two DataT instantiations (float, packed half).
//
// Every helper spells out the operation ORDER and the ROUNDING points of the reference's
// Matrix<M,N,DataT> / float16 (rt/math/matrix.hpp, rt/math/number.hpp) so that the kernels agree with
// the CPU oracle to the last few ulps: one rounding per + - * /, dot products accumulated left to
// right from zero, normalisation as "multiply by the rounded reciprocal of the norm".
// The translation unit must be compiled with -ffp-contract=off (no FMA contraction).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/lprt_fmath.h"

namespace lprt {

typedef _Float16 h16;

#define LPRT_DEV __device__ __forceinline__
#define LPRT_HD __host__ __device__ __forceinline__

// ---- scalar traits ------------------------------------------------------------------------------
template <typename T> struct Num;
template <> struct Num<float> {
    static constexpr bool is_half = false;
    static LPRT_DEV float from_f(float v) { return v; }
    static LPRT_DEV float div(float a, float b) { return a / b; } // IEEE (hipcc default: correctly rounded f32 divide)
    static LPRT_DEV float sqrt(float a) { return __builtin_sqrtf(a); }
    static LPRT_DEV float abs(float a) { return __builtin_fabsf(a); }
    static LPRT_DEV bool finite(float a) { return __builtin_isfinite(a); }
    static LPRT_DEV bool isnan(float a) { return a != a; }
    static LPRT_DEV float inf() { return __builtin_inff(); }
    static LPRT_DEV float nan() { return __builtin_nanf(""); }
};
template <> struct Num<h16> {
    static constexpr bool is_half = true;
    static LPRT_DEV h16 from_f(float v) { return (h16)v; } // v_cvt_f16_f32, round to nearest even
    // half quotient formed in f32 and rounded once (== correctly rounded half division, 24 >= 2*11+2)
    static LPRT_DEV h16 div(h16 a, h16 b) { return (h16)((float)a / (float)b); }
    static LPRT_DEV h16 sqrt(h16 a) { return (h16)__builtin_sqrtf((float)a); } // number.hpp:141
    static LPRT_DEV h16 abs(h16 a) { return __builtin_fabsf16(a); }
    static LPRT_DEV bool finite(h16 a) { return __builtin_isfinite((float)a); }
    static LPRT_DEV bool isnan(h16 a) { return a != a; }
    static LPRT_DEV h16 inf() { return (h16)__builtin_inff(); }
    static LPRT_DEV h16 nan() { return (h16)__builtin_nanf(""); }
};

template <typename T> LPRT_DEV T cvt(float v) { return Num<T>::from_f(v); }
template <typename T> LPRT_DEV T tdiv(T a, T b) { return Num<T>::div(a, b); }
template <typename T> LPRT_DEV T tsqrt(T a) { return Num<T>::sqrt(a); }
template <typename T> LPRT_DEV T tabs(T a) { return Num<T>::abs(a); }
template <typename T> LPRT_DEV bool tfinite(T a) { return Num<T>::finite(a); }
template <typename T> LPRT_DEV bool tisnan(T a) { return Num<T>::isnan(a); }
template <typename T> LPRT_DEV T tmax(T a, T b) { return a > b ? a : b; } // rt::max, number.hpp:161
template <typename T> LPRT_DEV T tmin(T a, T b) { return a < b ? a : b; }
// unary minus of a float16 goes through float in the reference (no operator-): exact either way
template <typename T> LPRT_DEV T tneg(T a) { return -a; }
// pow / exp / tan: half -> float -> libm -> half (number.hpp:132-133)
template <typename T> LPRT_DEV T tpow(T a, T p) { return cvt<T>(::powf((float)a, (float)p)); }
// pow(x, 2) and pow(x, 5) of the BRDFs (bsdf.hpp:29, 77, 126) as products in f32, rounded to DataT once like the reference's
// DataT(powf(...)).  The device powf is a ~170-instruction extended-precision log/exp (it does not know its exponent is a
// small integer) and shading called it twice per light: 40 % of k_shade's vector instructions.  x*x is the correctly rounded
// square; ((x*x)*(x*x))*x is within 1.5 ulp of x^5 -- CUDA documents 4 ulp for the powf the reference calls -- and both
// agree with powf on 0, 1, inf and NaN.
template <typename T> LPRT_DEV T tpow2(T a) {
    const float f = (float)a;
    return cvt<T>(f * f);
}
template <typename T> LPRT_DEV T tpow5(T a) {
    const float f = (float)a, f2 = f * f;
    return cvt<T>((f2 * f2) * f);
}
// tan(fovy / 2) shapes every primary ray: include/lprt_fmath.h's form, shared with the CPU restatement
template <typename T> LPRT_DEV T ttan(T a) { return cvt<T>(lprt_tanf((float)a)); }
// reciprocal used by Matrix::operator/=(x): DataT(1.0 / double(x)) == correctly rounded 1/x
template <typename T> LPRT_DEV T trcp(T a) { return cvt<T>(1.0f / (float)a); }
// (int) of a float as the hardware converts it (NaN -> 0, saturating), cuda.hpp:2302
LPRT_DEV int f2i(float f) { return (int)f; }

// ---- vectors ------------------------------------------------------------------------------------
template <typename T> struct V2 {
    T x, y;
};
template <typename T> struct V3 {
    T x, y, z;
};
template <typename T> struct alignas(sizeof(T) * 4) V4 {
    T x, y, z, w;
};

template <typename T> LPRT_DEV V3<T> mk3(T a, T b, T c) { return V3<T>{a, b, c}; }
template <typename T> LPRT_DEV V4<T> mk4(T a, T b, T c, T d) { return V4<T>{a, b, c, d}; }
template <typename T> LPRT_DEV V3<T> zero3() { return V3<T>{(T)0, (T)0, (T)0}; }
template <typename T> LPRT_DEV V3<T> xyz(const V4<T> &v) { return V3<T>{v.x, v.y, v.z}; }

template <typename T> LPRT_DEV V3<T> operator+(V3<T> a, V3<T> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <typename T> LPRT_DEV V3<T> operator-(V3<T> a, V3<T> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <typename T> LPRT_DEV V3<T> operator*(V3<T> a, T k) { return {a.x * k, a.y * k, a.z * k}; }
template <typename T> LPRT_DEV V3<T> operator*(T k, V3<T> a) { return {a.x * k, a.y * k, a.z * k}; } // k*M == M*k, matrix.hpp:702
template <typename T> LPRT_DEV V3<T> neg(V3<T> a) { return a * (T)(-1); }                            // matrix.hpp:717-720
template <typename T> LPRT_DEV V3<T> elemul(V3<T> a, V3<T> b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
template <typename T> LPRT_DEV V3<T> elediv(V3<T> a, V3<T> b) { return {tdiv(a.x, b.x), tdiv(a.y, b.y), tdiv(a.z, b.z)}; }
template <typename T> LPRT_DEV V2<T> operator+(V2<T> a, V2<T> b) { return {a.x + b.x, a.y + b.y}; }
template <typename T> LPRT_DEV V2<T> operator*(T k, V2<T> a) { return {a.x * k, a.y * k}; }

// dot_product: `sum{0}; sum += a[i]*b[i]` (matrix.hpp:805-812)
template <typename T> LPRT_DEV T dot(V3<T> a, V3<T> b) {
    T s = (T)0;
    s = s + a.x * b.x;
    s = s + a.y * b.y;
    s = s + a.z * b.z;
    return s;
}
template <typename T> LPRT_DEV T dot(V2<T> a, V2<T> b) {
    T s = (T)0;
    s = s + a.x * b.x;
    s = s + a.y * b.y;
    return s;
}
template <typename T> LPRT_DEV T norm2_squared(V3<T> a) { return dot(a, a); } // matrix.hpp:306-314
template <typename T> LPRT_DEV T norm2(V3<T> a) { return tsqrt(norm2_squared(a)); }
// normalize(): `*= DataT(1.0 / norm2())` (matrix.hpp:318-334)
template <typename T> LPRT_DEV V3<T> normalized(V3<T> a) { return a * trcp(norm2(a)); }
// Matrix / scalar: multiply by the rounded reciprocal (matrix.hpp:318-324)
template <typename T> LPRT_DEV V3<T> vdiv(V3<T> a, T k) { return a * trcp(k); }
template <typename T> LPRT_DEV V3<T> cross(V3<T> a, V3<T> b) { // matrix.hpp:796-803
    return {a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
// symmetric_vector (matrix.hpp:633-641): the dot product here has NO leading zero
template <typename T> LPRT_DEV V3<T> reflect_about(V3<T> v, V3<T> n) {
    T d = v.x * n.x + v.y * n.y + v.z * n.z;
    return (d * (T)2) * n - v;
}
// lerp (cuda.hpp:944-948)
template <typename T> LPRT_DEV V3<T> lerp3(V3<T> b, V3<T> v1, V3<T> v2, V3<T> v3) { return b.x * v1 + b.y * v2 + b.z * v3; }
template <typename T> LPRT_DEV V2<T> lerp2(V3<T> b, V2<T> v1, V2<T> v2, V2<T> v3) { return b.x * v1 + b.y * v2 + b.z * v3; }
template <typename T> LPRT_DEV void value_or(V3<T> &io, V3<T> d) { // cuda.hpp:1168-1177
    if (!tfinite(io.x))
        io.x = d.x;
    if (!tfinite(io.y))
        io.y = d.y;
    if (!tfinite(io.z))
        io.z = d.z;
}
template <typename V> LPRT_DEV V clip_both(V lo, V hi, V val) { return val < lo ? lo : (val > hi ? hi : val); } // cuda.hpp:881-888

// ---- 4x4 row-major matrix (Mat4<DataT>) -----------------------------------------------------------
template <typename T> struct M4 {
    T m[16];
};
// Matrix*Vector: zero accumulator, inner index in the middle loop (matrix.hpp:209-220)
template <typename T> LPRT_DEV V4<T> mul(const M4<T> &a, V4<T> v) {
    V4<T> r;
    T s;
    s = (T)0; s = s + a.m[0] * v.x; s = s + a.m[1] * v.y; s = s + a.m[2] * v.z; s = s + a.m[3] * v.w; r.x = s;
    s = (T)0; s = s + a.m[4] * v.x; s = s + a.m[5] * v.y; s = s + a.m[6] * v.z; s = s + a.m[7] * v.w; r.y = s;
    s = (T)0; s = s + a.m[8] * v.x; s = s + a.m[9] * v.y; s = s + a.m[10] * v.z; s = s + a.m[11] * v.w; r.z = s;
    s = (T)0; s = s + a.m[12] * v.x; s = s + a.m[13] * v.y; s = s + a.m[14] * v.z; s = s + a.m[15] * v.w; r.w = s;
    return r;
}
template <typename T> LPRT_DEV V3<T> xform_dir(const M4<T> &a, V3<T> d) { return xyz(mul(a, mk4(d.x, d.y, d.z, (T)0))); }
template <typename T> LPRT_DEV V3<T> xform_pos(const M4<T> &a, V3<T> p) { // to_vec3_as_pos: divide by w (matrix.hpp:599-605)
    V4<T> r = mul(a, mk4(p.x, p.y, p.z, (T)1));
    if (r.w == (T)1) // affine transforms: x / 1 == x exactly, skip three IEEE divisions
        return {r.x, r.y, r.z};
    return {tdiv(r.x, r.w), tdiv(r.y, r.w), tdiv(r.z, r.w)};
}

} // namespace lprt
