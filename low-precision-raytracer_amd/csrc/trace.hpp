// Device ray casting: two-level traversal + the reference's mixed-precision triangle test.
//
// Replaces Ray/Intersection, ray_aabb_intersection_{scene,object}, ray_triangle_intersection,
// search_intersection_in_object, scene_bvh.traversal and cast_ray (rt/rtrt/cuda.hpp:38-149, 193-354,
// 415-631).  What is kept bit-for-bit is the ARITHMETIC that decides hits (slab tests with the same
// slack constants, the precomputed-transform triangle test with its fp32 t and fp32 edge-zone re-test,
// ray transform without renormalisation).  What is NOT kept is the reference's stackless
// parent-pointer walk over 1-primitive leaves: traversal here runs over an implicitly indexed tree (heap
// slots, depth-bounded; types.hpp: BlasNode) with the boxes AND leaf words of both children stored in the
// parent, near-child-first, with a one-bit-per-level "pending sibling" mask instead of a stack (no LDS, no
// scratch).  Closest-hit results are order independent except at exact t ties.
#pragma once
#include "types.hpp"

namespace lprt {

template <typename T> LPRT_DEV float ray_moveforward_t() { return Num<T>::is_half ? 1e-1f : 1e-4f; } // cuda.hpp:934-942

// Ray::transformed (cuda.hpp:42-47): position divided by w, direction NOT renormalised.
template <typename T> LPRT_DEV RayT<T> transform_ray(const RayT<T> &r, const M4<T> &m) {
    RayT<T> o;
    o.o = xform_pos(m, r.o);
    o.d = xform_dir(m, r.d);
    return o;
}

// ---- T2: ray_aabb_intersection_scene / _object (cuda.hpp:70-149) ------------------------------------------------------
// Per-ray constants of the slab tests.  The differences (bound - origin) are formed in DataT, as the reference's float16
// subtraction does; from there on the test runs in f32 for BOTH instantiations: quotient as a product with the IEEE f32
// reciprocal of the direction, entry = max over the axes, exit = min, the reference's slack constants.  (Round 1 rounded
// every half slab parameter back to half, as the reference does, and then needed a repair path for the parameters that
// overflow 65504 -- every ray with a direction component below ~5e-4 against boxes a few metres away; f32 parameters of
// half boxes cannot overflow, the repair is gone and a box test is 29 instructions instead of ~65.)
template <typename T> struct RaySlab {
    V3<T> o;
#ifdef LPRT_REFSLAB_TU
    V3<T> d; // the reference-arithmetic variant divides by the direction itself (below)
#endif
    float ix, iy, iz;
    uint32_t slow;   // bit k: 1/d_k is not finite (the ray does not move along axis k): the tests take the general form below
    uint32_t par;    // = slow when the stationary-axis rejection is on (SceneDev::axis_reject), else 0
    uint32_t reject; // 0: the reference's plain test only (A/B switch)
};
template <typename T> LPRT_DEV RaySlab<T> make_slab(const RayT<T> &r, uint32_t reject) {
#ifdef LPRT_REFSLAB_TU
    RaySlab<T> s{r.o, r.d, 1.0f / (float)r.d.x, 1.0f / (float)r.d.y, 1.0f / (float)r.d.z, 0u, 0u, reject};
#else
    RaySlab<T> s{r.o, 1.0f / (float)r.d.x, 1.0f / (float)r.d.y, 1.0f / (float)r.d.z, 0u, 0u, reject};
#endif
    s.slow = (__builtin_isfinite(s.ix) ? 0u : 1u) | (__builtin_isfinite(s.iy) ? 0u : 2u) | (__builtin_isfinite(s.iz) ? 0u : 4u);
    s.par = reject ? s.slow : 0u;
    return s;
}
// A ray that does not move along an axis can only hit what its origin lies inside of on that axis.  The reference's slab
// test IGNORES such an axis (below), which is conservative and costs nothing as long as those rays are not cast; but
// they are: Box-Muller returns the pair (0, 0) when its uniform rounds to 1.0f (2^-25 of the draws), so every few
// frames one GI ray is exactly axis-aligned, passes every box in front of it and walks the whole BLAS (131 k nodes and
// 91 k triangle tests for ONE ray of the 262 k-triangle atrium: 136 ms).  Boxes whose extent on a stationary axis does
// not contain the origin, with a margin far wider than the triangle test's tolerance zones (1/8 of the box + 2^-7
// relative), are rejected here; the triangles the ray can hit are all still visited, so the result is unchanged.
template <typename T> LPRT_DEV bool stationary_axis_outside(float o, T lo, T hi) {
    const float l = (float)lo, h = (float)hi;
    const float m = (h - l) * 0.125f + (__builtin_fabsf(l) + __builtin_fabsf(h)) * 0.0078125f + 1e-6f;
    return o < l - m || o > h + m;
}
template <typename T> LPRT_DEV bool stationary_reject(const T *lo, const T *hi, const RaySlab<T> &rs) {
    return ((rs.par & 1u) && stationary_axis_outside((float)rs.o.x, lo[0], hi[0])) ||
           ((rs.par & 2u) && stationary_axis_outside((float)rs.o.y, lo[1], hi[1])) ||
           ((rs.par & 4u) && stationary_axis_outside((float)rs.o.z, lo[2], hi[2]));
}
// (bound - origin) / d: the difference in DataT, the quotient as a product with the f32 reciprocal.  Half: the product is
// written as fma(diff, inv, +0) so that the conversion folds into one v_fma_mix_f32 (same value; a -0 product becomes +0,
// which no comparison of the test can tell apart).
template <typename T> LPRT_DEV float slab_t(T bound, T origin, float inv);
template <> LPRT_DEV float slab_t<float>(float bound, float origin, float inv) { return (bound - origin) * inv; }
template <> LPRT_DEV float slab_t<h16>(h16 bound, h16 origin, float inv) { return __builtin_fmaf((float)(h16)(bound - origin), inv, 0.0f); }

// a0*b0 + a1*b1 + a2*b2 in f32 from DataT operands, summed left to right (row 2 of the triangle test, cuda.hpp:234-240).
// Half: a product of two halves is exact in f32, so fma(a, b, s) rounds once exactly like the separate add -- and is one
// v_fma_mix_f32 reading the halves directly instead of two conversions, a multiply and an add.
template <typename T> LPRT_DEV float dot3_f32(T a0, T a1, T a2, T b0, T b1, T b2);
template <> LPRT_DEV float dot3_f32<float>(float a0, float a1, float a2, float b0, float b1, float b2) { return a0 * b0 + a1 * b1 + a2 * b2; }
template <> LPRT_DEV float dot3_f32<h16>(h16 a0, h16 a1, h16 a2, h16 b0, h16 b1, h16 b2) {
    return __builtin_fmaf((float)a2, (float)b2, __builtin_fmaf((float)a1, (float)b1, (float)a0 * (float)b0));
}

#ifdef LPRT_REFSLAB_TU
// ---- the reference's half slab tests, operation for operation (namespace refslab, LPRT_OPT_REFERENCE_SLAB) ------------
// ray_aabb_intersection_scene / _object (cuda.hpp:70-149) with DataT = float16 (number.hpp:20-86): t1 = (lower - source) /
// direction and t2 = (upper - source) / direction are a half subtraction and a half division each (one rounding per
// operation), the pairs are ordered, axes with a non-finite t1 or t2 are ignored, entry = max / exit = min over the rest,
// and the slack expressions `t2min * DataT(1.001953f)` / `t2min + DataT(0.02f)` are half operations as well.  A parameter
// beyond 65504 is +-inf in half and takes its axis out of the test, exactly as in the reference.
// Two rejections are added that cannot change a result, only the time a hopeless ray takes (they reject boxes that hold
// no triangle the ray's triangle tests could accept): the stationary-axis rejection of the default variant, and for axes
// lost to OVERFLOW the round-1 repair -- the lost axis is re-evaluated in f32 against the box widened by 1/8 of its extent.
template <bool SCENE> LPRT_DEV bool slab_test_reference_half(const h16 *lo, const h16 *hi, const RaySlab<h16> &rs, float &tmn, float &tmx) {
    const h16 d[3] = {rs.d.x, rs.d.y, rs.d.z}, o[3] = {rs.o.x, rs.o.y, rs.o.z};
    h16 t1max = (h16)0, t2min = (h16)0;
    bool updated = false, ok[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        h16 a = tdiv<h16>((h16)(lo[k] - o[k]), d[k]), b = tdiv<h16>((h16)(hi[k] - o[k]), d[k]);
        if (b < a) { // swap_to_ordered
            const h16 c = a;
            a = b;
            b = c;
        }
        ok[k] = tfinite(a) && tfinite(b);
        if (!ok[k])
            continue;
        if (!updated || t1max < a)
            t1max = a;
        if (!updated || b < t2min)
            t2min = b;
        updated = true;
    }
    tmn = (float)t1max;
    tmx = (float)t2min;
    if (!updated)
        return false;
    if (rs.par && stationary_reject(lo, hi, rs))
        return false;
    if (rs.reject && !(ok[0] & ok[1] & ok[2])) {
        const float inv[3] = {rs.ix, rs.iy, rs.iz};
        float tn = tmn, tf = tmx;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (ok[k] || ((rs.slow >> k) & 1u))
                continue; // kept by the half test, or stationary (handled above)
            const float l = (float)lo[k], h = (float)hi[k];
            const float m = (h - l) * 0.125f + (__builtin_fabsf(l) + __builtin_fabsf(h)) * 0.0078125f + 1e-6f;
            const float ta = (l - m - (float)o[k]) * inv[k], tb = (h + m - (float)o[k]) * inv[k];
            tn = __builtin_fmaxf(tn, __builtin_fminf(ta, tb));
            tf = __builtin_fminf(tf, __builtin_fmaxf(ta, tb));
        }
        if (tf < -1e-3f || tn > tf * 1.002f + 0.02f)
            return false;
    }
    if (SCENE) {
        const h16 s = t2min + (h16)0.02f;
        return t1max <= s && 0.0f <= (float)s;
    }
    return t1max <= (h16)(t2min * (h16)1.001953f) && 0.0f <= (float)t2min;
}
#endif

// min / max of the slab parameters.  Half instantiation: the parameters come out of v_fma_mix_f32 / v_pk_fma_f32, which LLVM does
// not know to produce canonical values, so every operand of a __builtin_fminf / fmaxf got a `v_max_f32 x, x, x` in front of
// it -- six per box test, a quarter of its vector instructions, in kernels whose time is their vector-instruction count
// (DESIGN.md 9).  The hardware instructions themselves already ARE minnum / maxnum for everything arithmetic can produce
// (a quiet NaN operand yields the other one; signalling NaNs do not come out of arithmetic), so the half instantiation
// names them directly.  Single precision keeps the builtins (its parameters come from v_sub / v_mul: known canonical, and
// the compiler may fold modifiers into them).
template <typename T> struct SlabOps {
    static LPRT_DEV float mn(float a, float b) { return __builtin_fminf(a, b); }
    static LPRT_DEV float mx(float a, float b) { return __builtin_fmaxf(a, b); }
    static LPRT_DEV float mx3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
    static LPRT_DEV float mn3(float a, float b, float c) { return __builtin_fminf(__builtin_fminf(a, b), c); }
};
#ifndef LPRT_NO_HW_MINMAX
template <> struct SlabOps<h16> {
    static LPRT_DEV float mn(float a, float b) {
        float r;
        asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static LPRT_DEV float mx(float a, float b) {
        float r;
        asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
        return r;
    }
    static LPRT_DEV float mx3(float a, float b, float c) {
        float r;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
    }
    static LPRT_DEV float mn3(float a, float b, float c) {
        float r;
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
        return r;
    }
};
#endif

// All six parameters finite (every ray that moves along all three axes: a product with a finite reciprocal can overflow to
// +-inf, consistently signed, but never to NaN): entry = max3(min, min, min), exit = min3(max, max, max).
template <typename T, bool SCENE> LPRT_DEV bool slab_test_moving(const T *lo, const T *hi, const RaySlab<T> &rs, float &tmn, float &tmx) {
#ifdef LPRT_REFSLAB_TU
    if constexpr (Num<T>::is_half)
        return slab_test_reference_half<SCENE>(lo, hi, rs, tmn, tmx);
#endif
    const float a0 = slab_t(lo[0], rs.o.x, rs.ix), b0 = slab_t(hi[0], rs.o.x, rs.ix);
    const float a1 = slab_t(lo[1], rs.o.y, rs.iy), b1 = slab_t(hi[1], rs.o.y, rs.iy);
    const float a2 = slab_t(lo[2], rs.o.z, rs.iz), b2 = slab_t(hi[2], rs.o.z, rs.iz);
    typedef SlabOps<T> O;
    const float t1max = O::mx3(O::mn(a0, b0), O::mn(a1, b1), O::mn(a2, b2));
    const float t2min = O::mn3(O::mx(a0, b0), O::mx(a1, b1), O::mx(a2, b2));
    tmn = t1max;
    tmx = t2min;
    if (SCENE) {
        const float s = t2min + 0.02f;
        return t1max <= s && 0.0f <= s;
    }
    return t1max <= t2min * 1.001953f && 0.0f <= t2min;
}
// The object-level box test with the walk's culls folded in, decision for decision:
//     slab (entry <= exit * 1.001953  &&  0 <= exit)  &&  entry < best_t  &&  entry < max_d  &&  exit > min_d
//  =  entry <= exit * 1.001953  &&  entry < tcull  &&  exit > mfloor,      tcull = min(best_t, max_d),
//     mfloor = min_d if min_d >= 0, else the largest negative float (exit > mfloor <=> 0 <= exit; and exit > min_d >= 0 implies it)
// -- three comparisons per box instead of five (the compiler cannot know the sign of min_d).
LPRT_DEV float cull_floor(float min_d) { return min_d >= 0.0f ? min_d : -1.4e-45f; }
template <typename T> LPRT_DEV bool box_hit_moving(const T *lo, const T *hi, const RaySlab<T> &rs, float tcull, float mfloor, float &tnear) {
    const float a0 = slab_t(lo[0], rs.o.x, rs.ix), b0 = slab_t(hi[0], rs.o.x, rs.ix);
    const float a1 = slab_t(lo[1], rs.o.y, rs.iy), b1 = slab_t(hi[1], rs.o.y, rs.iy);
    const float a2 = slab_t(lo[2], rs.o.z, rs.iz), b2 = slab_t(hi[2], rs.o.z, rs.iz);
    typedef SlabOps<T> O;
    const float t1max = O::mx3(O::mn(a0, b0), O::mn(a1, b1), O::mn(a2, b2));
    const float t2min = O::mn3(O::mx(a0, b0), O::mx(a1, b1), O::mx(a2, b2));
    tnear = t1max;
    return t1max <= t2min * 1.001953f && t1max < tcull && t2min > mfloor;
}

// General form: axes whose entry/exit is not finite are ignored (parallel rays), as in the reference; entry = max over the
// remaining axes, exit = min (the reference's sequential update computes exactly these extrema).
template <typename T, bool SCENE> LPRT_DEV bool slab_test_general(const T *lo, const T *hi, const RaySlab<T> &rs, float &tmn, float &tmx) {
#ifdef LPRT_REFSLAB_TU
    if constexpr (Num<T>::is_half)
        return slab_test_reference_half<SCENE>(lo, hi, rs, tmn, tmx);
#endif
    const float a0 = slab_t(lo[0], rs.o.x, rs.ix), b0 = slab_t(hi[0], rs.o.x, rs.ix);
    const float a1 = slab_t(lo[1], rs.o.y, rs.iy), b1 = slab_t(hi[1], rs.o.y, rs.iy);
    const float a2 = slab_t(lo[2], rs.o.z, rs.iz), b2 = slab_t(hi[2], rs.o.z, rs.iz);
    const bool ok0 = __builtin_isfinite(a0) && __builtin_isfinite(b0), ok1 = __builtin_isfinite(a1) && __builtin_isfinite(b1),
               ok2 = __builtin_isfinite(a2) && __builtin_isfinite(b2);
    const float ninf = -__builtin_inff(), pinf = __builtin_inff();
    const float n0 = ok0 ? (a0 < b0 ? a0 : b0) : ninf, f0 = ok0 ? (a0 > b0 ? a0 : b0) : pinf;
    const float n1 = ok1 ? (a1 < b1 ? a1 : b1) : ninf, f1 = ok1 ? (a1 > b1 ? a1 : b1) : pinf;
    const float n2 = ok2 ? (a2 < b2 ? a2 : b2) : ninf, f2 = ok2 ? (a2 > b2 ? a2 : b2) : pinf;
    const float m01 = n0 > n1 ? n0 : n1, t1max = m01 > n2 ? m01 : n2;
    const float x01 = f0 < f1 ? f0 : f1, t2min = x01 < f2 ? x01 : f2;
    tmn = t1max;
    tmx = t2min;
    if (!(ok0 | ok1 | ok2))
        return false;
    if (rs.par && stationary_reject(lo, hi, rs))
        return false;
    if (SCENE) {
        const float s = t2min + 0.02f;
        return t1max <= s && 0.0f <= s;
    }
    return t1max <= t2min * 1.001953f && 0.0f <= t2min;
}
// PAR is kept as a template argument of the callers' instantiations (rays with a stationary axis take their own copy of the
// walk): PAR = false promises rs.slow == 0.
template <typename T, bool SCENE, bool PAR = false> LPRT_DEV bool slab_test(const T *lo, const T *hi, const RaySlab<T> &rs, float &tmn, float &tmx) {
    if (PAR)
        return slab_test_general<T, SCENE>(lo, hi, rs, tmn, tmx);
    return slab_test_moving<T, SCENE>(lo, hi, rs, tmn, tmx);
}

// a BLAS box against a ray inside traverse_blas: the folded test for rays that move along all axes, the general form otherwise
template <typename T, bool PAR> LPRT_DEV bool blas_box(const T *lo, const T *hi, const RaySlab<T> &rs, float best_t, float min_d, float max_d, float &tnear) {
#ifndef LPRT_REFSLAB_TU
    // (half only: in the single-precision per-pixel kernels, which sit at their register budget, the folded form measured
    // SLOWER -- Cornell k_trace_gi 0.647 -> 0.676 ms, primary rays 0.071 -> 0.092 -- while the half kernels gained 3-4 %)
    if (!PAR && Num<T>::is_half)
        return box_hit_moving<T>(lo, hi, rs, SlabOps<T>::mn(best_t, max_d), cull_floor(min_d), tnear);
#endif
    float x;
    return slab_test<T, false, PAR>(lo, hi, rs, tnear, x) && tnear < best_t && tnear < max_d && x > min_d;
}

// world-box / TLAS-node test: one code path per ray kind, chosen at run time (the scene level is a few tests per ray)
template <typename T> LPRT_DEV bool slab_test_scene(const T *lo, const T *hi, const RaySlab<T> &rs, float &tmn, float &tmx) {
    if (rs.slow)
        return slab_test_general<T, true>(lo, hi, rs, tmn, tmx);
    return slab_test_moving<T, true>(lo, hi, rs, tmn, tmx);
}

// The same fold for the scene-level test (slack + 0.02, no cull by the closest hit, cuda.hpp:98, 536-538):
//     entry <= exit + 0.02  &&  0 <= exit + 0.02  &&  entry < max_d  &&  exit > min_d
// `0 <= exit + 0.02f` is `exit >= -0.02f` exactly (the sum of two floats that close is exact), so with
// sfloor = min_d if min_d >= -0.02f, else the float just below -0.02f, the last three comparisons are two.
LPRT_DEV float scene_floor(float min_d) { return min_d >= -0.02f ? min_d : __builtin_bit_cast(float, 0xbca3d70bu); } // the float just below -0.02f (0xbca3d70a)
template <typename T> LPRT_DEV bool scene_box(const T *lo, const T *hi, const RaySlab<T> &rs, float min_d, float max_d, float sfloor, float &tnear) {
    float x;
#ifndef LPRT_REFSLAB_TU
    if (Num<T>::is_half && !rs.slow) { // (half only, as blas_box)
        const float a0 = slab_t(lo[0], rs.o.x, rs.ix), b0 = slab_t(hi[0], rs.o.x, rs.ix);
        const float a1 = slab_t(lo[1], rs.o.y, rs.iy), b1 = slab_t(hi[1], rs.o.y, rs.iy);
        const float a2 = slab_t(lo[2], rs.o.z, rs.iz), b2 = slab_t(hi[2], rs.o.z, rs.iz);
        typedef SlabOps<T> O;
        const float t1max = O::mx3(O::mn(a0, b0), O::mn(a1, b1), O::mn(a2, b2));
        const float t2min = O::mn3(O::mx(a0, b0), O::mx(a1, b1), O::mx(a2, b2));
        tnear = t1max;
        return t1max <= t2min + 0.02f && t1max < max_d && t2min > sfloor;
    }
#endif
    return slab_test_scene(lo, hi, rs, tnear, x) && tnear < max_d && x > min_d;
}

template <bool COUNT> struct Cnt {
    unsigned rays = 0, tlas_nodes = 0, aabbw = 0, objects = 0, blas_nodes = 0, tris = 0, fallbacks = 0;
    unsigned wave_steps[3] = {0, 0, 0}, lane_steps[3] = {0, 0, 0}; // per phase: 0 instance entry, 1 node step, 2 leaf
    // called by every lane that takes part in a pass through phase `ph`
    LPRT_DEV void step(int ph) {
        if (COUNT) {
            lane_steps[ph]++;
            if ((threadIdx.x & 63u) == (unsigned)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true)))
                wave_steps[ph]++;
        }
    }
};

// ---- T3: ray_triangle_intersection (cuda.hpp:193-354) ---------------------------------------------
template <typename T> LPRT_DEV void load_fallback(const SceneDev<T> &sc, uint32_t rec, const TriRec<T> &r, TriRecF32 &f);
template <> LPRT_DEV void load_fallback<float>(const SceneDev<float> &, uint32_t, const TriRec<float> &r, TriRecF32 &f) {
    // DataT = float: M_shift_f32 == M_shift and position_f32 == position
    f.m0[0] = r.m0[0]; f.m0[1] = r.m0[1]; f.m0[2] = r.m0[2]; f.v2x = r.v2x;
    f.m1[0] = r.m1[0]; f.m1[1] = r.m1[1]; f.m1[2] = r.m1[2]; f.v2y = r.v2y;
    f.m2[0] = r.m2[0]; f.m2[1] = r.m2[1]; f.m2[2] = r.m2[2]; f.v2z = r.v2z;
}
template <> LPRT_DEV void load_fallback<h16>(const SceneDev<h16> &sc, uint32_t rec, const TriRec<h16> &, TriRecF32 &f) {
    f = sc.tris_f32[rec];
}

// The whole (half) triangle record in ONE round trip: two vector loads issued back to back.  Left to itself the
// compiler fetches the record field by field where each is first needed -- the offset for the skip test, then row 2 for t,
// then (if t is in range) the rest: three dependent fetches per triangle in a loop that is bound by exactly that.  The empty
// asm statement makes every word of the record live at one point, so the loads cannot be sunk past the early exits.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4))); // a native 128-bit register tuple (HIP's uint4 is a class)
template <typename T> LPRT_DEV TriRec<T> load_tri(const TriRec<T> *p);
template <> LPRT_DEV TriRec<h16> load_tri<h16>(const TriRec<h16> *p) { // 32 B
    const u32x4 *src = reinterpret_cast<const u32x4 *>(p);
    u32x4 q0 = src[0], q1 = src[1];
    asm volatile("" : "+v"(q0), "+v"(q1));
    struct Raw {
        u32x4 a, b;
    } raw{q0, q1};
    return __builtin_bit_cast(TriRec<h16>, raw);
}
// single precision: measured slower with the forced form (Cornell 3.10 -> 3.17 ms, atrium 38.1 -> 38.4): the 64-byte record
// is four loads of which the early exits usually need two; the compiler's field-by-field order stays
template <> LPRT_DEV TriRec<float> load_tri<float>(const TriRec<float> *p) { return *p; }

// The head of an instance record in one round trip (see load_tri): entering an instance used to be a chain of dependent
// field fetches (triangle count -> flags -> root box -> ...), two to three per instance and ray.
template <typename T> LPRT_DEV ObjHead<T> load_obj_head(const ObjRec<T> *p);
template <> LPRT_DEV ObjHead<h16> load_obj_head<h16>(const ObjRec<h16> *p) { // 48 B
    const u32x4 *src = reinterpret_cast<const u32x4 *>(p);
    u32x4 q0 = src[0], q1 = src[1], q2 = src[2];
    asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2));
    struct Raw {
        u32x4 a, b, c;
    } raw{q0, q1, q2};
    return __builtin_bit_cast(ObjHead<h16>, raw);
}
template <> LPRT_DEV ObjHead<float> load_obj_head<float>(const ObjRec<float> *p) { // 64 B
    const u32x4 *src = reinterpret_cast<const u32x4 *>(p);
    u32x4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
    asm volatile("" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
    struct Raw {
        u32x4 a, b, c, d;
    } raw{q0, q1, q2, q3};
    return __builtin_bit_cast(ObjHead<float>, raw);
}

// skip_here / skip_off: SkipGeometry of the ray (cuda.hpp:446) -- the triangle a ray starts on is not tested
template <typename T, bool COUNT>
LPRT_DEV bool tri_test(const SceneDev<T> &sc, uint32_t rec, const RayT<T> &ray, uint32_t inscene_objid, Isect<T> &is,
                       T min_distance, T max_distance, Cnt<COUNT> &cnt, bool skip_here, uint32_t skip_off) {
    LPRT_CHECK_RET(sc, rec < sc.lim_tris, 201, rec, sc.lim_tris, false);
    const TriRec<T> r = load_tri(sc.tris + rec);
    if (skip_here && r.inobject_offset == skip_off)
        return false;
    if (COUNT)
        cnt.tris++;
    T O0 = ray.o.x - r.v2x, O1 = ray.o.y - r.v2y, O2 = ray.o.z - r.v2z;
    T D0 = ray.d.x, D1 = ray.d.y, D2 = ray.d.z;
    T Ox_0 = O0 * r.m0[0], Ox_1 = O1 * r.m0[1], Ox_2 = O2 * r.m0[2];
    T Dx_0 = D0 * r.m0[0], Dx_1 = D1 * r.m0[1], Dx_2 = D2 * r.m0[2];
    T Oy_0 = O0 * r.m1[0], Oy_1 = O1 * r.m1[1], Oy_2 = O2 * r.m1[2];
    T Dy_0 = D0 * r.m1[0], Dy_1 = D1 * r.m1[1], Dy_2 = D2 * r.m1[2];
    T Ox = Ox_0 + Ox_1 + Ox_2;
    T Dx = Dx_0 + Dx_1 + Dx_2;
    T Oy = Oy_0 + Oy_1 + Oy_2;
    T Dy = Dy_0 + Dy_1 + Dy_2;
    // row 2 and t always in fp32, from the DataT-rounded operands (cuda.hpp:234-240)
    float Oz = dot3_f32<T>(O0, O1, O2, r.m2[0], r.m2[1], r.m2[2]);
    float Dz = dot3_f32<T>(D0, D1, D2, r.m2[0], r.m2[1], r.m2[2]);
    float invDz = 1.0f / Dz;
    float t = -Oz * invDz;
    T t_Dx = cvt<T>(t * (float)Dx);
    T t_Dy = cvt<T>(t * (float)Dy);
    T u = Ox + t_Dx;
    T v = Oy + t_Dy;

    if (!(t > (float)min_distance && t < (float)is.t && t < (float)max_distance))
        return false;

    // running error bounds (cuda.hpp:247-261)
    const T delta1 = cvt<T>(0.0009765625f); // 2^-10
    const T delta2 = cvt<T>(0.00390625f);   // 2^-8
    T sOx = tabs(Ox_0) + tabs(Ox_1) + tabs(Ox_2);
    T sDx = tabs(Dx_0) + tabs(Dx_1) + tabs(Dx_2);
    T sOy = tabs(Oy_0) + tabs(Oy_1) + tabs(Oy_2);
    T sDy = tabs(Dy_0) + tabs(Dy_1) + tabs(Dy_2);
    T error_Ox = delta1 * sOx + delta2 * sOx;
    T error_Dx = delta1 * sDx + delta2 * sDx;
    T error_Oy = delta1 * sOy + delta2 * sOy;
    T error_Dy = delta1 * sDy + delta2 * sDy;
    T tT = cvt<T>(t);
    T error_u = (error_Ox + tT * error_Dx + delta1 * (tabs(Ox) + cvt<T>(3.f) * tabs(t_Dx))) * cvt<T>(0.2f);
    T error_v = (error_Oy + tT * error_Dy + delta1 * (tabs(Oy) + cvt<T>(3.f) * tabs(t_Dy))) * cvt<T>(0.2f);

    float w = 1.0f - (float)u - (float)v; // `1 - u - v` is evaluated in float for both DataT
    T zero = (T)0;
    if ((u >= tneg(error_u) && u <= zero) || (v >= tneg(error_v) && v <= zero) ||
        (w >= (float)tneg((T)(error_v + error_u)) && w <= 0.0f)) {
        // edge zone: recompute everything in fp32 with strict inequalities (cuda.hpp:267-328)
        if (COUNT)
            cnt.fallbacks++;
        TriRecF32 f;
        load_fallback<T>(sc, rec, r, f);
        float Of0 = (float)ray.o.x - f.v2x, Of1 = (float)ray.o.y - f.v2y, Of2 = (float)ray.o.z - f.v2z;
        float Df0 = (float)ray.d.x, Df1 = (float)ray.d.y, Df2 = (float)ray.d.z;
        float Oxf = Of0 * f.m0[0] + Of1 * f.m0[1] + Of2 * f.m0[2];
        float Dxf = Df0 * f.m0[0] + Df1 * f.m0[1] + Df2 * f.m0[2];
        float Oyf = Of0 * f.m1[0] + Of1 * f.m1[1] + Of2 * f.m1[2];
        float Dyf = Df0 * f.m1[0] + Df1 * f.m1[1] + Df2 * f.m1[2];
        float Ozf = Of0 * f.m2[0] + Of1 * f.m2[1] + Of2 * f.m2[2];
        float Dzf = Df0 * f.m2[0] + Df1 * f.m2[1] + Df2 * f.m2[2];
        float tf = -Ozf / Dzf;
        float uf = Oxf + tf * Dxf;
        float vf = Oyf + tf * Dyf;
        if (tf > (float)min_distance && tf < (float)is.t && tf < (float)max_distance) {
            if (uf > 0 && vf > 0 && uf + vf < 1) {
                is.hit = true;
                is.t = cvt<T>(tf);
                is.bary = mk3(cvt<T>(uf), cvt<T>(vf), cvt<T>(1 - uf - vf));
                is.inobject_offset = r.inobject_offset;
                is.inscene_objid = inscene_objid;
                return true;
            }
        }
        return false;
    }
    if (u > tneg(error_u)) {
        // `DataT(1.0 + error_u + error_v)`: double sum, then double -> float -> DataT (cuda.hpp:332)
        T lim = cvt<T>((float)(1.0 + (double)(float)error_u + (double)(float)error_v));
        if (v > tneg(error_v) && u + v < lim) {
            is.hit = true;
            is.t = tT;
            is.bary = mk3(u, v, cvt<T>(w));
            is.inobject_offset = r.inobject_offset;
            is.inscene_objid = inscene_objid;
            return true;
        }
    }
    return false;
}

// ---- T4: BLAS traversal (replaces search_intersection_in_object, cuda.hpp:415-504) -----------------
template <typename T, bool ANY, bool COUNT, bool PAR, bool WIDE>
LPRT_DEV void traverse_blas(const SceneDev<T> &sc, const RayT<T> &ray, const RaySlab<T> &rs, const ObjHead<T> &ob, uint32_t skip_vao,
                            uint32_t skip_off, Isect<T> &is, T min_distance, T max_distance, Cnt<COUNT> &cnt) {
    if (COUNT)
        cnt.objects++;
    cnt.step(0);
    if (ob.n_tris == 0)
        return;
    float tmn, tmx;
    const float min_d = (float)min_distance, max_d = (float)max_distance;
    if (COUNT)
        cnt.blas_nodes++;
    if (!(slab_test<T, false, PAR>(ob.root_lo, ob.root_hi, rs, tmn, tmx) && tmn < (float)is.t && tmn < max_d && tmx > min_d))
        return;
    const BlasNode<T> *inner = sc.blas_inner + ob.inner_base;
    const uint32_t D = ob.depth;
    const bool skip_here = (ob.vao_index == skip_vao);
    if (D <= 1) {
        // 1-3 triangles (a quad is two): the child boxes would cost more than the triangle tests they could save
        for (uint32_t k = 0; k < ob.n_tris; k++) {
            uint32_t rec = ob.tri_base + k;
            tri_test<T, COUNT>(sc, rec, ray, ob.inscene_objid, is, min_distance, max_distance, cnt, skip_here, skip_off);
            if (ANY && is.hit)
                return;
        }
        return;
    }
    // Two levels per step: the records of both children of node i are adjacent (heap order), and together they hold the four
    // GRANDCHILD boxes.  A grandchild box lies inside its parent's box and the slab test is monotone in the bounds, so a ray
    // that passes a grandchild box passes the child box too: the child boxes need not be tested.  One dependent fetch and
    // four box tests per two levels instead of two fetches and four to six tests; the deferred INNER grandchildren of a step
    // are four bits of a 64-bit mask (node 4i + j), popped deepest first; hit LEAF grandchildren hand over their triangles at
    // once, one range per child record (sibling leaves are adjacent in leaf order).  Works on every tree the builders make:
    // a child that is itself a leaf has a record too (its own box and leaf word in entry 0, nothing in entry 1:
    // build.hip), so the two records can be fetched without knowing what the children are.
    if (WIDE) {
        uint32_t i = 1, c = 0;
        unsigned long long pend = 0;
        bool node_live = true;
        for (;;) {
            uint32_t lfa = 0, lfb = 0;
            while (node_live && (lfa | lfb) == 0u) {
                cnt.step(1);
                uint32_t g = 0, first = 0; // hit INNER children of this step, the nearest of them
                if (c + 2 <= D) {
                    const BlasNode<T> ra = inner[2 * i - 1], rb = inner[2 * i];
                    if (COUNT)
                        cnt.blas_nodes += 4;
                    float n[4];
                    const float best_t = (float)is.t;
                    const bool h0 = blas_box<T, PAR>(ra.lo0, ra.hi0, rs, best_t, min_d, max_d, n[0]);
                    const bool h1 = blas_box<T, PAR>(ra.lo1, ra.hi1, rs, best_t, min_d, max_d, n[1]);
                    const bool h2 = blas_box<T, PAR>(rb.lo0, rb.hi0, rs, best_t, min_d, max_d, n[2]);
                    const bool h3 = blas_box<T, PAR>(rb.lo1, rb.hi1, rs, best_t, min_d, max_d, n[3]);
                    const bool l0 = ra.leaf[0] != LPRT_CHILD_INNER, l1 = ra.leaf[1] != LPRT_CHILD_INNER;
                    const bool l2 = rb.leaf[0] != LPRT_CHILD_INNER, l3 = rb.leaf[1] != LPRT_CHILD_INNER;
                    if (h0 && l0)
                        lfa = ra.leaf[0];
                    if (h1 && l1)
                        lfa = lfa ? lfa + (ra.leaf[1] & 7u) : ra.leaf[1];
                    if (h2 && l2)
                        lfb = rb.leaf[0];
                    if (h3 && l3)
                        lfb = lfb ? lfb + (rb.leaf[1] & 7u) : rb.leaf[1];
                    g = ((h0 && !l0) ? 1u : 0u) | ((h1 && !l1) ? 2u : 0u) | ((h2 && !l2) ? 4u : 0u) | ((h3 && !l3) ? 8u : 0u);
                    if (g) {
                        float best = __builtin_inff();
                        bool have = false;
#pragma unroll
                        for (uint32_t k = 0; k < 4; k++)
                            if (((g >> k) & 1u) && (!have || n[k] < best)) {
                                best = n[k];
                                first = k;
                                have = true;
                            }
                        pend |= (unsigned long long)(g & ~(1u << first)) << (2 * c); // bits 4*(c/2) ..: c is even here
                        i = 4 * i + first;
                        c += 2;
                    }
                } else { // one level left: the children are leaves (no node lies deeper than D)
                    const BlasNode<T> nd = inner[i - 1];
                    if (COUNT)
                        cnt.blas_nodes += 2;
                    float n0, n1;
                    const float best_t = (float)is.t;
                    const bool h0 = blas_box<T, PAR>(nd.lo0, nd.hi0, rs, best_t, min_d, max_d, n0);
                    const bool h1 = blas_box<T, PAR>(nd.lo1, nd.hi1, rs, best_t, min_d, max_d, n1);
                    if (h0 && nd.leaf[0] != LPRT_CHILD_INNER)
                        lfa = nd.leaf[0];
                    if (h1 && nd.leaf[1] != LPRT_CHILD_INNER)
                        lfa = lfa ? lfa + (nd.leaf[1] & 7u) : nd.leaf[1];
                }
                if (!g) { // nothing to descend into: the deepest deferred grandchild, if any
                    if (!pend) {
                        node_live = false;
                    } else {
                        const uint32_t bit = 63u - (uint32_t)__builtin_clzll(pend);
                        pend ^= 1ull << bit;
                        const uint32_t k2 = (bit >> 2) * 2, j = bit & 3u; // depth of the node the step started from, grandchild slot
                        i = ((i >> (c - k2)) << 2) | j;
                        c = k2 + 2;
                    }
                }
            }
            if (lfa | lfb)
                cnt.step(2);
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const uint32_t lf = q ? lfb : lfa;
                const uint32_t l = lf >> 3, r = l + (lf & 7u);
                for (uint32_t k = l; k < r; k++) {
                    const uint32_t rec = ob.tri_base + k;
                    tri_test<T, COUNT>(sc, rec, ray, ob.inscene_objid, is, min_distance, max_distance, cnt, skip_here, skip_off);
                    if (ANY && is.hit)
                        return;
                }
            }
            if (!node_live)
                return;
        }
    }
    // "while-while": every lane first walks inner nodes until it has triangles to test (or runs out of work), then the whole
    // wave tests triangles; the two code paths are not interleaved per iteration.  A record carries a LEAF WORD per child
    // (types.hpp: BlasNode): a hit child that is a leaf hands over its triangle range at once -- leaves sit at any depth, there
    // is no record to fetch for them and they are never deferred; when both children are hit leaves their ranges are adjacent
    // in leaf order and are tested as one.  Only INNER children are descended into (nearer first) or deferred (one pending bit
    // per level: the tree is implicitly indexed, see the header of this file).
    uint32_t i = 1, c = 0, pending = 0;
    bool node_live = true; // (i, c) is a node still to be visited
    for (;;) {
        uint32_t lf = 0; // (first << 3) | count of the triangles to test now; 0: none
        while (node_live && lf == 0u) {
            LPRT_CHECK(sc, ob.inner_base + i - 1 < sc.lim_inner, 203, ob.inner_base + i - 1, sc.lim_inner);
            const BlasNode<T> nd = inner[i - 1];
            if (COUNT)
                cnt.blas_nodes += 2;
            cnt.step(1);
            float n0, n1;
            const float best_t = (float)is.t;
            const bool h0 = blas_box<T, PAR>(nd.lo0, nd.hi0, rs, best_t, min_d, max_d, n0);
            const bool h1 = blas_box<T, PAR>(nd.lo1, nd.hi1, rs, best_t, min_d, max_d, n1);
            const bool l0 = nd.leaf[0] != LPRT_CHILD_INNER, l1 = nd.leaf[1] != LPRT_CHILD_INNER;
            if (h0 && l0)
                lf = nd.leaf[0];
            if (h1 && l1)
                lf = lf ? lf + (nd.leaf[1] & 7u) : nd.leaf[1];
            const bool g0 = h0 && !l0, g1 = h1 && !l1;
            if (g0 | g1) {
                const uint32_t first = (g0 && (!g1 || n0 <= n1)) ? 0u : 1u;
                if (g0 & g1)
                    pending |= 1u << (c + 1);
                i = 2 * i + first;
                c++;
            } else if (pending) {
                const uint32_t d = 31u - (uint32_t)__builtin_clz(pending);
                pending ^= 1u << d;
                i = (i >> (c - d)) ^ 1u;
                c = d;
            } else {
                node_live = false;
            }
        }
        if (lf) {
            cnt.step(2);
            const uint32_t l = lf >> 3, r = l + (lf & 7u);
            for (uint32_t k = l; k < r; k++) {
                const uint32_t rec = ob.tri_base + k;
                tri_test<T, COUNT>(sc, rec, ray, ob.inscene_objid, is, min_distance, max_distance, cnt, skip_here, skip_off);
                if (ANY && is.hit)
                    return;
            }
        }
        if (!node_live)
            return;
    }
}

// ---- T5: TLAS traversal + cast_ray (cuda.hpp:517-631) ---------------------------------------------
template <typename T, bool ANY, bool COUNT, bool WIDE = false>
LPRT_DEV Isect<T> cast_ray(const SceneDev<T> &sc, const RayT<T> &ray, uint32_t skip_vao, uint32_t skip_off, T min_distance,
                           T max_distance, Cnt<COUNT> &cnt) {
    Isect<T> is;
    is.hit = false;
    is.t = cvt<T>(1e5f); // +inf in half (cuda.hpp:54)
    is.inscene_objid = 0;
    is.inobject_offset = 0;
    is.bary = zero3<T>();
    if (COUNT)
        cnt.rays++;
    if (sc.n_objects == 0)
        return is;
    const RaySlab<T> rs = make_slab(ray, sc.axis_reject);
    const float min_d = (float)min_distance, max_d = (float)max_distance, sfloor = scene_floor(min_d);
    float tmn;
    if (sc.tlas_mode == 1u) {
        // Small scenes (the Cornell box has 8 instances): a tree over a handful of mutually overlapping boxes costs ~2n
        // slab tests per ray; scan the n world boxes instead.  The scan loop is wave-uniform, so the boxes arrive through
        // scalar loads, and the object phase starts with every lane's candidate list complete.
        uint32_t cand = 0;
        for (uint32_t k = 0; k < sc.n_objects; k++) {
            const ObjRec<T> *ob = sc.objrecs + k;
            if (COUNT)
                cnt.aabbw++;
            if (scene_box(ob->lo, ob->hi, rs, min_d, max_d, sfloor, tmn))
                cand |= 1u << k;
        }
        while (cand) {
            const uint32_t k = (uint32_t)__builtin_ctz(cand);
            cand &= cand - 1u;
            // (the per-pixel kernels read the record field by field, as the compiler orders it: the one-round-trip form of
            // load_obj_head measured slower here -- Cornell 3.11 -> 3.22 ms -- and faster in the ray streams, which use it)
            const ObjRec<T> ob = sc.objrecs[k];
            // with an identity W2L (ObjHead::pad[0], set by make_objrec) the transform would return the ray unchanged
            // (x*1 + y*0 + z*0 is exact): the instance is entered with the world ray and its reciprocals as they are
            RayT<T> lr = ray;
            RaySlab<T> lrs = rs;
            if (!(ob.pad[0] & 1u)) {
                lr = transform_ray(ray, ob.W2L);
                lrs = make_slab(lr, sc.axis_reject);
            }
            if (lrs.slow) // rare: the instantiation whose box tests ignore / check the stationary axes
                traverse_blas<T, ANY, COUNT, true, WIDE>(sc, lr, lrs, ob, skip_vao, skip_off, is, min_distance, max_distance, cnt);
            else
                traverse_blas<T, ANY, COUNT, false, WIDE>(sc, lr, lrs, ob, skip_vao, skip_off, is, min_distance, max_distance, cnt);
            if (ANY && is.hit)
                return is;
        }
        return is;
    }
    if (COUNT)
        cnt.tlas_nodes++;
    if (!scene_box(sc.tlas_root, sc.tlas_root + 3, rs, min_d, max_d, sfloor, tmn))
        return is;
    const uint32_t D = sc.tlas_depth;
    uint32_t i = 1, c = 0, pending = 0;
    for (;;) {
        if (c < D) {
            const InnerNode<T> nd = sc.tlas_inner[i - 1];
            if (COUNT)
                cnt.tlas_nodes += 2;
            float n0, n1;
            // no `tmin < intersection.t` cull at the scene level (cuda.hpp:536-538)
            bool h0 = scene_box(nd.lo0, nd.hi0, rs, min_d, max_d, sfloor, n0);
            bool h1 = scene_box(nd.lo1, nd.hi1, rs, min_d, max_d, sfloor, n1);
            if (h0 | h1) {
                uint32_t first = (h0 && (!h1 || n0 <= n1)) ? 0u : 1u;
                if (h0 & h1)
                    pending |= 1u << (c + 1);
                i = 2 * i + first;
                c++;
                continue;
            }
        } else {
            uint32_t lw = sc.tlas_leaf[i - (1u << D)];
            uint32_t first = lw >> 1, count = (lw & 1u) + 1u;
            for (uint32_t k = 0; k < count; k++) {
                LPRT_CHECK_RET(sc, first + k < sc.n_objects, 205, first + k, sc.n_objects, is);
                const ObjRec<T> *ob = sc.objrecs + first + k;
                if (COUNT)
                    cnt.aabbw++;
                if (scene_box(ob->lo, ob->hi, rs, min_d, max_d, sfloor, tmn)) {
                    const ObjRec<T> obv = *ob;
                    RayT<T> lr = ray;
                    RaySlab<T> lrs = rs;
                    if (!(obv.pad[0] & 1u)) {
                        lr = transform_ray(ray, obv.W2L);
                        lrs = make_slab(lr, sc.axis_reject);
                    }
                    if (lrs.slow)
                        traverse_blas<T, ANY, COUNT, true, WIDE>(sc, lr, lrs, obv, skip_vao, skip_off, is, min_distance, max_distance, cnt);
                    else
                        traverse_blas<T, ANY, COUNT, false, WIDE>(sc, lr, lrs, obv, skip_vao, skip_off, is, min_distance, max_distance, cnt);
                    if (ANY && is.hit)
                        return is;
                }
            }
        }
        if (!pending)
            return is;
        uint32_t d = 31u - (uint32_t)__builtin_clz(pending);
        pending ^= 1u << d;
        i = (i >> (c - d)) ^ 1u;
        c = d;
    }
}

// wave-aggregated counter flush: one atomic per counter per wave
template <bool COUNT> LPRT_DEV void flush_counters(const Cnt<COUNT> &c, Counters *out) {
    if (!COUNT)
        return;
    unsigned vals[13] = {c.rays, c.tlas_nodes, c.aabbw, c.objects, c.blas_nodes, c.tris, c.fallbacks, c.wave_steps[0], c.wave_steps[1], c.wave_steps[2],
                         c.lane_steps[0], c.lane_steps[1], c.lane_steps[2]};
    unsigned long long *dst = reinterpret_cast<unsigned long long *>(out);
#pragma unroll
    for (int k = 0; k < 13; k++) {
        unsigned v = vals[k];
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0 && v)
            atomicAdd(dst + k, (unsigned long long)v);
    }
}

} // namespace lprt
