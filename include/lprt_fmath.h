/* Single-precision transcendentals whose results SHAPE RAYS, written out so that every implementation of the path
 * computes the same bits.
 *
 * Where they are used (reference call sites):
 *   - Box-Muller of curand_normal: sqrtf(-2 logf(u)) * {sinf, cosf}(v)   rt/rtrt/cuda.hpp:1927-1928, 1967-1969
 *   - tan(fovy / 2) of the primary rays                                   rt/rtrt/cuda.hpp:803-805
 * The reference takes these from CUDA's libdevice (2-4 ulp by its documentation, and curand_normal itself uses the
 * fast intrinsics), so no implementation is "the" reference value: parity with the CUDA path is unpinned here either
 * way.  What CAN be pinned is that the HIP kernels and the CPU restatement cast bit-identical rays:
 * before this header the device's logf/sinf/cosf and the host's libm differed in the last bit for ~0.3 % of the GI
 * directions, and every such ray showed up as an "other ray" mismatch in the ray-by-ray parity tests.
 *
 * The functions use only IEEE-754 single-precision + - * / (each correctly rounded), fmaf (one rounding), rintf under
 * round-to-nearest-even, and integer operations on the bit pattern -- operations that give identical results on the
 * host and on gfx950.  BOTH translation units must be compiled with -ffp-contract=off (they are: csrc/Makefile,
 * oracle/Makefile) so that no other multiply-add is fused.
 *
 * Included by the product (csrc/shade.hpp) and by the test-infrastructure oracle (oracle/core.hpp, oracle/half.hpp).
 * Accuracy (tests/test_fmath.py, against double-precision libm): logf <= 0.85 ulp on (0, 1], sinf/cosf <= 1.6 ulp
 * on [0, 2 pi] (0.8 ulp of 1 absolute), tanf <= 3.2 ulp on (0, 1.5]; CUDA documents 1 / 2 / 4 ulp for the functions the reference calls.
 */
#ifndef LPRT_FMATH_H
#define LPRT_FMATH_H
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define LPRT_FM __host__ __device__ static inline
#else
#define LPRT_FM static inline
#endif

LPRT_FM uint32_t lprt_fm_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, sizeof u);
    return u;
}
LPRT_FM float lprt_fm_float(uint32_t u) {
    float f;
    memcpy(&f, &u, sizeof f);
    return f;
}

/* ln(x) for finite x > 0 (normal or subnormal).  x = 2^k * m with m in [sqrt(1/2), sqrt(2)); ln m through
 * s = f / (2 + f), f = m - 1: ln(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)), the classic fdlibm decomposition with a degree-4
 * even polynomial for R; k ln2 is added in two parts so that the sum keeps its low bits. */
LPRT_FM float lprt_logf(float x) {
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float L1 = 6.6666662693e-01f, L2 = 4.0000972152e-01f, L3 = 2.8498786688e-01f, L4 = 2.4279078841e-01f;
    uint32_t ix = lprt_fm_bits(x);
    int k = 0;
    if (ix < 0x00800000u) { /* subnormal: scale by 2^25 */
        x *= 33554432.0f;
        ix = lprt_fm_bits(x);
        k = -25;
    }
    ix += 0x3f800000u - 0x3f3504f3u; /* move the mantissa break to sqrt(2)/2 */
    k += (int)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    const float m = lprt_fm_float(ix);
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    const float w = z * z;
    const float t1 = w * (L2 + w * L4);
    const float t2 = z * (L1 + w * L3);
    const float R = t2 + t1;
    const float hfsq = 0.5f * f * f;
    const float dk = (float)k;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* sin and cos of a finite |x| <= 64.  k = nearest integer to x * 2/pi; r = x - k * pi/2 with pi/2 in two parts through
 * fmaf (k * hi is exact inside the fma); minimax polynomials on [-pi/4, pi/4]; quadrant by k mod 4. */
LPRT_FM void lprt_sincosf(float x, float *sn, float *cs) {
    const float two_over_pi = 6.3661977237e-01f;
    const float pio2_hi = 1.5707963705062866e+00f; /* (float)(pi/2) */
    const float pio2_lo = -4.3711390001862412e-08f; /* pi/2 - pio2_hi */
    const float kf = __builtin_rintf(x * two_over_pi);
    float r = __builtin_fmaf(kf, -pio2_hi, x);
    r = __builtin_fmaf(kf, -pio2_lo, r);
    const float z = r * r;
    /* sin r = r + r z (S1 + z (S2 + z S3)),  cos r = 1 - z/2 + z^2 (C1 + z (C2 + z C3)) */
    const float S1 = -1.6666654611e-01f, S2 = 8.3321608736e-03f, S3 = -1.9515295891e-04f;
    const float C1 = 4.1666645683e-02f, C2 = -1.3887316255e-03f, C3 = 2.4433157117e-05f;
    const float ps = r + r * z * (S1 + z * (S2 + z * S3));
    const float pc = 1.0f - 0.5f * z + z * z * (C1 + z * (C2 + z * C3));
    const int q = (int)kf & 3;
    const float s0 = (q & 1) ? pc : ps;
    const float c0 = (q & 1) ? ps : pc;
    *sn = (q & 2) ? -s0 : s0;
    *cs = ((q + 1) & 2) ? -c0 : c0;
}
LPRT_FM float lprt_sinf(float x) {
    float s, c;
    lprt_sincosf(x, &s, &c);
    return s;
}
LPRT_FM float lprt_cosf(float x) {
    float s, c;
    lprt_sincosf(x, &s, &c);
    return c;
}
/* tan as the quotient of the two (one correctly rounded division) */
LPRT_FM float lprt_tanf(float x) {
    float s, c;
    lprt_sincosf(x, &s, &c);
    return s / c;
}

#endif /* LPRT_FMATH_H */
