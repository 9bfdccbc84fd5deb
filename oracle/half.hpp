// TEST INFRASTRUCTURE — CPU oracle, never shipped or measured as the product.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything under oracle/.
//
// Software IEEE binary16 with one round-to-nearest-even per operation.
// Restates `float16` of the reference (rt/math/number.hpp:7-155): implicit construction from
// float, implicit conversion to float, + - * / and comparisons defined on (half, half) only, and
// tan/pow/abs/sqrt going half -> float -> libm -> half.  Because the interface is the same, C++
// overload resolution gives mixed expressions (float * half, int - half, -half ...) the same types
// the reference gets.
//
// Each arithmetic result is computed in float and rounded once to half; this is correctly rounded
// because float carries 24 >= 2*11+2 significand bits (double rounding is innocuous for + - * /
// and sqrt).  PINNED against the reference's vendored third_party/half/half.hpp by
// oracle/ref_half_check.cpp (built into oracle/_ref/, see oracle/Makefile).
#ifndef ORC_HALF_HPP
#define ORC_HALF_HPP

#include <cmath>
#include <cstdint>
#include <cstring>
#include "../include/lprt_fmath.h" // the transcendentals that shape rays, shared with the device code (see its header)

namespace orc {

inline uint16_t f32_to_f16_bits(float f) {
    uint32_t x;
    std::memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) { // inf / nan
        if (ax > 0x7f800000u)
            return uint16_t(sign | 0x7e00u | ((ax >> 13) & 0x3ffu)); // quiet nan
        return uint16_t(sign | 0x7c00u);
    }
    if (ax >= 0x477ff000u) // rounds to >= 65520 -> inf
        return uint16_t(sign | 0x7c00u);
    if (ax < 0x33000001u) // <= 2^-25: rounds to zero (exactly 2^-25 ties to even = 0)
        return uint16_t(sign);
    int32_t exp = int32_t(ax >> 23) - 127;
    uint32_t man = (ax & 0x7fffffu) | 0x800000u;
    if (exp < -14) { // subnormal half
        int shift = -14 - exp + 13; // bits to drop from the 24-bit significand
        uint32_t kept = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (kept & 1u)))
            kept++;
        return uint16_t(sign | kept);
    }
    uint32_t kept = man >> 13; // includes the implicit bit
    uint32_t rem = man & 0x1fffu;
    uint32_t h = (uint32_t(exp + 15) << 10) + (kept & 0x3ffu);
    if (rem > 0x1000u || (rem == 0x1000u && (kept & 1u)))
        h++; // carries into the exponent correctly
    return uint16_t(sign | h);
}

inline float f16_bits_to_f32(uint16_t h) {
    uint32_t sign = uint32_t(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t x;
    if (exp == 0) {
        if (man == 0) {
            x = sign;
        } else { // subnormal: normalise
            int e = -1;
            do {
                man <<= 1;
                e++;
            } while (!(man & 0x400u));
            x = sign | (uint32_t(127 - 15 - e) << 23) | ((man & 0x3ffu) << 13);
        }
    } else if (exp == 31) {
        x = sign | 0x7f800000u | (man << 13);
    } else {
        x = sign | ((exp + 112u) << 23) | (man << 13);
    }
    float f;
    std::memcpy(&f, &x, 4);
    return f;
}

struct half {
    uint16_t bits;
    half() = default;
    half(float v) : bits(f32_to_f16_bits(v)) {}  // number.hpp:16
    operator float() const { return f16_bits_to_f32(bits); } // number.hpp:18

    half &operator+=(half o) { return *this = half(float(*this) + float(o)); }
    half &operator-=(half o) { return *this = half(float(*this) - float(o)); }
    half &operator*=(half o) { return *this = half(float(*this) * float(o)); }
    half &operator/=(half o) { return *this = half(float(*this) / float(o)); }
    half operator+(half o) const { return half(float(*this) + float(o)); }
    half operator-(half o) const { return half(float(*this) - float(o)); }
    half operator*(half o) const { return half(float(*this) * float(o)); }
    half operator/(half o) const { return half(float(*this) / float(o)); }
    bool operator<(half o) const { return float(*this) < float(o); }
    bool operator>(half o) const { return float(*this) > float(o); }
    bool operator<=(half o) const { return float(*this) <= float(o); }
    bool operator>=(half o) const { return float(*this) >= float(o); }
    bool operator==(half o) const { return float(*this) == float(o); }
    bool operator!=(half o) const { return float(*this) != float(o); }
};

// number.hpp:132-155
inline half tan(half v) { return half(lprt_tanf(float(v))); } // only use: tan(fovy / 2) of the primary rays
inline half pow(half v, half p) { return half(std::pow(float(v), float(p))); }
inline half abs(half v) {
    half r;
    r.bits = uint16_t(v.bits & 0x7fffu);
    return r;
}
inline half sqrt(half v) { return half(std::sqrt(float(v))); }
inline bool isfinite(half v) { return (v.bits & 0x7c00u) != 0x7c00u; }
inline bool isnan(half v) { return (v.bits & 0x7c00u) == 0x7c00u && (v.bits & 0x3ffu) != 0; }

// float counterparts with the names the pipeline code uses unqualified (CUDA device overloads
// resolve tan/pow/abs/sqrt/exp on float to the single-precision functions).
inline float tan(float v) { return lprt_tanf(v); }
inline float pow(float v, float p) { return std::pow(v, p); }
inline float abs(float v) { return std::fabs(v); }
inline float sqrt(float v) { return std::sqrt(v); }
inline float exp(float v) { return std::exp(v); }
inline bool isfinite(float v) { return std::isfinite(v); }
inline bool isnan(float v) { return std::isnan(v); }

} // namespace orc
#endif
