// TEST INFRASTRUCTURE — thin C entry points over the REFERENCE's vendored half-precision library
// (third_party/half/half.hpp, half 2.2.0), compiled from where it lies under /root/reference into
// oracle/_ref/libref_half.so (never copied into this repo).  It is the host-side arithmetic of the
// reference's float16 (rt/math/number.hpp:9-11, 24, 33, 42, 51, 60-86): used to pin oracle/half.hpp.
#include <half.hpp>
#include <cmath>
#include <cstdint>
#include <cstring>

using half_float::half;

static half from_bits(uint16_t b) {
    half h;
    std::memcpy(&h, &b, 2);
    return h;
}
static uint16_t to_bits(half h) {
    uint16_t b;
    std::memcpy(&b, &h, 2);
    return b;
}

extern "C" {
uint16_t ref_f32_to_f16(float f) { return to_bits(half(f)); }
float ref_f16_to_f32(uint16_t b) { return float(from_bits(b)); }
// op: 0 add, 1 sub, 2 mul, 3 div, 4 sqrt(a) computed as the reference does: sqrt(float) -> half
uint16_t ref_half_op(int op, uint16_t a, uint16_t b) {
    half x = from_bits(a), y = from_bits(b);
    switch (op) {
    case 0: return to_bits(x + y);
    case 1: return to_bits(x - y);
    case 2: return to_bits(x * y);
    case 3: return to_bits(x / y);
    default: return to_bits(half(std::sqrt(float(x))));
    }
}
}
