// TEST INFRASTRUCTURE — CPU oracle (see oracle/half.hpp header note).
//
// Small fixed-size vector/matrix algebra with exactly the operation order of the reference's
// Matrix<M,N,T> (rt/math/matrix.hpp), and a cuRAND-XORWOW-compatible generator.
#ifndef ORC_CORE_HPP
#define ORC_CORE_HPP

#include "half.hpp"
#include <array>
#include <cstdint>
#include <vector>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace orc {

template <typename T> inline T tmax(const T &a, const T &b) { return a > b ? a : b; } // number.hpp:161
template <typename T> inline T tmin(const T &a, const T &b) { return a < b ? a : b; } // number.hpp:158

// ---------------------------------------------------------------------------------------------
// Vec<N,T>: column vector. Every operator restates the loop order of matrix.hpp.
// ---------------------------------------------------------------------------------------------
template <int N, typename T> struct Vec {
    T d[N];
    Vec() {
        for (int i = 0; i < N; i++)
            d[i] = T(0.f); // `DataT data[M][N] = {}`, matrix.hpp:119
    }
    T &operator[](int i) { return d[i]; }
    const T &operator[](int i) const { return d[i]; }

    // matrix.hpp:222-235: per-element multiply by the scalar
    Vec &operator*=(T k) {
        for (int i = 0; i < N; i++)
            d[i] *= k;
        return *this;
    }
    Vec operator*(T k) const {
        Vec r = *this;
        r *= k;
        return r;
    }
    Vec &operator+=(const Vec &o) {
        for (int i = 0; i < N; i++)
            d[i] += o.d[i];
        return *this;
    }
    Vec operator+(const Vec &o) const {
        Vec r = *this;
        return r += o;
    }
    Vec &operator-=(const Vec &o) {
        for (int i = 0; i < N; i++)
            d[i] -= o.d[i];
        return *this;
    }
    Vec operator-(const Vec &o) const {
        Vec r = *this;
        return r -= o;
    }
    // matrix.hpp:318: `(*this) *= 1.0 / x` — the reciprocal is formed in double and rounded to T
    Vec &operator/=(T x) { return (*this) *= T(float(1.0 / double(float(x)))); }
    Vec operator/(T x) const {
        Vec r = *this;
        r /= x;
        return r;
    }
    // matrix.hpp:306-316
    T norm2_squared() const {
        T sum(0.f);
        for (int i = 0; i < N; i++)
            sum += d[i] * d[i];
        return sum;
    }
    T norm2() const { return sqrt(norm2_squared()); }
    Vec &normalize() { return (*this) /= norm2(); }
    Vec normalized() const {
        Vec r = *this;
        return r /= norm2();
    }
};

template <typename T> using Vec2 = Vec<2, T>;
template <typename T> using Vec3 = Vec<3, T>;
template <typename T> using Vec4 = Vec<4, T>;

template <typename T> inline Vec3<T> mk3(T a, T b, T c) {
    Vec3<T> v;
    v[0] = a;
    v[1] = b;
    v[2] = c;
    return v;
}
template <typename T> inline Vec4<T> mk4(T a, T b, T c, T w) {
    Vec4<T> v;
    v[0] = a;
    v[1] = b;
    v[2] = c;
    v[3] = w;
    return v;
}
template <typename T> inline Vec2<T> mk2(T a, T b) {
    Vec2<T> v;
    v[0] = a;
    v[1] = b;
    return v;
}

template <int N, typename T> inline Vec<N, T> operator*(T k, const Vec<N, T> &v) { return v * k; } // matrix.hpp:702-705
template <int N, typename T> inline Vec<N, T> operator-(const Vec<N, T> &v) { return v * T(-1.f); } // matrix.hpp:717-720

template <int N, typename T> inline Vec<N, T> elemul(const Vec<N, T> &a, const Vec<N, T> &b) {
    Vec<N, T> r;
    for (int i = 0; i < N; i++)
        r[i] = a[i] * b[i];
    return r;
}
template <int N, typename T> inline Vec<N, T> elediv(const Vec<N, T> &a, const Vec<N, T> &b) {
    Vec<N, T> r;
    for (int i = 0; i < N; i++)
        r[i] = a[i] / b[i];
    return r;
}
// matrix.hpp:805-812
template <int N, typename T> inline T dot_product(const Vec<N, T> &a, const Vec<N, T> &b) {
    T sum(0.f);
    for (int i = 0; i < N; i++)
        sum += a[i] * b[i];
    return sum;
}
// matrix.hpp:796-803
template <typename T> inline Vec3<T> cross_product(const Vec3<T> &a, const Vec3<T> &b) {
    Vec3<T> t;
    t[0] = a[1] * b[2] - b[1] * a[2];
    t[1] = a[2] * b[0] - b[2] * a[0];
    t[2] = a[0] * b[1] - b[0] * a[1];
    return t;
}
// matrix.hpp:633-641 (data[0][1] aliases data[1][0] in a 3x1 matrix): no leading zero in the sum
template <typename T> inline Vec3<T> symmetric_vector(const Vec3<T> &v, const Vec3<T> &n) {
    return ((v[0] * n[0] + v[1] * n[1] + v[2] * n[2]) * T(2.f)) * n - v;
}
template <typename T> inline Vec3<T> xyz(const Vec4<T> &v) { return mk3(v[0], v[1], v[2]); }
template <typename T> inline Vec4<T> as_pos4(const Vec3<T> &v) { return mk4(v[0], v[1], v[2], T(1.0f)); } // matrix.hpp:615-622
template <typename T> inline Vec4<T> as_dir4(const Vec3<T> &v) { return mk4(v[0], v[1], v[2], T(0.f)); }  // matrix.hpp:624-631
template <typename T> inline Vec3<T> to_vec3_as_pos(const Vec4<T> &v) {                                 // matrix.hpp:599-605
    Vec3<T> r;
    for (int i = 0; i < 3; i++)
        r[i] = v[i] / v[3];
    return r;
}

template <typename T> struct Mat4 {
    T d[4][4];
    Mat4() {
        for (auto &r : d)
            for (auto &c : r)
                c = T(0.f);
    }
    // matrix.hpp:209-220: zero-initialised accumulator, j (inner dimension) as the middle loop
    Vec4<T> operator*(const Vec4<T> &v) const {
        Vec4<T> r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++)
                r[i] += d[i][j] * v[j];
        return r;
    }
};
template <typename T> struct Mat3 {
    T d[3][3];
    Mat3() {
        for (auto &r : d)
            for (auto &c : r)
                c = T(0.f);
    }
};

template <typename T> inline Mat4<T> mat4_from(const float *m) {
    Mat4<T> r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.d[i][j] = T(m[i * 4 + j]);
    return r;
}

// matrix.hpp:829-834 (DataT = float at its only call sites)
inline float DifferenceOfProducts(float a, float b, float c, float d) {
    float cd = c * d;
    float err = std::fma(-c, d, cd);
    float dop = std::fma(a, b, -cd);
    return dop + err;
}
// matrix.hpp:836-843
inline Vec3<float> cross_product_difference(const Vec3<float> &a, const Vec3<float> &b) {
    Vec3<float> t;
    t[0] = DifferenceOfProducts(a[1], b[2], b[1], a[2]);
    t[1] = DifferenceOfProducts(a[2], b[0], b[2], a[0]);
    t[2] = DifferenceOfProducts(a[0], b[1], b[0], a[1]);
    return t;
}
// matrix.hpp:407-427
inline Mat3<float> inversed_3_3(const Mat3<float> &m) {
    const auto &a = m.d;
    Mat3<float> r;
    float det = a[0][0] * DifferenceOfProducts(a[1][1], a[2][2], a[2][1], a[1][2]) -
                a[0][1] * DifferenceOfProducts(a[1][0], a[2][2], a[1][2], a[2][0]) +
                a[0][2] * DifferenceOfProducts(a[1][0], a[2][1], a[1][1], a[2][0]);
    float invdet = 1 / det;
    r.d[0][0] = DifferenceOfProducts(a[1][1], a[2][2], a[2][1], a[1][2]) * invdet;
    r.d[0][1] = DifferenceOfProducts(a[0][2], a[2][1], a[0][1], a[2][2]) * invdet;
    r.d[0][2] = DifferenceOfProducts(a[0][1], a[1][2], a[0][2], a[1][1]) * invdet;
    r.d[1][0] = DifferenceOfProducts(a[1][2], a[2][0], a[1][0], a[2][2]) * invdet;
    r.d[1][1] = DifferenceOfProducts(a[0][0], a[2][2], a[0][2], a[2][0]) * invdet;
    r.d[1][2] = DifferenceOfProducts(a[1][0], a[0][2], a[0][0], a[1][2]) * invdet;
    r.d[2][0] = DifferenceOfProducts(a[1][0], a[2][1], a[2][0], a[1][1]) * invdet;
    r.d[2][1] = DifferenceOfProducts(a[2][0], a[0][1], a[0][0], a[2][1]) * invdet;
    r.d[2][2] = DifferenceOfProducts(a[0][0], a[1][1], a[1][0], a[0][1]) * invdet;
    return r;
}

// ---------------------------------------------------------------------------------------------
// cuRAND XORWOW (device API, CUDA Toolkit 12.3 per the reference's CMakeCache).  THIRD-PARTY,
// NOT under /root/reference: restated from the published algorithm (Marsaglia xorwow + cuRAND's
// documented seeding: seed scramble constants, subsequence = 2^67 steps, Weyl increment 362437,
// uniform = x*2^-32 + 2^-33 in (0,1], Box-Muller normals in pairs).  PARITY UNPINNED: there is
// no cuRAND here to check the stream against.
// Call sites: rt/rtrt/cuda.hpp:1715-1717, 1763-1764, 1906, 1913, 1927-1928, 1955, 1967-1969,
// 2353-2357.
// ---------------------------------------------------------------------------------------------
struct XorwowState {
    uint32_t v[5];
    uint32_t d;
    int boxmuller_flag;
    float boxmuller_extra;
};

inline uint32_t xorwow_next(XorwowState &s) {
    uint32_t t = s.v[0] ^ (s.v[0] >> 2);
    s.v[0] = s.v[1];
    s.v[1] = s.v[2];
    s.v[2] = s.v[3];
    s.v[3] = s.v[4];
    s.v[4] = (s.v[4] ^ (s.v[4] << 4)) ^ (t ^ (t << 1));
    s.d += 362437u;
    return s.v[4] + s.d;
}

// 160x160 GF(2) matrix stored by columns: col[i] = image of basis vector e_i (bit i%32 of word i/32).
struct XorwowMatrix {
    uint32_t col[160][5];
    void apply(const uint32_t in[5], uint32_t out[5]) const {
        uint32_t r[5] = {0, 0, 0, 0, 0};
        for (int w = 0; w < 5; w++) {
            uint32_t bits = in[w];
            while (bits) {
                int b = __builtin_ctz(bits);
                bits &= bits - 1;
                const uint32_t *c = col[w * 32 + b];
                for (int k = 0; k < 5; k++)
                    r[k] ^= c[k];
            }
        }
        for (int k = 0; k < 5; k++)
            out[k] = r[k];
    }
    static XorwowMatrix multiply(const XorwowMatrix &a, const XorwowMatrix &b) { // a*b: apply b, then a
        XorwowMatrix c;
        for (int i = 0; i < 160; i++)
            a.apply(b.col[i], c.col[i]);
        return c;
    }
};

class XorwowSkip {
  public:
    XorwowSkip() {
        // one-step matrix A (the xorshift part is GF(2)-linear; the Weyl counter d is separate)
        XorwowMatrix m;
        for (int i = 0; i < 160; i++) {
            XorwowState s{};
            s.v[i / 32] = 1u << (i % 32);
            xorwow_next(s);
            for (int k = 0; k < 5; k++)
                m.col[i][k] = s.v[k];
        }
        for (int i = 0; i < 67; i++) // A^(2^67): one subsequence
            m = XorwowMatrix::multiply(m, m);
        pow2_[0] = m;
        for (int k = 1; k < 32; k++)
            pow2_[k] = XorwowMatrix::multiply(pow2_[k - 1], pow2_[k - 1]);
        // byte tables for M = pow2_[0] (sequential n -> n+1 walks)
        for (int byte = 0; byte < 20; byte++)
            for (int val = 0; val < 256; val++) {
                uint32_t r[5] = {0, 0, 0, 0, 0};
                for (int b = 0; b < 8; b++)
                    if (val & (1 << b))
                        for (int k = 0; k < 5; k++)
                            r[k] ^= pow2_[0].col[byte * 8 + b][k];
                for (int k = 0; k < 5; k++)
                    lut_[byte][val][k] = r[k];
            }
    }
    // v <- M^n v
    void skip_subsequences(uint32_t v[5], uint64_t n) const {
        for (int k = 0; k < 32 && n; k++, n >>= 1)
            if (n & 1)
                pow2_[k].apply(v, v);
    }
    // v <- M v
    void next_subsequence(uint32_t v[5]) const {
        uint32_t r[5] = {0, 0, 0, 0, 0};
        for (int w = 0; w < 5; w++)
            for (int b = 0; b < 4; b++) {
                const uint32_t *e = lut_[w * 4 + b][(v[w] >> (8 * b)) & 0xff];
                for (int k = 0; k < 5; k++)
                    r[k] ^= e[k];
            }
        for (int k = 0; k < 5; k++)
            v[k] = r[k];
    }
    const XorwowMatrix &power(int k) const { return pow2_[k]; }

  private:
    XorwowMatrix pow2_[32]; // M^(2^k)
    uint32_t lut_[20][256][5];
};

inline const XorwowSkip &xorwow_skip() {
    static const XorwowSkip s;
    return s;
}

// curand_init(seed, subsequence, offset = 0)
inline XorwowState xorwow_seed_state(uint64_t seed) {
    XorwowState st{};
    uint32_t s0 = uint32_t(seed) ^ 0xaad26b49u;
    uint32_t s1 = uint32_t(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0;
    uint32_t t1 = 2591861531u * s1;
    st.d = 6615241u + t1 + t0;
    st.v[0] = 123456789u + t0;
    st.v[1] = 362436069u ^ t0;
    st.v[2] = 521288629u + t1;
    st.v[3] = 88675123u ^ t1;
    st.v[4] = 5783321u + t0;
    st.boxmuller_flag = 0;
    st.boxmuller_extra = 0.f;
    return st;
}
inline XorwowState xorwow_init(uint64_t seed, uint64_t subsequence) {
    XorwowState st = xorwow_seed_state(seed);
    xorwow_skip().skip_subsequences(st.v, subsequence); // d is unchanged: 2^67*n*362437 = 0 mod 2^32
    return st;
}

inline float xorwow_uniform(XorwowState &s) { // curand_uniform: (0, 1]
    uint32_t x = xorwow_next(s);
    return float(x) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}
inline float xorwow_normal(XorwowState &s) { // curand_normal: Box-Muller, second value cached
    if (s.boxmuller_flag) {
        s.boxmuller_flag = 0;
        return s.boxmuller_extra;
    }
    uint32_t x = xorwow_next(s);
    uint32_t y = xorwow_next(s);
    float u = float(x) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
    float v = float(y) * 1.46291807e-09f + (1.46291807e-09f / 2.0f); // 2*pi*2^-32
    // the transcendentals that shape rays come from include/lprt_fmath.h on both sides (see its header)
    float r = std::sqrt(-2.0f * lprt_logf(u)), sn, cs;
    lprt_sincosf(v, &sn, &cs);
    s.boxmuller_extra = r * cs;
    s.boxmuller_flag = 1;
    return r * sn;
}

} // namespace orc
#endif
