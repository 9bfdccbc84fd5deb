// SVGF temporal/spatial denoiser, TAA, output and temporal back-up kernels.
//
// Replaces preprocess_normal_depth, temporal_accumulation_color, temporal_accumulation_moments,
// wavelet_filter<stride>, svgf_denoise (rt/rtrt/cuda.hpp:1187-1473, 1636-1711), add_denoised_color,
// temporal_anti_aliasing, write_to_surface2d, copy_last_frame_pixel_id/transform (cuda.hpp:2201-2263,
// 2418-2455).  Formulas: SURVEY.md Appendix C.
//
// The reference runs svgf_denoise twice per frame (colored / white lobe, imgui_window.cu:232-238),
// recomputing identical normal/depth planes and identical geometric weights.  Here both lobes are
// filtered by the same thread: w_z, w_n and h are evaluated once per tap and shared; only w_l differs.
// Colour and variance of a lobe travel as one {r,g,b,var} vector (one load per tap).
#include "kernels.hpp"
#include "shade.hpp"

namespace lprt {

template <typename T> struct SvgfConst {
    static LPRT_DEV T eps() { return cvt<T>(1e-5f); } // a half subnormal (cuda.hpp:1261)
};

// Edge-stopping weights are exp() / pow(x,128) of per-tap quantities: 4 transcendentals per tap made the filters
// ALU-bound (390 us per a-trous level at 1080p).  The default build evaluates them with the hardware exp2 and 7
// squarings (relative error <= ~4e-6 in f32, invisible after rounding to half); -DLPRT_EXACT_SVGF_MATH restores the
// libm-accurate forms the parity of the other stages is measured with.
LPRT_DEV float svgf_exp(float x) {
#ifdef LPRT_EXACT_SVGF_MATH
    return ::expf(x);
#else
    return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f);
#endif
}
LPRT_DEV float svgf_div(float a, float b) { // a / b: IEEE in the exact build, a * rcp(b) (1 ulp) otherwise
#ifdef LPRT_EXACT_SVGF_MATH
    return a / b;
#else
    return a * __builtin_amdgcn_rcpf(b);
#endif
}
LPRT_DEV float svgf_pow128(float x) {
#ifdef LPRT_EXACT_SVGF_MATH
    return ::powf(x, 128.f);
#else
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8, x32 = x16 * x16, x64 = x32 * x32;
    return x64 * x64;
#endif
}

// sample_linear (cuda.hpp:1085-1115) on a 4-vector plane: xyz blended, result / sum via the rounded reciprocal
template <typename T> LPRT_DEV V3<T> sample_linear3(const TemporalMapT<T> &m, const V4<T> *arr, uint32_t W, V3<T> dflt) {
    if (m.frame_count == 0)
        return dflt;
    T sum = (T)0;
    V3<T> r = zero3<T>();
    for (int i = 0; i < 4; i++) {
        T w = m.weights[i];
        if (w == (T)0)
            continue;
        sum = sum + w;
        r = r + w * xyz(arr[(uint32_t)m.pos[i][1] * W + m.pos[i][0]]);
    }
    return vdiv(r, sum);
}
template <typename T> LPRT_DEV T sample_linear1(const TemporalMapT<T> &m, const T *arr, uint32_t W, T dflt) {
    if (m.frame_count == 0)
        return dflt;
    T sum = (T)0, r = (T)0;
    for (int i = 0; i < 4; i++) {
        T w = m.weights[i];
        if (w == (T)0)
            continue;
        sum = sum + w;
        r = r + w * arr[(uint32_t)m.pos[i][1] * W + m.pos[i][0]];
    }
    return tdiv(r, sum); // scalar: a true division
}

// ---- preprocess_normal_depth (cuda.hpp:1442-1473): only the gradient needs a plane of its own ---------------
template <typename T> __global__ void __launch_bounds__(256) k_depth_gradient(FrameDev<T> fb) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= fb.win.npix())
        return;
    const uint32_t W = fb.win.W, x = p % W, ly = p / W;
    const T dp = fb.g_normal_depth[p].w;
    V2<T> g;
    // (no neighbour to difference against in a one-pixel-wide / one-row window: gradient 0 instead of the reference's read
    // past the surface)
    g.x = (x == 0) ? (W > 1 ? fb.g_normal_depth[p + 1].w - dp : (T)0) : dp - fb.g_normal_depth[p - 1].w;
    g.y = (ly == 0) ? (fb.win.rows() > 1 ? fb.g_normal_depth[p + W].w - dp : (T)0) : dp - fb.g_normal_depth[p - W].w;
    fb.depth_gradient[p] = g;
}

// ---- temporal_accumulation_color (cuda.hpp:1187-1243), both lobes -------------------------------------------------
// Reads the lobe's cv_a (this frame) and cv_b (history), writes cv_c.rgb and illum: out of place, so the 9x9
// window always sees this frame's unmodified input (the reference updates in place and races).
// One workgroup = a 32x8 pixel tile; the (32+8)x(8+8) input window of both lobes is staged in LDS once (5 loads per
// thread instead of 162) and the 81 taps are LDS reads, accumulated in the reference's order (x outer, y inner).
constexpr int TC_BX = 32, TC_BY = 8, TC_R = 4, TC_TW = TC_BX + 2 * TC_R, TC_TH = TC_BY + 2 * TC_R;
template <typename T> struct PackedV4 { // a V4<T> as two pairs for packed arithmetic
    typedef T v2 __attribute__((ext_vector_type(2)));
    v2 xy, zw;
};
template <typename T>
__global__ void __launch_bounds__(256) k_temporal_color(FrameDev<T> fb, T color_w, T moments_w, int always_test) {
    __shared__ V4<T> tile[2][TC_TH][TC_TW];
    __shared__ uint32_t s_nonfinite; // some staged colour component is inf / NaN: the window sums must test every sample
    const int W = (int)fb.win.W, R = (int)fb.win.rows();
    const int tx = (int)(threadIdx.x % TC_BX), ty = (int)(threadIdx.x / TC_BX);
    const int x0 = (int)blockIdx.x * TC_BX - TC_R, y0 = (int)blockIdx.y * TC_BY - TC_R;
    if (threadIdx.x == 0)
        s_nonfinite = 0u;
    __syncthreads();
    bool bad = false, nonzero = false;
    for (int e = (int)threadIdx.x; e < TC_TW * TC_TH; e += 256) {
        const int lx = e % TC_TW, ly = e / TC_TW;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx >= 0 && gx < W && gy >= 0 && gy < R) {
            const uint32_t q = (uint32_t)(gy * W + gx);
            const V4<T> c0 = fb.lobe[0].cv_a[q], c1 = fb.lobe[1].cv_a[q];
            tile[0][ly][lx] = c0;
            tile[1][ly][lx] = c1;
            bad = bad || !(tfinite(c0.x) && tfinite(c0.y) && tfinite(c0.z) && tfinite(c1.x) && tfinite(c1.y) && tfinite(c1.z));
            nonzero = nonzero || !(c0.x == (T)0 && c0.y == (T)0 && c0.z == (T)0 && c1.x == (T)0 && c1.y == (T)0 && c1.z == (T)0);
        } else {
            // outside the frame: zeros, so that the untested window sums below can run over all 81 samples of every pixel (a sum
            // that starts at +0 is never -0, and x + (+0) == x for every other x: the zeros change no bit of it)
            tile[0][ly][lx] = tile[1][ly][lx] = V4<T>{(T)0, (T)0, (T)0, (T)0};
        }
    }
    {
        const uint32_t flags = (__builtin_amdgcn_ballot_w64(bad) ? 1u : 0u) | (__builtin_amdgcn_ballot_w64(nonzero) ? 2u : 0u);
        if (flags && (threadIdx.x & 63) == 0)
            atomicOr(&s_nonfinite, flags);
    }
    __syncthreads();
    // workgroup-uniform (LPRT_SVGF_ALWAYS_TEST=1 forces the per-sample tests for A/B runs)
    const bool all_finite = (s_nonfinite & 1u) == 0u && !always_test;
    // every staged colour is zero (a tile of background without a skybox): the window sums of +-0 are +0 whatever the count,
    // the 81-sample loops are left out and the rest of the pass runs on those sums
    const bool all_zero = (s_nonfinite & 2u) == 0u && !always_test;
    const int px = (int)blockIdx.x * TC_BX + tx, py = (int)blockIdx.y * TC_BY + ty;
    if (px >= W || py >= R)
        return;
    const uint32_t p = (uint32_t)(py * W + px);
    const TemporalMapT<T> ti = fb.svgf_map[p];
    const T zero = (T)0, one = (T)1, half_ = cvt<T>(0.5f);
#pragma unroll
    for (int l = 0; l < 2; l++) {
        V3<T> pc = xyz(tile[l][ty + TC_R][tx + TC_R]);
        V3<T> m1 = zero3<T>(), m2 = zero3<T>(), wt = zero3<T>();
        if (all_zero) {
            wt = mk3(one, one, one); // 0 / n = +0 for every count n >= 1
        } else if (all_finite) {
            // every sample of the window counts: the same sums in the same order without the per-sample tests; the
            // sample count is a small integer, exact in either precision
            // two channels per instruction (v_pk_add/mul_f16, v_pk_add/mul_f32): each lane of a packed operation is an
            // IEEE operation of its own, rounded like the scalar one; the w channel rides along unused
            typedef typename PackedV4<T>::v2 v2;
            v2 a_xy = {zero, zero}, a_zw = a_xy, b_xy = a_xy, b_zw = a_xy;
            int cnt = 0;
            if (sizeof(T) == 2) {
                // half: all 81 samples of every pixel, no bounds tests -- the samples outside the frame are staged as zeros and
                // summed along, the ones inside are a rectangle (0.140 -> 0.133 ms; in single precision the nine 16-byte reads
                // of a column in flight cost 60 registers and three waves per SIMD: 0.141 -> 0.149, the tested loop stays)
                cnt = ((px + 4 < W ? px + 4 : W - 1) - (px - 4 > 0 ? px - 4 : 0) + 1) * ((py + 4 < R ? py + 4 : R - 1) - (py - 4 > 0 ? py - 4 : 0) + 1);
                const V4<T> *t_win = &tile[l][ty][tx]; // the window's first sample: every other one at a constant offset
#pragma unroll 1
                for (int i = -4; i <= 4; i++) {
#pragma unroll
                    for (int j = -4; j <= 4; j++) {
                        const PackedV4<T> q = __builtin_bit_cast(PackedV4<T>, t_win[(j + TC_R) * TC_TW + (i + TC_R)]);
                        a_xy = a_xy + q.xy;
                        b_xy = b_xy + q.xy * q.xy;
                        a_zw = a_zw + q.zw;
                        b_zw = b_zw + q.zw * q.zw;
                    }
                }
            } else {
                for (int i = -4; i <= 4; i++) {
                    int qx = px + i;
                    if (qx < 0 || qx >= W)
                        continue;
#pragma unroll
                    for (int j = -4; j <= 4; j++) {
                        int qy = py + j;
                        if (qy < 0 || qy >= R)
                            continue;
                        const PackedV4<T> q = __builtin_bit_cast(PackedV4<T>, tile[l][ty + TC_R + j][tx + TC_R + i]);
                        cnt++;
                        a_xy = a_xy + q.xy;
                        b_xy = b_xy + q.xy * q.xy;
                        a_zw = a_zw + q.zw;
                        b_zw = b_zw + q.zw * q.zw;
                    }
                }
            }
            m1 = mk3((T)a_xy.x, (T)a_xy.y, (T)a_zw.x);
            m2 = mk3((T)b_xy.x, (T)b_xy.y, (T)b_zw.x);
            wt = mk3(cvt<T>((float)cnt), cvt<T>((float)cnt), cvt<T>((float)cnt));
        } else {
            for (int i = -4; i <= 4; i++) {
                int qx = px + i;
                if (qx < 0 || qx >= W)
                    continue;
#pragma unroll
                for (int j = -4; j <= 4; j++) {
                    int qy = py + j;
                    if (qy < 0 || qy >= R)
                        continue;
                    V4<T> q = tile[l][ty + TC_R + j][tx + TC_R + i];
                    if (tfinite(q.x)) { wt.x = wt.x + one; m1.x = m1.x + q.x; m2.x = m2.x + q.x * q.x; }
                    if (tfinite(q.y)) { wt.y = wt.y + one; m1.y = m1.y + q.y; m2.y = m2.y + q.y * q.y; }
                    if (tfinite(q.z)) { wt.z = wt.z + one; m1.z = m1.z + q.z; m2.z = m2.z + q.z * q.z; }
                }
            }
        }
        m1 = elediv(m1, wt);
        m2 = elediv(m2, wt);
        value_or(pc, m1);
        V3<T> var = m2 - elemul(m1, m1);
        T sx = tsqrt(var.x), sy = tsqrt(var.y), sz = tsqrt(var.z);
        pc.x = clip_both(m1.x - half_ * sx, m1.x + half_ * sx, pc.x);
        pc.y = clip_both(m1.y - half_ * sy, m1.y + half_ * sy, pc.y);
        pc.z = clip_both(m1.z - half_ * sz, m1.z + half_ * sz, pc.z);
        V3<T> hist = sample_linear3(ti, fb.lobe[l].cv_b, fb.win.W, pc);
        value_or(hist, pc);
        V3<T> out = color_w * pc + (one - color_w) * hist;
        T illum = cvt<T>(0.2126f) * out.x + cvt<T>(0.7152f) * out.y + cvt<T>(0.0722f) * out.z;
        fb.lobe[l].illum[p] = illum;
        // temporal_accumulation_moments for a pixel with >= 4 frames of history needs nothing but this pixel's luminance and
        // its history taps (cuda.hpp:1307-1330): done here, in the thread that has both -- k_moments keeps the pixels whose
        // moments come from the 5x5 neighbourhood of THIS pass's output (frame_count < 4).
        T lvar = zero; // variance of the luminance (cv_c.w)
        if (ti.frame_count >= 4u) {
            const T i2 = illum * illum;
            T miu1 = (one - moments_w) * sample_linear1(ti, fb.lobe[l].miu1_hist, fb.win.W, zero) + illum * moments_w;
            T miu2 = (one - moments_w) * sample_linear1(ti, fb.lobe[l].miu2_hist, fb.win.W, zero) + i2 * moments_w;
            if (!tfinite(miu1))
                miu1 = illum;
            if (!tfinite(miu2))
                miu2 = i2;
            fb.lobe[l].miu1_cur[p] = miu1;
            fb.lobe[l].miu2_cur[p] = miu2;
            lvar = miu2 - miu1 * miu1; // variance, not clamped at 0
            fb.lobe[l].var_c[p] = lvar;
        }
        fb.lobe[l].cv_c[p] = mk4(out.x, out.y, out.z, lvar);
    }
}

// geometric part of a tap's weight, shared by both lobes: h * w_z * w_n is NOT pre-multiplied because the
// reference multiplies w_z * w_n (* w_l) first and h last
// w_n of a neighbour whose normal equals the centre's bit for bit is pow(max(0, dot(n_p, n_p)), 128): one value per pixel
// a * b + c.  Half: two roundings, as the reference's float16 operators round (the half kernels reproduce the reference's
// long-sequence behaviour pixel for pixel, tests/test_gpu_long_sequences.py).  Single precision: ONE rounding (v_fma_f32) --
// the a-trous taps are bound by their vector instructions (~1,750 per pixel), the north star holds single-precision radiance
// to 1e-3 relative L2 (the fused form moves it by ~1e-7), and a multiply-add per tap and channel is a third of the tap's
// arithmetic.  -DLPRT_SVGF_NO_FMA restores the two-rounding form.
template <typename T> LPRT_DEV T fmadd(T a, T b, T c) { return a * b + c; }
#ifndef LPRT_SVGF_NO_FMA
template <> LPRT_DEV float fmadd<float>(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
#endif
template <typename T> LPRT_DEV T dot_fm(V3<T> a, V3<T> b) { // dot_product (matrix.hpp:805-812): zero, then left to right
    T s = (T)0;
    s = fmadd(a.x, b.x, s);
    s = fmadd(a.y, b.y, s);
    s = fmadd(a.z, b.z, s);
    return s;
}
template <typename T> LPRT_DEV T normal_weight(V3<T> n_p, V3<T> n_q) { return cvt<T>(svgf_pow128((float)tmax((T)0, dot_fm(n_p, n_q)))); }
template <typename T>
LPRT_DEV void geo_weights(T depth_p, T depth_q, V2<T> grad, T dx, T dy, V3<T> n_p, V3<T> n_q, T &w_z, T &w_n, T w_nn, bool flat_ok) {
    // exp(-abs(dz) / (SIGMA_Z * abs(dot(grad, dp) + EPS))): the unary minus makes the quotient a float.
    // Traced G-buffers carry depth 1 on every covered pixel (Q11), so dz is 0 for all taps of nearly every wave: then
    // the quotient is -0 (the denominator is never 0: dot(grad, dp) is an integer multiple of the stride, EPS is not)
    // and exp(-0) = 1 exactly.  The wave skips the reciprocal and the exponential when no lane has a depth step.
    w_z = (T)1;
    if (__builtin_amdgcn_ballot_w64(!(depth_p == depth_q))) {
        float num = -(float)tabs(depth_p - depth_q);
        float den = (float)((T)1 * tabs(dot(grad, V2<T>{dx, dy}) + SvgfConst<T>::eps()));
        w_z = cvt<T>(svgf_exp(svgf_div(num, den)));
    }
    // Flat surfaces: every lane's neighbour carries the centre's normal, dot(n_p, n_q) IS dot(n_p, n_p), and the wave skips
    // the dot product and the seven squarings for the per-pixel value the caller computed once.
    const bool same = n_q.x == n_p.x && n_q.y == n_p.y && n_q.z == n_p.z;
    if (flat_ok && !__builtin_amdgcn_ballot_w64(!same))
        w_n = w_nn;
    else
        w_n = normal_weight(n_p, n_q);
}

#ifndef LPRT_WV_SKIP_EMPTY
#define LPRT_WV_SKIP_EMPTY 1 // waves of pixels without a normal leave the weighted windows out (k_moments, k_wavelet)
#endif
// ---- temporal_accumulation_moments (cuda.hpp:1245-1337), both lobes ---------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_moments(FrameDev<T> fb) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= fb.win.npix())
        return;
    if (fb.valid_frame_count[p] >= 4u) // (= svgf_map[p].frame_count) done by k_temporal_color
        return;
    const int W = (int)fb.win.W, R = (int)fb.win.rows();
    const int px = (int)(p % fb.win.W), py = (int)(p / fb.win.W);
    const T zero = (T)0;
    const T hk[3] = {cvt<T>(0.375f), cvt<T>(0.25f), cvt<T>(0.0625f)};
    T miu1[2] = {zero, zero}, miu2[2] = {zero, zero};
    { // frame_count < 4: spatial estimate over the 5x5 neighbourhood (cuda.hpp:1262-1305)
        const V4<T> nd_p = fb.g_normal_depth[p];
        const V2<T> grad = fb.depth_gradient[p];
        T weight[2] = {zero, zero};
        // (a wave of pixels without a normal leaves the window out, as in k_wavelet: every h * w is +0 or NaN, the sums stay +0)
        const bool no_normal = nd_p.x == zero && nd_p.y == zero && nd_p.z == zero;
        const bool wave_has_taps = !LPRT_WV_SKIP_EMPTY || __builtin_amdgcn_ballot_w64(!no_normal) != 0ull;
        for (int i = -2; wave_has_taps && i <= 2; i++) {
            int qx = px + i;
            if (qx < 0 || qx >= W)
                continue;
            for (int j = -2; j <= 2; j++) {
                int qy = py + j;
                if (qy < 0 || qy >= R)
                    continue;
                uint32_t q = (uint32_t)(qy * W + qx);
                T h = hk[i < 0 ? -i : i] * hk[j < 0 ? -j : j];
                const V4<T> nd_q = fb.g_normal_depth[q];
                T w_z, w_n;
                geo_weights(nd_p.w, nd_q.w, grad, cvt<T>((float)i), cvt<T>((float)j), xyz(nd_p), xyz(nd_q), w_z, w_n, zero, false);
                T hw = h * (w_z * w_n);
                if (tfinite(hw)) {
#pragma unroll
                    for (int l = 0; l < 2; l++) {
                        T iq = fb.lobe[l].illum[q];
                        if (tfinite(iq)) {
                            miu1[l] = miu1[l] + hw * iq;
                            miu2[l] = miu2[l] + hw * iq * iq;
                            weight[l] = weight[l] + hw;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int l = 0; l < 2; l++) {
            miu1[l] = tdiv(miu1[l], weight[l]);
            miu2[l] = tdiv(miu2[l], weight[l]);
        }
    }
#pragma unroll
    for (int l = 0; l < 2; l++) {
        fb.lobe[l].miu1_cur[p] = miu1[l];
        fb.lobe[l].miu2_cur[p] = miu2[l];
        const T var = miu2[l] - miu1[l] * miu1[l]; // variance, not clamped at 0
        fb.lobe[l].cv_c[p].w = var;
        fb.lobe[l].var_c[p] = var;
    }
}

// ---- wavelet_filter<stride> (cuda.hpp:1339-1440), both lobes ---------------------------------------------------------------
// One workgroup filters 64 x 4 points of ONE phase of the stride-s lattice (rows oy + s*k): on that lattice the a-trous
// filter is a dense 5-row stencil, so the 8 lattice rows x (64 + 4s) columns it needs are staged in LDS once (every
// load a full coalesced row segment) and the 25 taps x 4 planes become LDS reads — for every stride, including 8 and
// 16 where a plain image tile would not fit.  Per thread: ~10-16 global loads instead of 125.
#ifndef LPRT_WV_LDS_BELOW
#define LPRT_WV_LDS_BELOW 8
#endif
constexpr int WV_BX = 64;
#ifndef LPRT_WV_FAST32
#define LPRT_WV_FAST32 0 // the one-test taps and the flat-normal skip in single precision: measured slower, see below
#endif
#ifndef LPRT_WV_FLAT32
#define LPRT_WV_FLAT32 0 // the flat-normal skip in single precision (interpolated normals of a flat wall rarely agree to the bit there)
#endif
#ifndef LPRT_WV_XCD
#define LPRT_WV_XCD 1
#endif
#ifndef LPRT_WV_EAGER
#define LPRT_WV_EAGER 0 // both lobes' colours requested with the tap's normal / depth / illumination instead of after the weight test
#endif
#ifndef LPRT_WV_BY32
#define LPRT_WV_BY32 8
#endif
// rows of lattice points per workgroup: single precision stages 56 B per texel and is bound by how many waves its LDS tiles
// leave room for -- 8 rows (512 threads) amortise the 4 halo rows better than 4 (6 instead of 4-5 waves per SIMD at strides
// 1 and 2); half precision has room to spare and keeps 4
#ifndef LPRT_WV_BY16
#define LPRT_WV_BY16 4
#endif
template <typename T> struct WvRows { static constexpr int value = sizeof(T) == 4 ? LPRT_WV_BY32 : LPRT_WV_BY16; };
template <typename T, int STRIDE>
__global__ void __launch_bounds__(WV_BX * WvRows<T>::value) k_wavelet(FrameDev<T> fb, const V4<T> *in0, V4<T> *out0, const V4<T> *in1, V4<T> *out1, const T *vin0, T *vout0,
                                                 const T *vin1, T *vout1, int always_test, int lat0, int row_hi) {
    constexpr int WV_BY = WvRows<T>::value, WV_THREADS = WV_BX * WV_BY;
    constexpr int TW = WV_BX + 4 * STRIDE, TH = WV_BY + 4;
    // fp32 at strides 8 and 16: four staged planes (56 B per texel) leave room for only 3 / 2 workgroups per CU, and
    // the kernel is latency-bound; there the colour planes are read straight from L1/L2 (rows are still contiguous
    // across the wave) and LDS keeps the 24 B per texel that both lobes share.
    constexpr bool LDS_COLOR = sizeof(T) == 2 || STRIDE < LPRT_WV_LDS_BELOW;
    __shared__ V4<T> s_nd[TH][TW], s_c0[LDS_COLOR ? TH : 1][LDS_COLOR ? TW : 1], s_c1[LDS_COLOR ? TH : 1][LDS_COLOR ? TW : 1];
    __shared__ V2<T> s_il[TH][TW];
    __shared__ uint32_t s_bad[WV_BY]; // per wave: some colour / variance it staged is inf or NaN (then the taps test every sample)
    const int W = (int)fb.win.W, R = (int)fb.win.rows();
    const int tx = (int)(threadIdx.x % WV_BX), ty = (int)(threadIdx.x / WV_BX);
#if LPRT_WV_XCD
    // Workgroup -> tile.  Workgroups go to the 8 XCDs round-robin in launch order and every XCD has an L2 of its own, so tiles that
    // share halo rows and columns (a third to two thirds of what a tile reads) were fetched by different L2s.  Here the
    // workgroups of one XCD take a contiguous run of the tile order (phase, then tile row, then tile column): a tile's
    // neighbours are its XCD's previous workgroup and the one a tile row earlier.  The work per tile is uniform, so the runs end together.
    const uint32_t gx = gridDim.x, n_wg = gx * gridDim.y, lin = blockIdx.y * gx + blockIdx.x;
    const uint32_t xcd = lin & 7u, run = n_wg >> 3, extra = n_wg & 7u;
    const uint32_t vlin = xcd * run + (xcd < extra ? xcd : extra) + (lin >> 3);
    const uint32_t bx = vlin % gx, vy = vlin / gx, nby = gridDim.y / STRIDE;
    const int oy = (int)(vy / nby), by = (int)(vy % nby);
#else
    const uint32_t bx = blockIdx.x;
    const int oy = (int)(blockIdx.y % STRIDE), by = (int)(blockIdx.y / STRIDE);
#endif
    const int x0 = (int)bx * WV_BX - 2 * STRIDE; // image x of tile column 0
    // lat0 / row_hi: the rows this launch has to produce start at lattice row lat0 of every phase and end before window row
    // row_hi (launch_wavelet: a rank's strip needs level l only within the reach of the levels after it)
    const int r0 = lat0 + WV_BY * by - 2;                 // lattice row of tile row 0
    constexpr bool FAST = LDS_COLOR && (sizeof(T) == 2 || LPRT_WV_FAST32);
    const T zero = (T)0;
    // ---- every global read of the workgroup is requested before the first one is waited for: the thread's own 3 x 3 variances
    // and depth gradient, then its share of the tile (2-4 texels).  Requested where they are used -- tile loop, barrier, sigma --
    // they were three dependent round trips to memory per workgroup, and the three workgroups of a CU take them in step.
    // Addresses outside the window are clamped into it (the value is dropped), so nothing here branches.
    const int px = (int)bx * WV_BX + tx, py = oy + STRIDE * (lat0 + WV_BY * by + ty);
    const bool mine = px < W && py < row_hi;
    const int pxc = px < W ? px : W - 1, pyc = py < R ? py : R - 1;
    const uint32_t p = (uint32_t)(pyc * W + pxc);
    T pv0[9], pv1[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { // k = 3 (i + 1) + (j + 1): x outer, y inner
        const int qx = pxc + k / 3 - 1, qy = pyc + k % 3 - 1;
        const uint32_t q = (uint32_t)((qy < 0 ? 0 : qy >= R ? R - 1 : qy) * W + (qx < 0 ? 0 : qx >= W ? W - 1 : qx));
        pv0[k] = vin0[q];
        pv1[k] = vin1[q];
    }
    const V2<T> grad = fb.depth_gradient[p];
    constexpr int NIT = (TW * TH + WV_THREADS - 1) / WV_THREADS;
    V4<T> r_nd[NIT], r_c0[LDS_COLOR ? NIT : 1], r_c1[LDS_COLOR ? NIT : 1];
    V2<T> r_il[NIT];
    bool r_in[NIT];
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        // (threads past the tile's end in the last round all take its last texel, and all write it: no branch around the round)
        const int e = min((int)threadIdx.x + it * WV_THREADS, TW * TH - 1);
        const int lx = e % TW, ly = e / TW;
        const int gx = x0 + lx, gy = oy + STRIDE * (r0 + ly);
        r_in[it] = gx >= 0 && gx < W && gy >= 0 && gy < R;
        const uint32_t q = (uint32_t)((gy < 0 ? 0 : gy >= R ? R - 1 : gy) * W + (gx < 0 ? 0 : gx >= W ? W - 1 : gx));
        r_nd[it] = fb.g_normal_depth[q];
        if (LDS_COLOR) {
            r_c0[it] = in0[q];
            r_c1[it] = in1[q];
        }
        r_il[it] = V2<T>{fb.lobe[0].illum[q], fb.lobe[1].illum[q]};
    }
    bool bad = false;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const int e = min((int)threadIdx.x + it * WV_THREADS, TW * TH - 1);
        const int lx = e % TW, ly = e / TW;
        s_nd[ly][lx] = r_nd[it];
        if (LDS_COLOR) {
            const V4<T> c0 = r_c0[it], c1 = r_c1[it];
            s_c0[ly][lx] = c0;
            s_c1[ly][lx] = c1;
            if (FAST)
                bad = bad || (r_in[it] && !(tfinite(c0.x) && tfinite(c0.y) && tfinite(c0.z) && tfinite(c0.w) && tfinite(c1.x) && tfinite(c1.y) &&
                                            tfinite(c1.z) && tfinite(c1.w)));
        }
        // a tap outside the window is skipped (cuda.hpp:1375-1380).  Here it carries a NaN illumination: its luminance weight, and
        // with it h * w of both lobes, is NaN, and the test every tap makes anyway (`isfinite(h * w)`) drops it -- the 25 x 2
        // bounds tests per pixel leave the tap loop
        s_il[ly][lx] = r_in[it] ? r_il[it] : V2<T>{Num<T>::nan(), Num<T>::nan()};
    }
    // every lane stores its wave's flag (one word, one value per wave).  [The same store from lane 0 alone, behind a branch, made
    // workgroups with non-finite samples take the one-test path on MI355X -- ISA identical but for the store's lane mask; the
    // `svgf_always_test` A/B test and tools/gpu_ab_hash.sh catch it.]
    const bool wave_bad = __builtin_amdgcn_ballot_w64(bad) != 0;
    if (FAST)
        s_bad[threadIdx.x >> 6] = wave_bad ? 1u : 0u;
    __syncthreads();
    // workgroup-uniform: with every staged sample finite, a tap needs one test (its own weight) instead of six, both
    // weight sums coincide and colour + variance accumulate as two packed pairs
    bool all_finite = FAST && !always_test;
    if (FAST) {
        uint32_t any_bad = 0u;
#pragma unroll
        for (int w = 0; w < WV_BY; w++)
            any_bad |= s_bad[w];
        all_finite = all_finite && any_bad == 0u;
    }
    if (!mine)
        return;
    const int cx = tx + 2 * STRIDE, cy = ty + 2; // tile coordinates of the centre
    const T hk[3] = {cvt<T>(0.375f), cvt<T>(0.25f), cvt<T>(0.0625f)};
    const T gk[2] = {cvt<T>(0.5f), cvt<T>(0.25f)};
    const V4<T> *in[2] = {in0, in1};
    V4<T> *out[2] = {out0, out1};
    // 1. 3x3 gaussian of the variance (image neighbours, not lattice neighbours) -> sigma_l
    T sg[2] = {zero, zero};
    T gw = zero;
#pragma unroll
    for (int i = -1; i <= 1; i++) {
        int qx = px + i;
        if (qx < 0 || qx >= W)
            continue;
#pragma unroll
        for (int j = -1; j <= 1; j++) {
            int qy = py + j;
            if (qy < 0 || qy >= R)
                continue;
            T g = gk[i < 0 ? -i : i] * gk[j < 0 ? -j : j];
            sg[0] = fmadd(g, pv0[3 * (i + 1) + (j + 1)], sg[0]);
            sg[1] = fmadd(g, pv1[3 * (i + 1) + (j + 1)], sg[1]);
            gw = gw + g;
        }
    }
    sg[0] = tsqrt(tdiv(sg[0], gw));
    sg[1] = tsqrt(tdiv(sg[1], gw));
    // w_l = exp(-|l_p - l_q| / (4 sigma_l + eps)): the denominator is per pixel
    const float den_l[2] = {(float)(cvt<T>(4.f) * sg[0] + SvgfConst<T>::eps()), (float)(cvt<T>(4.f) * sg[1] + SvgfConst<T>::eps())};
#ifndef LPRT_EXACT_SVGF_MATH
    // the per-pixel reciprocal and the log2(e) of exp() leave the tap loop: w_l = exp2(-|dl| * (log2e / den))
    const float k_l[2] = {__builtin_amdgcn_rcpf(den_l[0]) * 1.44269504088896340736f, __builtin_amdgcn_rcpf(den_l[1]) * 1.44269504088896340736f};
#endif
    // 2. 5x5 a-trous taps, x outer / y inner as in the reference
    // every tap of the thread at a constant offset from one LDS address
    const int t0 = ty * TW + tx;
    const V4<T> *t_nd = &s_nd[0][0] + t0, *t_c0 = &s_c0[0][0] + (LDS_COLOR ? t0 : 0), *t_c1 = &s_c1[0][0] + (LDS_COLOR ? t0 : 0);
    const V2<T> *t_il = &s_il[0][0] + t0;
    const V4<T> nd_p = s_nd[cy][cx];
    // (half only: in single precision the skip measured 3 % slower, 0.650 vs 0.628 ms -- that kernel is not bound by its VALU count)
    const bool FLAT = (sizeof(T) == 2 || LPRT_WV_FLAT32) && !always_test;
    const T w_nn = normal_weight(xyz(nd_p), xyz(nd_p));
    const V2<T> il_p = s_il[cy][cx];
    const T illum_p[2] = {il_p.x, il_p.y};
    T nvar[2] = {zero, zero}, wv[2] = {zero, zero}, wc[2] = {zero, zero};
    V3<T> ncol[2] = {zero3<T>(), zero3<T>()};
    // A pixel whose normal is the zero vector -- every pixel no primary ray covered -- gives every tap the normal weight
    // pow(max(0, 0 . n_q), 128) = +0 (or NaN, where n_q is not finite): h * w is +0 or NaN for all 25 taps, the sums stay at
    // the +0 they start from and the pixel keeps its own colour and variance (the 0 / 0 below).  A wave of such pixels leaves the
    // taps out and arrives at the same sums: a third of the Cornell frame is background.  (LPRT_WV_SKIP_EMPTY=0: never)
    const bool no_normal = nd_p.x == zero && nd_p.y == zero && nd_p.z == zero;
    const bool wave_has_taps = !LPRT_WV_SKIP_EMPTY || always_test || __builtin_amdgcn_ballot_w64(!no_normal) != 0ull;
    if (!wave_has_taps) {
    } else if (all_finite) {
        typedef typename PackedV4<T>::v2 v2;
        v2 acc_xy[2] = {{zero, zero}, {zero, zero}}, acc_zw[2] = {{zero, zero}, {zero, zero}}; // colour x y | colour z, variance
        T wsum[2] = {zero, zero};
#pragma unroll
        for (int i = -2; i <= 2; i++) {
#pragma unroll
            for (int j = -2; j <= 2; j++) {
                const int tq = (j + 2) * TW + (i + 2) * STRIDE; // the tap, relative to the thread's top-left one
                T h = hk[i < 0 ? -i : i] * hk[j < 0 ? -j : j];
                const V4<T> nd_q = t_nd[tq];
                T w_z, w_n;
                geo_weights(nd_p.w, nd_q.w, grad, cvt<T>((float)(STRIDE * i)), cvt<T>((float)(STRIDE * j)), xyz(nd_p), xyz(nd_q), w_z, w_n, w_nn, FLAT);
                T wzn = w_z * w_n;
                const V2<T> il_q = t_il[tq];
                const T iq[2] = {il_q.x, il_q.y};
                constexpr bool EAGER = LDS_COLOR && LPRT_WV_EAGER;
                const V4<T> cq_e[2] = {t_c0[EAGER ? tq : 0], t_c1[EAGER ? tq : 0]};
#pragma unroll
                for (int l = 0; l < 2; l++) {
                    float num = -(float)tabs(illum_p[l] - iq[l]);
#ifdef LPRT_EXACT_SVGF_MATH
                    T w_l = cvt<T>(svgf_exp(svgf_div(num, den_l[l])));
#else
                    T w_l = cvt<T>(__builtin_amdgcn_exp2f(num * k_l[l]));
#endif
                    T hw = h * (wzn * w_l);
                    if (tfinite(hw)) {
                        if (sizeof(T) == 4) {
                            // single precision: scalar statements, which the compiler pairs into v_pk_fma_f32 itself; the explicit
                            // pairs below cost it nine register copies per tap around the branch
                            const V4<T> cq = EAGER ? cq_e[l] : (l == 0 ? t_c0 : t_c1)[LDS_COLOR ? tq : 0];
                            nvar[l] = fmadd(hw * hw, cq.w, nvar[l]);
                            ncol[l] = mk3(fmadd(hw, cq.x, ncol[l].x), fmadd(hw, cq.y, ncol[l].y), fmadd(hw, cq.z, ncol[l].z));
                        } else {
                            const PackedV4<T> cq = __builtin_bit_cast(PackedV4<T>, EAGER ? cq_e[l] : (l == 0 ? t_c0 : t_c1)[LDS_COLOR ? tq : 0]);
                            const v2 hh = {hw, hw}, hq = {hw, hw * hw};
                            acc_xy[l] = acc_xy[l] + hh * cq.xy;
                            acc_zw[l] = acc_zw[l] + hq * cq.zw; // hw * z | (hw * hw) * variance
                        }
                        wsum[l] = wsum[l] + hw;
                    }
                }
            }
        }
#pragma unroll
        for (int l = 0; l < 2; l++) {
            if (sizeof(T) == 2) {
                ncol[l] = mk3((T)acc_xy[l].x, (T)acc_xy[l].y, (T)acc_zw[l].x);
                nvar[l] = (T)acc_zw[l].y;
            }
            wv[l] = wc[l] = wsum[l];
        }
    } else {
#pragma unroll
        for (int i = -2; i <= 2; i++) {
#pragma unroll
            for (int j = -2; j <= 2; j++) {
                const int tq = (j + 2) * TW + (i + 2) * STRIDE; // the tap, relative to the thread's top-left one
                T h = hk[i < 0 ? -i : i] * hk[j < 0 ? -j : j];
                const V4<T> nd_q = t_nd[tq];
                T w_z, w_n;
                geo_weights(nd_p.w, nd_q.w, grad, cvt<T>((float)(STRIDE * i)), cvt<T>((float)(STRIDE * j)), xyz(nd_p), xyz(nd_q), w_z, w_n, w_nn, FLAT);
                T wzn = w_z * w_n;
                const V2<T> il_q = t_il[tq];
                const T iq[2] = {il_q.x, il_q.y};
                constexpr bool EAGER = LDS_COLOR && LPRT_WV_EAGER;
                const V4<T> cq_e[2] = {t_c0[EAGER ? tq : 0], t_c1[EAGER ? tq : 0]};
#pragma unroll
                for (int l = 0; l < 2; l++) {
                    float num = -(float)tabs(illum_p[l] - iq[l]);
#ifdef LPRT_EXACT_SVGF_MATH
                    T w_l = cvt<T>(svgf_exp(svgf_div(num, den_l[l])));
#else
                    T w_l = cvt<T>(__builtin_amdgcn_exp2f(num * k_l[l]));
#endif
                    T hw = h * (wzn * w_l);
                    if (tfinite(hw)) {
                        const V4<T> cq = EAGER ? cq_e[l] : LDS_COLOR ? (l == 0 ? t_c0 : t_c1)[tq] : in[l][(uint32_t)((int)p + STRIDE * (j * W + i))];
                        if (tfinite(cq.w)) {
                            nvar[l] = fmadd(hw * hw, cq.w, nvar[l]);
                            wv[l] = wv[l] + hw;
                        }
                        if (tfinite(cq.x) && tfinite(cq.y) && tfinite(cq.z)) {
                            ncol[l] = mk3(fmadd(hw, cq.x, ncol[l].x), fmadd(hw, cq.y, ncol[l].y), fmadd(hw, cq.z, ncol[l].z));
                            wc[l] = wc[l] + hw;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int l = 0; l < 2; l++) {
        T nv = tdiv(nvar[l], wv[l] * wv[l]);
        V3<T> nc = vdiv(ncol[l], wc[l]);
        V4<T> centre = LDS_COLOR ? (l == 0 ? s_c0[cy][cx] : s_c1[cy][cx]) : in[l][p];
        value_or(nc, xyz(centre));
        if (!tfinite(nv))
            nv = centre.w;
        out[l][p] = mk4(nc.x, nc.y, nc.z, nv);
        (l == 0 ? vout0 : vout1)[p] = nv;
    }
}

// ---- add_denoised_color + temporal_anti_aliasing + write_to_surface2d + copy_last_frame_pixel_id ---------------------
template <typename T>
__global__ void __launch_bounds__(256) k_finalize(SceneDev<T> sc, FrameDev<T> fb, const V4<T> *gi_colored, const V4<T> *gi_white,
                                                  V4<T> *taa_next, int add_colored, int add_white, int demodulate, T taa_w, float4 *ext_out,
                                                  uint32_t ext_first_row, uint32_t ext_rows) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= fb.win.npix())
        return;
    const T zero = (T)0, one = (T)1;
    V3<T> color = xyz(fb.color_inprogress[p]);
    V3<T> albedo = demodulate ? mk3(one, one, one) : xyz(fb.albedo[p]);
    if (add_colored)
        color = color + elemul(xyz(gi_colored[p]), albedo);
    if (add_white)
        color = color + xyz(gi_white[p]);
    // TAA (cuda.hpp:2418-2439)
    V3<T> hist = sample_linear3(fb.taa_map[p], fb.taa_history, fb.win.W, color);
    value_or(hist, color);
    V3<T> out = hist * (one - taa_w) + color * taa_w;
    V4<T> o4 = mk4(out.x, out.y, out.z, zero);
    fb.color_inprogress[p] = o4;
    taa_next[p] = o4; // history for the next frame (imgui_window.cu:260-262), double-buffered
    // display gamma x^(1/2.2) (write_to_surface2d) as exp2(log2(x) / 2.2) on the transcendental unit: relative error ~1e-6 of a
    // value that becomes 8 bits on screen (the device powf is ~170 instructions per channel; CUDA's, which the reference
    // calls, is specified to 4 ulp); 0 -> 0, 1 -> 1, inf -> inf, negative or NaN -> NaN as with powf
    const float ig = 1 / 2.2f;
    const float4 rgba = make_float4(__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)out.x) * ig), __builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)out.y) * ig),
                                    __builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)out.z) * ig), 1.f);
    fb.output[p] = rgba;
    if (ext_out) { // this rank's rows of the gathered frame (sharded rendering)
        const uint32_t row = fb.win.row_begin + p / fb.win.W;
        if (row >= ext_first_row && row - ext_first_row < ext_rows)
            ext_out[(size_t)(row - ext_first_row) * fb.win.W + p % fb.win.W] = rgba;
    }
    // copy_last_frame_pixel_id (cuda.hpp:2230-2251)
    uint2 id;
    if ((fb.g_color[p] >> 24) == 0) {
        id = make_uint2(0xffffffffu, fb.last_frame_id[p].y);
    } else {
        id = make_uint2(sc.objects[fb.g_objid[p]].objectid, fb.g_offset[p]);
    }
    fb.last_frame_id[p] = id;
    fb.prev_frame_count[p] = fb.valid_frame_count[p]; // snapshot for the next frame's temporal map
}

// copy_last_frame_transform (cuda.hpp:2253-2263).  The slot is indexed by objectid (the VAO), so instances that share a mesh
// write the same slot: in the reference that is a race between threads (whichever instance's store lands last is the
// "last frame transform" of all of them); the oracle's sequential loop makes the LAST instance in index order win, and so
// does this kernel -- a thread stores only if no later instance names its VAO (n <= 1024 instances: the scan is nothing).
// Found by tests/test_gpu_frame.py::test_launch_structure_never_changes_a_frame: the 150-instance field rendered its
// second frame differently from run to run.
template <typename T> __global__ void k_copy_last_transform(SceneDev<T> sc, M4<T> *last_L2W, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const uint32_t id = sc.objects[i].objectid;
    for (uint32_t j = i + 1; j < n; j++)
        if (sc.objects[j].objectid == id)
            return;
    last_L2W[id] = sc.objects[i].L2W;
}

// float32 read-back conversions
template <typename T> __global__ void k_read_v4(const V4<T> *src, float *dst, uint32_t n, int channels) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n)
        return;
    V4<T> v = src[p];
    float f[4] = {(float)v.x, (float)v.y, (float)v.z, (float)v.w};
    for (int c = 0; c < channels; c++)
        dst[(size_t)p * channels + c] = f[c];
}
__global__ void k_read_u32(const uint32_t *src, float *dst, uint32_t n, int rgba8) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n)
        return;
    uint32_t v = src[p];
    if (rgba8) {
        for (int c = 0; c < 4; c++)
            dst[(size_t)p * 4 + c] = (float)((v >> (8 * c)) & 255u);
    } else {
        dst[p] = (float)v;
    }
}

__global__ void k_read_u8(const uint8_t *src, float *dst, uint32_t n) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p < n)
        dst[p] = (float)src[p];
}

// ---- launchers -------------------------------------------------------------------------------------------------------
static inline dim3 grid_for(uint32_t n) { return dim3((n + 255) / 256); }

template <typename T> void launch_depth_gradient(hipStream_t st, const FrameDev<T> &fb) {
    hipLaunchKernelGGL(k_depth_gradient<T>, grid_for(fb.win.npix()), dim3(256), 0, st, fb);
}
template <typename T> void launch_temporal_color(hipStream_t st, const FrameDev<T> &fb, float color_w, float moments_w, bool always_test) {
    dim3 grid((fb.win.W + TC_BX - 1) / TC_BX, (fb.win.rows() + TC_BY - 1) / TC_BY);
    hipLaunchKernelGGL(k_temporal_color<T>, grid, dim3(256), 0, st, fb, (T)color_w, (T)moments_w, always_test ? 1 : 0);
}
template <typename T> void launch_moments(hipStream_t st, const FrameDev<T> &fb, float moments_w) {
    (void)moments_w; // the weight is only used by the history branch, which runs in k_temporal_color
    hipLaunchKernelGGL(k_moments<T>, grid_for(fb.win.npix()), dim3(256), 0, st, fb);
}
// level = 0..4 -> stride 1,2,4,8,16.  Buffer rotation (cuda.hpp:1678-1710 with cv_c in the role of color_inout, which
// holds the integrated colour): L1 c->b (b = next frame's history), L2 b->a, L3 a->c, L4 c->a, L5 a->c; result in cv_c.
// row_lo / row_hi: window rows whose results the rest of the frame uses (the whole window, or the strip a rank owns): level l
// is needed within 64 - 2^(l+2) rows of them -- the taps of the levels after it -- and nowhere else.
template <typename T> void launch_wavelet(hipStream_t st, const FrameDev<T> &fb, int level, bool always_test, uint32_t row_lo, uint32_t row_hi) {
    const uint32_t stride = 1u << level, reach = 64u - (4u << level), R = fb.win.rows();
    const uint32_t lo = row_lo > reach ? row_lo - reach : 0u, hi = std::min(R, row_hi + reach);
    const uint32_t lat0 = lo / stride;                                   // the same first lattice row for every phase
    const uint32_t lattice_rows = (hi - lat0 * stride + stride - 1) / stride; // per phase (upper bound)
    constexpr int WV_BY = WvRows<T>::value;
    const dim3 g((fb.win.W + WV_BX - 1) / WV_BX, stride * ((lattice_rows + WV_BY - 1) / WV_BY)), b(WV_BX * WV_BY);
    const SvgfLobe<T> &a = fb.lobe[0], &w = fb.lobe[1];
    const int at = always_test ? 1 : 0;
    switch (level) {
    case 0: hipLaunchKernelGGL((k_wavelet<T, 1>), g, b, 0, st, fb, a.cv_c, a.cv_b, w.cv_c, w.cv_b, a.var_c, a.var_b, w.var_c, w.var_b, at, (int)lat0, (int)hi); break;
    case 1: hipLaunchKernelGGL((k_wavelet<T, 2>), g, b, 0, st, fb, a.cv_b, a.cv_a, w.cv_b, w.cv_a, a.var_b, a.var_a, w.var_b, w.var_a, at, (int)lat0, (int)hi); break;
    case 2: hipLaunchKernelGGL((k_wavelet<T, 4>), g, b, 0, st, fb, a.cv_a, a.cv_c, w.cv_a, w.cv_c, a.var_a, a.var_c, w.var_a, w.var_c, at, (int)lat0, (int)hi); break;
    case 3: hipLaunchKernelGGL((k_wavelet<T, 8>), g, b, 0, st, fb, a.cv_c, a.cv_a, w.cv_c, w.cv_a, a.var_c, a.var_a, w.var_c, w.var_a, at, (int)lat0, (int)hi); break;
    default: hipLaunchKernelGGL((k_wavelet<T, 16>), g, b, 0, st, fb, a.cv_a, a.cv_c, w.cv_a, w.cv_c, a.var_a, a.var_c, w.var_a, w.var_c, at, (int)lat0, (int)hi); break;
    }
}
template <typename T>
void launch_finalize(hipStream_t st, const SceneDev<T> &sc, const FrameDev<T> &fb, const V4<T> *gi_colored, const V4<T> *gi_white, V4<T> *taa_next,
                     bool add_colored, bool add_white, bool demodulate, float taa_w, float4 *ext_out, uint32_t ext_first_row, uint32_t ext_rows) {
    hipLaunchKernelGGL(k_finalize<T>, grid_for(fb.win.npix()), dim3(256), 0, st, sc, fb, gi_colored, gi_white, taa_next, add_colored ? 1 : 0,
                       add_white ? 1 : 0, demodulate ? 1 : 0, (T)taa_w, ext_out, ext_first_row, ext_rows);
}
template <typename T> void launch_copy_last_transform(hipStream_t st, const SceneDev<T> &sc, M4<T> *last_L2W, uint32_t n_objects) {
    if (n_objects)
        hipLaunchKernelGGL(k_copy_last_transform<T>, dim3((n_objects + 63) / 64), dim3(64), 0, st, sc, last_L2W, n_objects);
}
template <typename T> void launch_read_v4(hipStream_t st, const V4<T> *src, float *dst, uint32_t n, int channels) {
    hipLaunchKernelGGL(k_read_v4<T>, grid_for(n), dim3(256), 0, st, src, dst, n, channels);
}
void launch_read_u32(hipStream_t st, const uint32_t *src, float *dst, uint32_t n, bool rgba8) {
    hipLaunchKernelGGL(k_read_u32, grid_for(n), dim3(256), 0, st, src, dst, n, rgba8 ? 1 : 0);
}

void launch_read_u8(hipStream_t st, const uint8_t *src, float *dst, uint32_t n) {
    hipLaunchKernelGGL(k_read_u8, grid_for(n), dim3(256), 0, st, src, dst, n);
}

#define INST(T)                                                                                                        \
    template void launch_depth_gradient<T>(hipStream_t, const FrameDev<T> &);                                          \
    template void launch_temporal_color<T>(hipStream_t, const FrameDev<T> &, float, float, bool);                      \
    template void launch_moments<T>(hipStream_t, const FrameDev<T> &, float);                                          \
    template void launch_wavelet<T>(hipStream_t, const FrameDev<T> &, int, bool, uint32_t, uint32_t);                                      \
    template void launch_finalize<T>(hipStream_t, const SceneDev<T> &, const FrameDev<T> &, const V4<T> *,             \
                                     const V4<T> *, V4<T> *, bool, bool, bool, float, float4 *, uint32_t, uint32_t);  \
    template void launch_copy_last_transform<T>(hipStream_t, const SceneDev<T> &, M4<T> *, uint32_t);                  \
    template void launch_read_v4<T>(hipStream_t, const V4<T> *, float *, uint32_t, int);
INST(float)
INST(h16)
#undef INST

} // namespace lprt
