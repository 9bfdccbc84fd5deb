// curand_init(seed, subsequence = pixel index, offset = 0) for every pixel of the window, without the
// per-thread skip-ahead the reference pays in every shade launch (rt/rtrt/cuda.hpp:1763-1764, 2353-2354).
//
// XORWOW's xorshift state v (160 bits) evolves linearly over GF(2); one subsequence = 2^67 steps = one fixed
// 160x160 matrix M, so state(n) = M^n * v0(seed).  (The Weyl counter d is untouched: 2^67 * 362437 = 0 mod 2^32.)
// Writing n in base 64, n = 64^3*j3 + 64^2*j2 + 64*j1 + j0:
//     state(n) = M^j0 * (M64^j1 * (M4096^j2 * W3[j3])),   W3[j3] = M^(64^3 * j3) * v0   (a handful, on the host)
// Every level is the same kernel: dst[s*64 + j] = P_level[j] * src[s], with the 64 matrices P_level[j] FIXED
// (seed independent, 3 x 64 x 5 KB in HBM, L2 resident) and j WAVE-UNIFORM, so the matrix streams through
// scalar loads while the 64 lanes hold 64 different source vectors: one matvec per pixel, ~1.8k VALU ops,
// instead of ~16 dependent matvecs with per-lane matrices.
#include "kernels.hpp"
#include <algorithm>
#include <mutex>
#include <string>

namespace lprt {

// matrix image in HBM: column i (image of basis vector e_i) padded to 8 words -> one 32-B scalar load
struct XorMat {
    uint32_t col[160][8];
};

// level-3 source vectors (host-computed), passed by value: no copy, no sync.  Up to RNG_MAX_BATCH seeds per launch
// (gridDim.z) x up to 16 vectors each (frames up to 4.19 Mpixel); larger frames go one seed per launch (64 vectors).
constexpr int RNG_MAX_BATCH = 8;
struct TopBatch {
    uint32_t v[RNG_MAX_BATCH][5][16];
};
struct TopVectors {
    uint32_t v[5][64];
};

// dst[(s*64 + j) - dst_first] = P[j] * src[s].  One block = 128 consecutive s x 16 consecutive j; a wave owns 4 j and
// every lane carries TWO source vectors (s, s + 64), so each column read from LDS serves two matvecs.
// The wave's current matrix (5 KB) is staged in LDS with vector loads and its columns are then read back as
// wave-uniform broadcasts: scalar loads of the columns (the obvious form) are limited by scalar-cache fills.
// acc ^= column & mask is ONE v_bitop3_b32 (truth table 0x78 = a ^ (b & c)).
// Results leave through registers: a lane keeps the 4 consecutive destinations (its wave's 4 j) of each of its two
// sources and stores them as one 16-byte vector per state word -- the four waves of a block complete the 64-byte line.
// (Round 1 transposed them through a 43.5-KB LDS tile for 64-byte runs per source; with the 20 KB of staged matrices that
// was 64 KB per workgroup = two workgroups per CU = TWO waves per SIMD for a kernel that does nothing but issue
// v_bitop3: rocprofv3 showed SQ_ACTIVE_INST_VALU at 0.49 of the wave cycles, half the issue rate.  Without the tile the
// limit is the registers: 4 waves per SIMD.)
// blockIdx.z = seed of a batch: src / dst planes of seed z start at z * src_seed_stride / z * dst_seed_stride.
constexpr int RNG_S = 128; // sources per block
LPRT_DEV uint32_t xor_and(uint32_t acc, uint32_t col, uint32_t mask) { return __builtin_amdgcn_bitop3_b32(acc, col, mask, 0x78); }
// JW = matrices per wave: 4 for the last level (a column read serves eight matvecs' worth of stores per lane), 1 for the two
// small upper levels, whose few sources leave most lanes idle and whose cost is the LATENCY of one wave working through its
// matrices one after the other -- three dependent launches of ~40 us each were a fixed 120 us per frame, 22 % of the frame
// of a rank that renders an eighth of the image.
// LIST: the sources are the entries of `runs` (runs[0] of them, k_rng_live_runs) instead of 0 .. n_src - 1
template <int TOP, int JW, bool LIST = false>
__global__ void __launch_bounds__(256, 4) k_rng_expand(const XorMat *__restrict__ P, const uint32_t *__restrict__ src, TopBatch topb, TopVectors top1,
                                                       uint32_t n_src, uint32_t src_stride, uint32_t src_seed_stride, uint32_t *__restrict__ dst,
                                                       uint32_t dst_first, uint32_t dst_count, uint32_t dst_stride, uint32_t dst_seed_stride,
                                                       const uint32_t *__restrict__ runs = nullptr) {
    __shared__ uint4 mat[4][160 * 2]; // [wave][column][8 words]
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t sA = blockIdx.x * RNG_S + lane, sB = sA + 64;
    if (LIST) {
        const uint32_t n_runs = runs[0];
        if (blockIdx.x * RNG_S >= n_runs)
            return; // (the grid is sized for every run of the range)
        sA = sA < n_runs ? runs[1u + sA] : 0xffffffffu;
        sB = sB < n_runs ? runs[1u + sB] : 0xffffffffu;
    }
    const uint32_t jbase = blockIdx.y * (4 * JW);
    const uint32_t z = blockIdx.z;
    src += (size_t)z * src_seed_stride;
    dst += (size_t)z * dst_seed_stride;
    uint32_t va[5] = {0, 0, 0, 0, 0}, vb[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < 5; k++) {
        if (sA < n_src)
            va[k] = TOP == 2 ? topb.v[z][k][sA & 15] : (TOP == 1 ? top1.v[k][sA & 63] : src[k * src_stride + sA]);
        if (sB < n_src)
            vb[k] = TOP == 2 ? topb.v[z][k][sB & 15] : (TOP == 1 ? top1.v[k][sB & 63] : src[k * src_stride + sB]);
    }
    uint32_t ra[5][JW], rb[5][JW]; // [state word][j of this wave]
#pragma unroll
    for (uint32_t jj = 0; jj < JW; jj++) {
        const uint32_t jl = wave * JW + jj; // 0 .. 4 JW - 1 within the block
        const uint4 *m = reinterpret_cast<const uint4 *>(P + jbase + jl);
        for (uint32_t q = lane; q < 320; q += 64)
            mat[wave][q] = m[q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        uint32_t a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0;
#pragma unroll
        for (int w = 0; w < 5; w++) {
            const uint32_t bitsa = va[w], bitsb = vb[w];
#pragma unroll 8
            for (int b = 0; b < 32; b++) {
                const uint4 c03 = mat[wave][2 * (w * 32 + b)];
                const uint32_t c4 = mat[wave][2 * (w * 32 + b) + 1].x;
                const uint32_t ma = (uint32_t)__builtin_amdgcn_sbfe((int)bitsa, b, 1); // 0 or ~0
                const uint32_t mb = (uint32_t)__builtin_amdgcn_sbfe((int)bitsb, b, 1);
                a0 = xor_and(a0, c03.x, ma);
                a1 = xor_and(a1, c03.y, ma);
                a2 = xor_and(a2, c03.z, ma);
                a3 = xor_and(a3, c03.w, ma);
                a4 = xor_and(a4, c4, ma);
                b0 = xor_and(b0, c03.x, mb);
                b1 = xor_and(b1, c03.y, mb);
                b2 = xor_and(b2, c03.z, mb);
                b3 = xor_and(b3, c03.w, mb);
                b4 = xor_and(b4, c4, mb);
            }
        }
        ra[0][jj] = a0; ra[1][jj] = a1; ra[2][jj] = a2; ra[3][jj] = a3; ra[4][jj] = a4;
        rb[0][jj] = b0; rb[1][jj] = b1; rb[2][jj] = b2; rb[3][jj] = b3; rb[4][jj] = b4;
        __builtin_amdgcn_wave_barrier(); // the next staging overwrites mat[wave]: all lanes must be done reading
    }
    // write-out: destinations n0 .. n0+3 of a source are consecutive; one 16-byte store per state word when the run lies
    // inside [dst_first, dst_first + dst_count) and starts on a 16-byte boundary of its plane, single words otherwise
    const bool planes_aligned = ((reinterpret_cast<uintptr_t>(dst) | ((size_t)dst_stride * 4u)) & 15u) == 0u;
#pragma unroll
    for (int half = 0; half < 2; half++) {
        const uint32_t ss = half == 0 ? sA : sB;
        if (ss >= n_src)
            continue;
        const uint32_t n0 = ss * 64 + jbase + wave * JW;
        const bool inside = n0 >= dst_first && n0 - dst_first + (uint32_t)JW <= dst_count;
        if (JW == 4 && inside && planes_aligned && ((n0 - dst_first) & 3u) == 0u) {
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const uint32_t *r = half == 0 ? ra[k] : rb[k];
                *reinterpret_cast<uint4 *>(dst + (size_t)k * dst_stride + (n0 - dst_first)) = make_uint4(r[0], r[JW > 1 ? 1 : 0], r[JW > 2 ? 2 : 0], r[JW > 3 ? 3 : 0]);
            }
        } else if (JW == 2 && inside && planes_aligned && ((n0 - dst_first) & 1u) == 0u) {
#pragma unroll
            for (int k = 0; k < 5; k++) {
                const uint32_t *r = half == 0 ? ra[k] : rb[k];
                *reinterpret_cast<uint2 *>(dst + (size_t)k * dst_stride + (n0 - dst_first)) = make_uint2(r[0], r[JW > 1 ? 1 : 0]);
            }
        } else {
#pragma unroll
            for (uint32_t jj = 0; jj < JW; jj++) {
                const uint32_t n = n0 + jj;
                if (n < dst_first || n - dst_first >= dst_count)
                    continue;
#pragma unroll
                for (int k = 0; k < 5; k++)
                    dst[(size_t)k * dst_stride + (n - dst_first)] = half == 0 ? ra[k][jj] : rb[k][jj];
            }
        }
    }
}

// runs[1 ..] = the runs r (64 pixel indices each, run 0 = global pixels 64 * s_begin ..) of [0, n_runs) that hold a covered pixel
// of the range [first, first + count) (g_color: the range's plane, alpha != 0 = covered), runs[0] = their number (zeroed
// before the launch).  One wave per run and round, 8 rounds: a workgroup collects its up to 128 live runs in LDS and appends
// them with ONE atomic -- 2,025 atomics on the one counter (a workgroup per 16 runs) took 26 us of a 1080p frame.
constexpr uint32_t RNG_RUN_ROUNDS = 8;
__global__ void __launch_bounds__(1024) k_rng_live_runs(const uint32_t *__restrict__ g_color, uint32_t first, uint32_t count, uint32_t s_begin, uint32_t n_runs,
                                                        uint32_t *__restrict__ runs) {
    __shared__ uint32_t s_list[16 * RNG_RUN_ROUNDS], s_n, s_base;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0)
        s_n = 0u;
    __syncthreads();
#pragma unroll
    for (uint32_t it = 0; it < RNG_RUN_ROUNDS; it++) { // (the eight rounds' loads are independent: in flight together)
        const uint32_t r = (blockIdx.x * RNG_RUN_ROUNDS + it) * 16u + wave, n = (s_begin + r) * 64u + lane;
        const bool covered = r < n_runs && n >= first && n - first < count && (g_color[n - first] >> 24) != 0u;
        if (__builtin_amdgcn_ballot_w64(covered) != 0ull && lane == 0)
            s_list[atomicAdd(&s_n, 1u)] = r;
    }
    __syncthreads();
    if (threadIdx.x == 0)
        s_base = s_n ? atomicAdd(&runs[0], s_n) : 0u;
    __syncthreads();
    if (threadIdx.x < s_n)
        runs[1u + s_base + threadIdx.x] = s_list[threadIdx.x];
}

// ---- host-side GF(2) algebra for the fixed matrices ---------------------------------------------------------------------
namespace {
struct HMat {
    uint32_t col[160][5];
};
inline void h_apply(const HMat &m, const uint32_t in[5], uint32_t out[5]) {
    uint32_t r[5] = {0, 0, 0, 0, 0};
    for (int w = 0; w < 5; w++) {
        uint32_t bits = in[w];
        while (bits) {
            int b = __builtin_ctz(bits);
            bits &= bits - 1;
            const uint32_t *c = m.col[w * 32 + b];
            r[0] ^= c[0]; r[1] ^= c[1]; r[2] ^= c[2]; r[3] ^= c[3]; r[4] ^= c[4];
        }
    }
    for (int k = 0; k < 5; k++)
        out[k] = r[k];
}
inline HMat h_mul(const HMat &a, const HMat &b) { // apply b first, then a
    HMat c;
    for (int i = 0; i < 160; i++)
        h_apply(a, b.col[i], c.col[i]);
    return c;
}
inline HMat h_identity() {
    HMat m{};
    for (int i = 0; i < 160; i++)
        m.col[i][i / 32] = 1u << (i % 32);
    return m;
}
struct Tables {
    std::vector<XorMat> levels; // [3][64]
    HMat m_top;                 // M^(64^3)
};
const Tables &tables() {
    static Tables t;
    static std::once_flag once;
    std::call_once(once, [] {
        // one xorwow step as a matrix
        HMat a{};
        for (int i = 0; i < 160; i++) {
            uint32_t v[5] = {0, 0, 0, 0, 0};
            v[i / 32] = 1u << (i % 32);
            uint32_t tt = v[0] ^ (v[0] >> 2);
            uint32_t n4 = (v[4] ^ (v[4] << 4)) ^ (tt ^ (tt << 1));
            a.col[i][0] = v[1];
            a.col[i][1] = v[2];
            a.col[i][2] = v[3];
            a.col[i][3] = v[4];
            a.col[i][4] = n4;
        }
        for (int i = 0; i < 67; i++)
            a = h_mul(a, a); // M = A^(2^67)
        t.levels.resize(3 * 64);
        HMat base = a;
        for (int level = 0; level < 3; level++) {
            HMat p = h_identity();
            for (int j = 0; j < 64; j++) {
                XorMat &x = t.levels[level * 64 + j];
                for (int i = 0; i < 160; i++) {
                    for (int k = 0; k < 5; k++)
                        x.col[i][k] = p.col[i][k];
                    x.col[i][5] = x.col[i][6] = x.col[i][7] = 0;
                }
                p = h_mul(base, p);
            }
            base = p; // base^64
        }
        t.m_top = base;
    });
    return t;
}
} // namespace

void RngStore::init(hipStream_t st) {
    const Tables &t = tables();
    d_mats.reserve(sizeof(XorMat) * t.levels.size());
    LPRT_HIP(hipMemcpyAsync(d_mats.ptr, t.levels.data(), sizeof(XorMat) * t.levels.size(), hipMemcpyHostToDevice, st));
    LPRT_HIP(hipStreamSynchronize(st));
    ready = true;
}

static inline void seed_state(uint64_t seed, uint32_t v[5], uint32_t &d) { // curand_init seed scramble
    uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u, s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
    uint32_t t0 = 1099087573u * s0, t1 = 2591861531u * s1;
    d = 6615241u + t1 + t0;
    v[0] = 123456789u + t0;
    v[1] = 362436069u ^ t0;
    v[2] = 521288629u + t1;
    v[3] = 88675123u ^ t1;
    v[4] = 5783321u + t0;
}

// fills planes[i] ([5][count] each, seed stride 5*count) with the states of pixels [first, first+count) for every
// seed of the batch; d_out[i] = d(seed i).  No host synchronisation.
void RngStore::generate(hipStream_t st, const uint64_t *seeds, uint32_t n_seeds, uint32_t total_pixels, uint32_t first, uint32_t count,
                        uint32_t *planes, uint32_t *d_out) {
    generate(st, seeds, n_seeds, n_seeds, total_pixels, first, count, first, count, planes, d_out);
}

// Two pixel ranges in one batch: the first n_head seeds cover [first, first + count), the others [first2, first2 + count2)
// (sharded frames: the TAA jitter seed covers a rank's whole window, the per-sample seeds only the rows it traces).  The
// upper expansion levels do not depend on the range and run once for all seeds; planes = the head seeds' planes
// ([5][count] each), then the others' ([5][count2] each).
void RngStore::generate(hipStream_t st, const uint64_t *seeds, uint32_t n_seeds, uint32_t n_head, uint32_t total_pixels, uint32_t first,
                        uint32_t count, uint32_t first2, uint32_t count2, uint32_t *planes, uint32_t *d_out, const uint32_t *g_color) {
    const Tables &t = tables();
    const uint32_t n1 = (total_pixels + 63) / 64, n2 = (n1 + 63) / 64, n3 = (n2 + 63) / 64;
    if (n3 > 64)
        throw std::string("frame too large for the RNG expansion (more than 2^24 pixels)");
    const XorMat *mats = d_mats.as<XorMat>();
    const uint32_t w2_stride = n3 * 64, w1_stride = n2 * 64;
    if (first == first2 && count == count2)
        n_head = n_seeds; // one range: one last-level launch per batch
    else
        g_color = nullptr;
    const uint32_t *runs = nullptr;
    if (g_color && count != 0) {
        const uint32_t s_begin = first / 64, n_runs = (first + count + 63) / 64 - s_begin;
        d_runs.reserve(sizeof(uint32_t) * ((size_t)n_runs + 1));
        LPRT_HIP(hipMemsetAsync(d_runs.ptr, 0, sizeof(uint32_t), st));
        hipLaunchKernelGGL(k_rng_live_runs, dim3((n_runs + 16 * RNG_RUN_ROUNDS - 1) / (16 * RNG_RUN_ROUNDS)), dim3(1024), 0, st, g_color, first, count, s_begin, n_runs, d_runs.as<uint32_t>());
        runs = d_runs.as<uint32_t>();
    }
    uint32_t done = 0;
    while (done < n_seeds) {
        const uint32_t batch = n3 <= 16 ? std::min<uint32_t>(RNG_MAX_BATCH, n_seeds - done) : 1;
        d_w2.reserve(sizeof(uint32_t) * 5 * (size_t)w2_stride * batch);
        d_w1.reserve(sizeof(uint32_t) * 5 * (size_t)w1_stride * batch);
        TopBatch topb{};
        TopVectors top1{};
        for (uint32_t b = 0; b < batch; b++) {
            uint32_t v[5];
            seed_state(seeds[done + b], v, d_out[done + b]);
            for (uint32_t j = 0; j < n3; j++) {
                for (int k = 0; k < 5; k++) {
                    if (n3 <= 16)
                        topb.v[b][k][j] = v[k];
                    else
                        top1.v[k][j] = v[k];
                }
                h_apply(t.m_top, v, v);
            }
        }
        if (n3 <= 16)
            hipLaunchKernelGGL((k_rng_expand<2, 1>), dim3((n3 + RNG_S - 1) / RNG_S, 16, batch), dim3(256), 0, st, mats + 128, (const uint32_t *)nullptr, topb, top1, n3,
                               0u, 0u, d_w2.as<uint32_t>(), 0u, w2_stride, w2_stride, 5 * w2_stride);
        else
            hipLaunchKernelGGL((k_rng_expand<1, 1>), dim3((n3 + RNG_S - 1) / RNG_S, 16, batch), dim3(256), 0, st, mats + 128, (const uint32_t *)nullptr, topb, top1, n3,
                               0u, 0u, d_w2.as<uint32_t>(), 0u, w2_stride, w2_stride, 5 * w2_stride);
        hipLaunchKernelGGL((k_rng_expand<0, 1>), dim3((n2 + RNG_S - 1) / RNG_S, 16, batch), dim3(256), 0, st, mats + 64, d_w2.as<uint32_t>(), topb, top1, n2, w2_stride,
                           5 * w2_stride, d_w1.as<uint32_t>(), 0u, w1_stride, w1_stride, 5 * w1_stride);
        // last level only for the L1 vectors that cover a seed's pixel range: one launch per range present in the batch
        for (int part = 0; part < 2; part++) {
            const uint32_t z0 = part == 0 ? done : std::max(done, n_head), z1 = part == 0 ? std::min(done + batch, n_head) : done + batch;
            if (z0 >= z1)
                continue;
            const uint32_t f = part == 0 ? first : first2, c = part == 0 ? count : count2;
            if (c == 0)
                continue;
            const uint32_t s_begin = f / 64, s_end = (f + c + 63) / 64;
            uint32_t *out = part == 0 ? planes + (size_t)z0 * 5 * count : planes + (size_t)n_head * 5 * count + (size_t)(z0 - n_head) * 5 * count2;
            // matrices per wave: as many (up to 4: wider stores, fewer matrix stagings) as still leave the chip a full set of
            // workgroups -- a strip's launch is a fraction of one and then costs the latency of a wave's serial matvecs
            const uint32_t bx = (s_end - s_begin + RNG_S - 1) / RNG_S, nz = z1 - z0;
            const uint32_t *srcp = d_w1.as<uint32_t>() + (size_t)(z0 - done) * 5 * w1_stride + s_begin;
            if (runs && part == 0 && bx * 4u * nz >= 768u)
                hipLaunchKernelGGL((k_rng_expand<0, 4, true>), dim3(bx, 4, nz), dim3(256), 0, st, mats, srcp, topb, top1, s_end - s_begin, w1_stride, 5 * w1_stride,
                                   out, f - s_begin * 64, c, c, 5 * c, runs);
            else if (bx * 4u * nz >= 768u)
                hipLaunchKernelGGL((k_rng_expand<0, 4>), dim3(bx, 4, nz), dim3(256), 0, st, mats, srcp, topb, top1, s_end - s_begin, w1_stride, 5 * w1_stride, out,
                                   f - s_begin * 64, c, c, 5 * c);
            else if (bx * 8u * nz >= 768u)
                hipLaunchKernelGGL((k_rng_expand<0, 2>), dim3(bx, 8, nz), dim3(256), 0, st, mats, srcp, topb, top1, s_end - s_begin, w1_stride, 5 * w1_stride, out,
                                   f - s_begin * 64, c, c, 5 * c);
            else
                hipLaunchKernelGGL((k_rng_expand<0, 1>), dim3(bx, 16, nz), dim3(256), 0, st, mats, srcp, topb, top1, s_end - s_begin, w1_stride, 5 * w1_stride, out,
                                   f - s_begin * 64, c, c, 5 * c);
        }
        done += batch;
    }
    LPRT_HIP(hipGetLastError());
}

} // namespace lprt
