// Geometry preparation (B3) and acceleration-structure construction on the GPU (B1, B2).
//
// B3 replaces RDResource::compute_m (rt/rtrt/memory.hpp:621-648): per-triangle inverse transform
//    M_shift_f32 / M_shift and position_f32, computed on the device with the same
//    DifferenceOfProducts cofactors (rt/math/matrix.hpp:407-427, 829-843).
// B1 replaces ObjectBVH::build_object_bvh (rt/trace/object_bvh.hpp:49-134), a single-threaded CPU
//    median split, by two GPU builders over one record format (types.hpp: BlasNode, implicit heap slots):
//    a binned surface-area builder (k_sah_level: level by level from device queues, one wave or one
//    workgroup per node, stable ballot partition, no refit) for primitives large enough to have a
//    hierarchy, and the reference's rule -- object median along the widest axis -- level by level with an
//    own LSD radix sort (wave ballot ranking, no Thrust/rocPRIM) and bottom-up boxes otherwise.
// B2 replaces SceneBVH::update (rt/trace/scene_bvh.hpp:50-133), rebuilt per frame on the CPU, by a
//    single-workgroup kernel (LDS bitonic sort + the same tree).
#include "build.hpp"
#include "scene_dev.hpp"
#include <cstdlib>
#include <string>

namespace lprt {

// ---- helpers ------------------------------------------------------------------------------------
__device__ __forceinline__ int f32_ordered(float f) {
    int i = __float_as_int(f);
    return i ^ ((i >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float f32_unordered(int i) { return __int_as_float(i ^ ((i >> 31) & 0x7fffffff)); }

// round a float bound to DataT towards -inf / +inf so stored boxes always contain the f32 box (Q2)
template <typename T> __device__ __forceinline__ T round_down(float f);
template <typename T> __device__ __forceinline__ T round_up(float f);
template <> __device__ __forceinline__ float round_down<float>(float f) { return f; }
template <> __device__ __forceinline__ float round_up<float>(float f) { return f; }
__device__ __forceinline__ h16 h16_step(h16 h, bool up) { // next representable half towards +inf (up) or -inf
    unsigned short b = __builtin_bit_cast(unsigned short, h);
    if ((b & 0x7fff) == 0)
        b = up ? 0x0001 : 0x8001;
    else if (((b & 0x8000) == 0) == up)
        b = (unsigned short)(b + 1);
    else
        b = (unsigned short)(b - 1);
    return __builtin_bit_cast(h16, b);
}
template <> __device__ __forceinline__ h16 round_down<h16>(float f) {
    h16 h = (h16)f;
    if ((float)h > f)
        h = h16_step(h, false);
    return h;
}
template <> __device__ __forceinline__ h16 round_up<h16>(float f) {
    h16 h = (h16)f;
    if ((float)h < f)
        h = h16_step(h, true);
    return h;
}

__device__ __forceinline__ float diff_of_products(float a, float b, float c, float d) { // matrix.hpp:829-834
    float cd = c * d;
    float err = __builtin_fmaf(-c, d, cd);
    float dop = __builtin_fmaf(a, b, -cd);
    return dop + err;
}

// ---- B3 -----------------------------------------------------------------------------------------
template <typename T> __global__ void k_convert_vertices(const lprt_vertex *in, Vertex<T> *out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float *s = reinterpret_cast<const float *>(in + i);
    T *d = reinterpret_cast<T *>(out + i);
#pragma unroll
    for (int k = 0; k < 16; k++)
        d[k] = cvt<T>(s[k]); // float16(float), number.hpp:16
}

// one thread per triangle (global EBO order): M2 = inverse([v0-v2, v1-v2, cross(v0-v2,v1-v2) - v2])
__global__ void k_tri_prepare(const lprt_vertex *vbo, const uint32_t *ebo, uint32_t n_tris, const uint32_t *tri_prim,
                              TriRecF32 *mf32, float *tri_bounds, int *prim_cbounds) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = g < n_tris;
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    uint32_t p = 0xffffffffu;
    if (valid) {
        const float *p0 = vbo[ebo[3 * g + 0]].position;
        const float *p1 = vbo[ebo[3 * g + 1]].position;
        const float *p2 = vbo[ebo[3 * g + 2]].position;
        float v0[3] = {p0[0], p0[1], p0[2]}, v1[3] = {p1[0], p1[1], p1[2]}, v2[3] = {p2[0], p2[1], p2[2]};
        float e0[3] = {v0[0] - v2[0], v0[1] - v2[1], v0[2] - v2[2]};
        float e1[3] = {v1[0] - v2[0], v1[1] - v2[1], v1[2] - v2[2]};
        float c[3] = {diff_of_products(e0[1], e1[2], e1[1], e0[2]) - v2[0], // cross_product_difference(...) - v2 (Q1)
                      diff_of_products(e0[2], e1[0], e1[2], e0[0]) - v2[1],
                      diff_of_products(e0[0], e1[1], e1[0], e0[1]) - v2[2]};
        float a[3][3] = {{e0[0], e1[0], c[0]}, {e0[1], e1[1], c[1]}, {e0[2], e1[2], c[2]}};
        float det = a[0][0] * diff_of_products(a[1][1], a[2][2], a[2][1], a[1][2]) -
                    a[0][1] * diff_of_products(a[1][0], a[2][2], a[1][2], a[2][0]) +
                    a[0][2] * diff_of_products(a[1][0], a[2][1], a[1][1], a[2][0]);
        float invdet = 1 / det;
        TriRecF32 r;
        r.m0[0] = diff_of_products(a[1][1], a[2][2], a[2][1], a[1][2]) * invdet;
        r.m0[1] = diff_of_products(a[0][2], a[2][1], a[0][1], a[2][2]) * invdet;
        r.m0[2] = diff_of_products(a[0][1], a[1][2], a[0][2], a[1][1]) * invdet;
        r.m1[0] = diff_of_products(a[1][2], a[2][0], a[1][0], a[2][2]) * invdet;
        r.m1[1] = diff_of_products(a[0][0], a[2][2], a[0][2], a[2][0]) * invdet;
        r.m1[2] = diff_of_products(a[1][0], a[0][2], a[0][0], a[1][2]) * invdet;
        r.m2[0] = diff_of_products(a[1][0], a[2][1], a[2][0], a[1][1]) * invdet;
        r.m2[1] = diff_of_products(a[2][0], a[0][1], a[0][0], a[2][1]) * invdet;
        r.m2[2] = diff_of_products(a[0][0], a[1][1], a[1][0], a[0][1]) * invdet;
        r.v2x = v2[0];
        r.v2y = v2[1];
        r.v2z = v2[2];
        mf32[g] = r;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            lo[k] = fminf(v0[k], fminf(v1[k], v2[k]));
            hi[k] = fmaxf(v0[k], fmaxf(v1[k], v2[k]));
            tri_bounds[6 * g + k] = lo[k];
            tri_bounds[6 * g + 3 + k] = hi[k];
        }
        p = tri_prim[g];
    }
    // centroid bounds per primitive (normalisation box of the Morton codes): wave reduction when the whole
    // wave belongs to one primitive (the common case), per-lane atomics otherwise
    float cen[3] = {0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2])};
    // Then the workgroup's waves of one primitive are merged in LDS and ONE of them adds: the atomics of a primitive all land on
    // the same six words, where the L2 serves them one after the other (the atrium: 4,100 waves on 24 primitives).
    __shared__ uint32_t s_prim[16];
    __shared__ float s_box[16][6];
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
    uint32_t p0w = __shfl(p, 0, 64);
    const bool uniform = __all(p == p0w) != 0;
    float mn[3] = {cen[0], cen[1], cen[2]}, mx[3] = {cen[0], cen[1], cen[2]};
    if (uniform && p0w != 0xffffffffu)
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int k = 0; k < 3; k++) {
                mn[k] = fminf(mn[k], __shfl_xor(mn[k], off, 64));
                mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off, 64));
            }
    if (lane == 0u) {
        s_prim[wv] = uniform ? p0w : 0xffffffffu; // (a wave across a primitive boundary, or past the end: nothing to merge)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            s_box[wv][k] = mn[k];
            s_box[wv][3 + k] = mx[k];
        }
    }
    __syncthreads();
    if (uniform) {
        if (p0w == 0xffffffffu)
            return;
        // the first wave of each run of equal primitives in the workgroup adds for the run
        if (lane == 0u && (wv == 0u || s_prim[wv - 1u] != p0w)) {
            for (uint32_t q = wv + 1u; q < n_waves && s_prim[q] == p0w; q++)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    mn[k] = fminf(mn[k], s_box[q][k]);
                    mx[k] = fmaxf(mx[k], s_box[q][3 + k]);
                }
#pragma unroll
            for (int k = 0; k < 3; k++) {
                atomicMin(prim_cbounds + 6 * p0w + k, f32_ordered(mn[k]));
                atomicMax(prim_cbounds + 6 * p0w + 3 + k, f32_ordered(mx[k]));
            }
        }
    } else if (valid) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            atomicMin(prim_cbounds + 6 * p + k, f32_ordered(cen[k]));
            atomicMax(prim_cbounds + 6 * p + 3 + k, f32_ordered(cen[k]));
        }
    }
}

__device__ __forceinline__ uint32_t expand_bits10(uint32_t v) { // 10 bits -> every third bit
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t morton30(float x, float y, float z) { // x,y,z in [0,1]
    uint32_t xi = (uint32_t)fminf(fmaxf(x * 1024.0f, 0.0f), 1023.0f);
    uint32_t yi = (uint32_t)fminf(fmaxf(y * 1024.0f, 0.0f), 1023.0f);
    uint32_t zi = (uint32_t)fminf(fmaxf(z * 1024.0f, 0.0f), 1023.0f);
    return (expand_bits10(xi) << 2) | (expand_bits10(yi) << 1) | expand_bits10(zi);
}

__global__ void k_morton_keys(const float *tri_bounds, const uint32_t *tri_prim, const int *prim_cbounds, uint32_t n_tris,
                              unsigned long long *keys, uint32_t *vals) {
    uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_tris)
        return;
    uint32_t p = tri_prim[g];
    float q[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float lo = f32_unordered(prim_cbounds[6 * p + k]), hi = f32_unordered(prim_cbounds[6 * p + 3 + k]);
        float cen = 0.5f * (tri_bounds[6 * g + k] + tri_bounds[6 * g + 3 + k]);
        float ext = hi - lo;
        q[k] = ext > 0.0f ? (cen - lo) / ext : 0.0f;
    }
    keys[g] = ((unsigned long long)p << 32) | morton30(q[0], q[1], q[2]);
    vals[g] = g;
}

// ---- LSD radix sort: 8-bit digits, stable, (u64 key, u32 value) ---------------------------------------
constexpr int RS_THREADS = 256;
constexpr int RS_KPT = 8; // keys per thread
constexpr int RS_TILE = RS_THREADS * RS_KPT;

__global__ void k_radix_hist(const unsigned long long *keys, uint32_t n, int shift, uint32_t *hist, uint32_t n_blocks) {
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    uint32_t base = blockIdx.x * RS_TILE;
    for (int r = 0; r < RS_KPT; r++) {
        uint32_t i = base + r * RS_THREADS + threadIdx.x;
        if (i < n)
            atomicAdd(&h[(keys[i] >> shift) & 0xff], 1u);
    }
    __syncthreads();
    hist[threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x]; // digit-major: a global scan yields scatter bases
}

// exclusive scan of `total` counters (256 * n_blocks, a multiple of 4) by one workgroup of 1024 threads: tiles of 16 K
// counters, 16 consecutive ones per thread (four 16-B loads), wave scan by shuffles, the 16 wave totals through LDS.
// 32 K counters (the 262 k-triangle atrium) are two tiles; the former one-element-per-thread Hillis-Steele loop took
// 136 us there, two thirds of the whole BLAS build.
constexpr int SCAN_THREADS = 1024, SCAN_PER_THREAD = 16;
__global__ void __launch_bounds__(SCAN_THREADS) k_exclusive_scan(uint32_t *data, uint32_t total) {
    __shared__ uint32_t wsum[SCAN_THREADS / 64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < total; base += SCAN_THREADS * SCAN_PER_THREAD) {
        const uint32_t i0 = base + SCAN_PER_THREAD * threadIdx.x;
        uint32_t v[SCAN_PER_THREAD];
#pragma unroll
        for (int j = 0; j < SCAN_PER_THREAD; j += 4) {
            uint4 q = make_uint4(0u, 0u, 0u, 0u);
            if (i0 + j < total) // total % 4 == 0 and i0 + j % 4 == 0: a whole quad or nothing
                q = *reinterpret_cast<const uint4 *>(data + i0 + j);
            v[j] = q.x, v[j + 1] = q.y, v[j + 2] = q.z, v[j + 3] = q.w;
        }
        uint32_t sum = 0;
#pragma unroll
        for (int j = 0; j < SCAN_PER_THREAD; j++) { // v[j] <- exclusive prefix within the thread
            const uint32_t x = v[j];
            v[j] = sum;
            sum += x;
        }
        uint32_t incl = sum;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= (uint32_t)off)
                incl += up;
        }
        if (lane == 63)
            wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0, tile = 0;
#pragma unroll
        for (uint32_t w = 0; w < SCAN_THREADS / 64; w++) {
            const uint32_t x = wsum[w];
            before += w < wave ? x : 0u;
            tile += x;
        }
        const uint32_t excl = carry + before + incl - sum;
#pragma unroll
        for (int j = 0; j < SCAN_PER_THREAD; j += 4)
            if (i0 + j < total)
                *reinterpret_cast<uint4 *>(data + i0 + j) = make_uint4(excl + v[j], excl + v[j + 1], excl + v[j + 2], excl + v[j + 3]);
        carry += tile;
        __syncthreads(); // wsum is rewritten by the next tile
    }
}

// Stable scatter.  Wave w of a block owns keys [w*64*KPT, (w+1)*64*KPT) of the tile, visited in rounds of
// 64 consecutive keys.  Within a round, lanes holding the same digit find each other with 8 ballots; the
// lowest such lane reserves their run in the wave's digit counter (LDS) — wavefront ballot + prefix.
__global__ void k_radix_scatter(const unsigned long long *keys_in, const uint32_t *vals_in, unsigned long long *keys_out,
                                uint32_t *vals_out, uint32_t n, int shift, const uint32_t *hist, uint32_t n_blocks) {
    constexpr int WAVES = RS_THREADS / 64;
    __shared__ uint32_t wcount[WAVES][256];
    __shared__ uint32_t wbase[WAVES][256];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int d = lane; d < 256; d += 64)
        wcount[wave][d] = 0;
    __syncthreads();
    unsigned long long key[RS_KPT];
    uint32_t val[RS_KPT], rank[RS_KPT];
    const uint32_t base = blockIdx.x * RS_TILE + wave * 64 * RS_KPT;
    const unsigned long long lanes_below = (1ull << lane) - 1ull;
#pragma unroll
    for (int r = 0; r < RS_KPT; r++) {
        uint32_t i = base + r * 64 + lane;
        bool valid = i < n;
        key[r] = valid ? keys_in[i] : ~0ull;
        val[r] = valid ? vals_in[i] : 0u;
        uint32_t digit = (uint32_t)(key[r] >> shift) & 0xffu;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            unsigned long long m = __ballot((digit >> b) & 1u);
            peers &= ((digit >> b) & 1u) ? m : ~m;
        }
        uint32_t before = (uint32_t)__popcll(peers & lanes_below);
        uint32_t start = 0;
        if (valid && before == 0) { // leader of the run
            start = wcount[wave][digit];
            wcount[wave][digit] = start + (uint32_t)__popcll(peers);
        }
        int leader = valid ? __builtin_ctzll(peers) : (int)lane;
        start = __shfl(start, leader, 64);
        rank[r] = start + before;
        // LDS counter updates of this round must be visible to the next round's leaders (same wave: in order)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    __syncthreads();
    // per-digit base of each wave = global base of (digit, block) + counts of the waves before it
    for (int d = threadIdx.x; d < 256; d += RS_THREADS) {
        uint32_t run = hist[d * n_blocks + blockIdx.x];
        for (int w = 0; w < WAVES; w++) {
            wbase[w][d] = run;
            run += wcount[w][d];
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_KPT; r++) {
        uint32_t i = base + r * 64 + lane;
        if (i < n) {
            uint32_t digit = (uint32_t)(key[r] >> shift) & 0xffu;
            uint32_t dst = wbase[wave][digit] + rank[r];
            keys_out[dst] = key[r];
            vals_out[dst] = val[r];
        }
    }
}

// ---- median-split ordering (default BLAS builder) -----------------------------------------------------------------
// The reference splits every node at the object median along the longest axis of its bounds
// (rt/trace/object_bvh.hpp:111-125).  The same hierarchy is produced here level by level: at depth d every node's
// range is sorted along the node's own longest axis (16-bit quantised centroid key, stable radix sort with the
// node's start position as the high digits, so elements never leave their node); the implicit tree then splits each
// range at l + (r-l)/2.  A plain Morton order (k_morton_keys, LPRT_BLAS_BUILDER=morton) builds 3x faster but its
// index-median splits cut the Z-curve at arbitrary places: sibling boxes overlap and traversal visited 3.3x more
// boxes on the 262k-triangle atrium.
__host__ __device__ inline void segment_at(uint32_t q, uint32_t d, uint32_t n, uint32_t &l, uint32_t &r) {
    l = 0;
    r = n;
    for (uint32_t b = 0; b < d; b++) {
        uint32_t mid = l + (r - l) / 2;
        if (q < mid)
            r = mid;
        else
            l = mid;
    }
}
// ---- the lower levels of the median-split ordering, one workgroup per subtree -----------------------------------------
// Once every node of a level holds at most LOCAL_MAX triangles, the rest of that subtree's ordering needs only the
// subtree's own elements: one workgroup keeps them in LDS and runs the remaining levels there — per level the bounds of
// every sub-node (LDS atomics in order-preserving int space), the same 16-bit key along the sub-node's longest axis as
// k_seg_keys, and a bitonic sort of the composite key (sub-node | key | current position).  The position in the low bits
// makes the order total, so the result is exactly the stable sort the global radix passes produce: the permutation is
// bit-identical to the all-global build (LPRT_BLAS_LOCAL=0), at one launch instead of 18 per level.
constexpr uint32_t LOCAL_MAX = 1024;
struct LocalJob {
    uint32_t prim, first, count, depth0; // triangles [first, first + count) of the primitive's sorted range; node depth
};
__global__ void __launch_bounds__(256) k_local_levels(uint32_t *perm, const float *tri_bounds, const BlasInfo *blas, const LocalJob *jobs) {
    __shared__ unsigned long long key[LOCAL_MAX];
    __shared__ uint32_t val[LOCAL_MAX];
    __shared__ int bnd[LOCAL_MAX][6];
    const LocalJob job = jobs[blockIdx.x];
    const BlasInfo bi = blas[job.prim];
    const uint32_t n = job.count, tid = threadIdx.x;
    uint32_t *gperm = perm + bi.tri_base + job.first;
    for (uint32_t e = tid; e < LOCAL_MAX; e += 256)
        val[e] = e < n ? gperm[e] : 0xffffffffu;
    __syncthreads();
    for (uint32_t d = job.depth0; d < bi.depth; d++) {
        const uint32_t n_sub = 1u << (d - job.depth0);
        for (uint32_t i = tid; i < n_sub * 6; i += 256)
            bnd[i / 6][i % 6] = (i % 6) < 3 ? 0x7fffffff : (int)0x80000000;
        __syncthreads();
        // 1. bounds of every sub-node: union of its triangles' boxes
        for (uint32_t e = tid; e < n; e += 256) {
            uint32_t l = 0, r = bi.n_tris, j = 0;
            const uint32_t q = job.first + e;
            for (uint32_t b = 0; b < d; b++) {
                const uint32_t mid = l + (r - l) / 2;
                j <<= 1;
                if (q < mid) {
                    r = mid;
                } else {
                    l = mid;
                    j |= 1u;
                }
            }
            const uint32_t js = j & (n_sub - 1u);
            const float *b = tri_bounds + 6 * (size_t)val[e];
            for (int k = 0; k < 3; k++) {
                atomicMin(&bnd[js][k], f32_ordered(b[k]));
                atomicMax(&bnd[js][3 + k], f32_ordered(b[3 + k]));
            }
        }
        __syncthreads();
        // 2. composite keys
        for (uint32_t e = tid; e < LOCAL_MAX; e += 256) {
            unsigned long long kk = ~0ull; // padding sorts to the end
            if (e < n) {
                uint32_t l = 0, r = bi.n_tris, j = 0;
                const uint32_t q = job.first + e;
                for (uint32_t b = 0; b < d; b++) {
                    const uint32_t mid = l + (r - l) / 2;
                    j <<= 1;
                    if (q < mid) {
                        r = mid;
                    } else {
                        l = mid;
                        j |= 1u;
                    }
                }
                const uint32_t js = j & (n_sub - 1u);
                float lo[3], hi[3];
                for (int k = 0; k < 3; k++) {
                    lo[k] = f32_unordered(bnd[js][k]);
                    hi[k] = f32_unordered(bnd[js][3 + k]);
                }
                const float w0 = hi[0] - lo[0], w1 = hi[1] - lo[1], w2 = hi[2] - lo[2];
                int axis = 0; // strictly longest axis, x on ties (object_bvh.hpp:111-119)
                if (w1 > w0 && w1 > w2)
                    axis = 1;
                else if (w2 > w0 && w2 > w1)
                    axis = 2;
                const float *b = tri_bounds + 6 * (size_t)val[e];
                const float c = 0.5f * (b[axis] + b[3 + axis]);
                const float ext = hi[axis] - lo[axis];
                const float qq = ext > 0.0f ? (c - lo[axis]) / ext : 0.0f;
                const uint32_t q16 = (uint32_t)fminf(fmaxf(qq * 65535.0f, 0.0f), 65535.0f);
                kk = ((unsigned long long)js << 26) | ((unsigned long long)q16 << 10) | e;
            }
            key[e] = kk;
        }
        __syncthreads();
        // 3. bitonic sort of (key, val) over LOCAL_MAX slots
        for (uint32_t k2 = 2; k2 <= LOCAL_MAX; k2 <<= 1)
            for (uint32_t j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                for (uint32_t i = tid; i < LOCAL_MAX; i += 256) {
                    const uint32_t x = i ^ j2;
                    if (x > i) {
                        const unsigned long long a = key[i], b = key[x];
                        const bool up = (i & k2) == 0;
                        if ((a > b) == up) {
                            key[i] = b;
                            key[x] = a;
                            const uint32_t t = val[i];
                            val[i] = val[x];
                            val[x] = t;
                        }
                    }
                }
                __syncthreads();
            }
    }
    for (uint32_t e = tid; e < n; e += 256)
        gperm[e] = val[e];
}
__global__ void k_iota(uint32_t *p, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] = i;
}
// bounds[6*i .. 6*i+5] = (+max,+max,+max,-max,-max,-max) in order-preserving int space
__global__ void k_init_bounds(int *bounds, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6 * n)
        bounds[i] = (i % 6) < 3 ? 0x7fffffff : (int)0x80000000;
}
__global__ void k_seg_bounds(const uint32_t *perm, const float *tri_bounds, const uint32_t *tri_prim, const BlasInfo *blas, uint32_t n_tris,
                             uint32_t d, int *seg_bounds) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t seg = 0xffffffffu;
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    if (s < n_tris) {
        const BlasInfo bi = blas[tri_prim[s]];
        if (d < bi.depth) {
            uint32_t l, r;
            segment_at(s - bi.tri_base, d, bi.n_tris, l, r);
            seg = bi.tri_base + l;
            const float *b = tri_bounds + 6 * (size_t)perm[s];
            for (int k = 0; k < 3; k++) {
                lo[k] = b[k];
                hi[k] = b[3 + k];
            }
        }
    }
    const uint32_t seg0 = __shfl(seg, 0, 64);
    if (__all(seg == seg0)) { // whole wave inside one node: reduce first
        if (seg0 == 0xffffffffu)
            return;
        for (int off = 32; off > 0; off >>= 1)
            for (int k = 0; k < 3; k++) {
                lo[k] = fminf(lo[k], __shfl_xor(lo[k], off, 64));
                hi[k] = fmaxf(hi[k], __shfl_xor(hi[k], off, 64));
            }
        if ((threadIdx.x & 63) == 0)
            for (int k = 0; k < 3; k++) {
                atomicMin(seg_bounds + 6 * (size_t)seg0 + k, f32_ordered(lo[k]));
                atomicMax(seg_bounds + 6 * (size_t)seg0 + 3 + k, f32_ordered(hi[k]));
            }
    } else if (seg != 0xffffffffu) {
        for (int k = 0; k < 3; k++) {
            atomicMin(seg_bounds + 6 * (size_t)seg + k, f32_ordered(lo[k]));
            atomicMax(seg_bounds + 6 * (size_t)seg + 3 + k, f32_ordered(hi[k]));
        }
    }
}
// rank_base (optional): per primitive, the number of level-d nodes of the primitives before it.  The high part of the key is
// then the dense node rank instead of the node's start position — the same order with far fewer bits to sort (the few
// thousand nodes of the upper levels need 12 bits, a position needs 18).
__global__ void k_seg_keys(const uint32_t *perm, const float *tri_bounds, const uint32_t *tri_prim, const BlasInfo *blas, uint32_t n_tris, uint32_t d,
                           const int *seg_bounds, unsigned long long *keys, const uint32_t *rank_base) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_tris)
        return;
    const uint32_t prim = tri_prim[s];
    const BlasInfo bi = blas[prim];
    if (d >= bi.depth) { // this primitive's tree is complete: keep the element where it is
        keys[s] = rank_base ? (((unsigned long long)rank_base[prim] << 16) | (s - bi.tri_base)) : ((unsigned long long)s << 16);
        return;
    }
    uint32_t l = 0, r = bi.n_tris, j = 0;
    {
        const uint32_t q = s - bi.tri_base;
        for (uint32_t b = 0; b < d; b++) {
            const uint32_t mid = l + (r - l) / 2;
            j <<= 1;
            if (q < mid) {
                r = mid;
            } else {
                l = mid;
                j |= 1u;
            }
        }
    }
    const uint32_t seg = bi.tri_base + l;
    float lo[3], hi[3];
    for (int k = 0; k < 3; k++) {
        lo[k] = f32_unordered(seg_bounds[6 * (size_t)seg + k]);
        hi[k] = f32_unordered(seg_bounds[6 * (size_t)seg + 3 + k]);
    }
    const float w0 = hi[0] - lo[0], w1 = hi[1] - lo[1], w2 = hi[2] - lo[2];
    int axis = 0; // strictly longest axis, x on ties (object_bvh.hpp:111-119)
    if (w1 > w0 && w1 > w2)
        axis = 1;
    else if (w2 > w0 && w2 > w1)
        axis = 2;
    const float *b = tri_bounds + 6 * (size_t)perm[s];
    const float c = 0.5f * (b[axis] + b[3 + axis]);
    const float ext = hi[axis] - lo[axis];
    const float q = ext > 0.0f ? (c - lo[axis]) / ext : 0.0f;
    const uint32_t q16 = (uint32_t)fminf(fmaxf(q * 65535.0f, 0.0f), 65535.0f);
    keys[s] = ((unsigned long long)(rank_base ? rank_base[prim] + j : seg) << 16) | q16;
}

// ---- tree emission ---------------------------------------------------------------------------------
// sorted range [l, r) of leaf k in a perfect tree of depth D over n primitives: every node splits its range
// at l + (r-l)/2 (the reference's split_pos = size/2, object_bvh.hpp:121, on the Morton order)
__host__ __device__ inline void leaf_range(uint32_t k, uint32_t D, uint32_t n, uint32_t &l, uint32_t &r) {
    l = 0;
    r = n;
    for (int b = (int)D - 1; b >= 0; b--) {
        uint32_t mid = l + (r - l) / 2;
        if ((k >> b) & 1u)
            l = mid;
        else
            r = mid;
    }
}

template <typename T>
__global__ void k_emit_triangles(const uint32_t *sorted_vals, const TriRecF32 *mf32, const float *tri_bounds,
                                 const uint32_t *tri_prim, const lprt_mesh_vao *vaos, uint32_t n_tris, TriRec<T> *tris,
                                 TriRecF32 *tris_f32, float *sorted_bounds) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_tris)
        return;
    uint32_t g = sorted_vals[s];
    TriRecF32 f = mf32[g];
    TriRec<T> r;
#pragma unroll
    for (int k = 0; k < 3; k++) { // static_cast<Matrix<3,3,DataT>>(M2): one rounding per entry (memory.hpp:645)
        r.m0[k] = cvt<T>(f.m0[k]);
        r.m1[k] = cvt<T>(f.m1[k]);
        r.m2[k] = cvt<T>(f.m2[k]);
    }
    r.v2x = cvt<T>(f.v2x); // vbo position in DataT
    r.v2y = cvt<T>(f.v2y);
    r.v2z = cvt<T>(f.v2z);
    r.inobject_offset = 3 * g - vaos[tri_prim[g]].ebo_offset; // 3 x triangle index within the primitive
    tris[s] = r;
    if (tris_f32)
        tris_f32[s] = f;
#pragma unroll
    for (int k = 0; k < 6; k++)
        sorted_bounds[6 * s + k] = tri_bounds[6 * g + k];
}

// leaf_word: LPRT_CHILD_INNER, or (first << 3) | count of the triangles node i holds as a leaf (types.hpp: BlasNode)
template <typename T> __device__ __forceinline__ void store_child_box(BlasNode<T> *inner, T *roots, const BlasInfo &bi, uint32_t i,
                                                                      const float lo[3], const float hi[3], uint32_t leaf_word) {
    // node i (heap index, 1-based) of this BLAS: its box lives in its parent's record, or in the root slot
    if (i == 1) {
        T *rb = roots + 6 * bi.root_base;
        for (int k = 0; k < 3; k++) {
            rb[k] = round_down<T>(lo[k]);
            rb[3 + k] = round_up<T>(hi[k]);
        }
    } else {
        BlasNode<T> *p = inner + bi.inner_base + (i >> 1) - 1;
        T *dlo = (i & 1u) ? p->lo1 : p->lo0;
        T *dhi = (i & 1u) ? p->hi1 : p->hi0;
        for (int k = 0; k < 3; k++) {
            dlo[k] = round_down<T>(lo[k]);
            dhi[k] = round_up<T>(hi[k]);
        }
        p->leaf[i & 1u] = leaf_word;
    }
}

// grid.y = primitive; one thread per leaf
template <typename T>
__global__ void k_leaf_boxes(const BlasInfo *blas, const float *sorted_bounds, uint32_t *leaf_words, BlasNode<T> *inner,
                             T *roots) {
    const BlasInfo bi = blas[blockIdx.y];
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (bi.n_tris == 0 || k >= (1u << bi.depth))
        return;
    uint32_t l, r;
    leaf_range(k, bi.depth, bi.n_tris, l, r);
    leaf_words[bi.leaf_base + k] = (l << 1) | (r - l - 1);
    float lo[3], hi[3];
    const float *b = sorted_bounds + 6 * (size_t)(bi.tri_base + l);
    for (int a = 0; a < 3; a++) {
        lo[a] = b[a];
        hi[a] = b[3 + a];
    }
    if (r - l == 2)
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], b[6 + a]);
            hi[a] = fmaxf(hi[a], b[9 + a]);
        }
    store_child_box<T>(inner, roots, bi, (1u << bi.depth) + k, lo, hi, (l << 3) | (r - l));
}

// one launch per depth d (deepest first): node = union of the two child boxes already stored in it
template <typename T> __global__ void k_inner_level(const BlasInfo *blas, uint32_t d, BlasNode<T> *inner, T *roots) {
    const BlasInfo bi = blas[blockIdx.y];
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (bi.n_tris == 0 || d >= bi.depth || j >= (1u << d))
        return;
    uint32_t i = (1u << d) + j;
    const BlasNode<T> nd = inner[bi.inner_base + i - 1];
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) { // child boxes are already DataT values: their union is exact in DataT
        lo[a] = fminf((float)nd.lo0[a], (float)nd.lo1[a]);
        hi[a] = fmaxf((float)nd.hi0[a], (float)nd.hi1[a]);
    }
    store_child_box<T>(inner, roots, bi, i, lo, hi, LPRT_CHILD_INNER);
}

// ---- B1, surface-area builder ---------------------------------------------------------------------------------------------
// The reference splits every node at the object median of the first-vertex coordinates along the widest axis
// (object_bvh.hpp:76-134); the contract is what cast_ray reports, not the topology, and on the 262 k-triangle
// atrium the median tree costs a GI ray 83 box tests.  This builder chooses, per node, the axis AND the plane by the binned
// surface-area heuristic -- 16 bins of the triangles' centroids per axis, cost(plane) = area(left) n_left + area(right) n_right
// -- and stops as soon as a node holds <= SAH_LEAF triangles: 62 box tests per GI ray (-26 %), 4.8 instead of 7.1 triangle
// tests (CPU replica of the walk: tools/bvh_quality.cpp).  The tree keeps the IMPLICIT heap indexing of the median tree
// (children of slot i at 2i, 2i + 1; no child pointers, deferred siblings as one bit per level) by bounding the depth:
// D = ceil(log2(ceil(n / SAH_LEAF))) + 1, and a plane is admissible only if each side still fits the levels below it
// (count <= SAH_LEAF << levels).  The object median by position is always admissible, so the bound is never violated.
// Level by level, ONE WAVE PER NODE: pass 1 centroid bounds, pass 2 binning (LDS atomics, 336 words per wave), 45 lanes
// evaluate the 45 planes, pass 3 a stable partition (ballot ranks) that also yields the exact child boxes.  The node's record
// -- both child boxes and both leaf words -- is complete when its wave ends: no bottom-up refit.
constexpr uint32_t SAH_BINS = 16, SAH_LEAF = 2, SAH_BIG = 256, SAH_ONE_AXIS = 4096;
// An open node (more than SAH_LEAF triangles) waiting for its level: queue entry.  lo / hi: the domain its centroids are binned
// over -- the centroid bounds of the primitive for a root (k_tri_prepare), the exact box of its triangles for every other node
// (the parent's partition pass computes it anyway; a pass of its own over the centroids bought nothing measurable in tree
// quality: tools/bvh_quality.cpp).
struct SahItem {
    uint32_t prim, slot, start, count;
    float lo[3], hi[3];
    uint32_t pad[2];
};
// queues of one build: [kind 0: nodes of <= SAH_BIG triangles, one wave each | kind 1: larger, one workgroup each][level parity]
struct SahQueues {
    SahItem *items[2][2];
    uint32_t *count; // [2 kinds][levels + 1]
    uint32_t levels;
};
__device__ __forceinline__ void sah_push(const SahQueues &q, uint32_t level, const SahItem &it) {
    const uint32_t kind = it.count > SAH_BIG ? 1u : 0u;
    const uint32_t pos = atomicAdd(q.count + kind * (q.levels + 1u) + level, 1u);
    q.items[kind][level & 1u][pos] = it;
}
__global__ void k_sah_init(const BlasInfo *blas, uint32_t n_vaos, const int *prim_cbounds, SahQueues q) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_vaos)
        return;
    const BlasInfo bi = blas[p];
    if (bi.n_tris <= 3u)
        return;
    SahItem it{};
    it.prim = p;
    it.slot = 1u;
    it.start = 0u;
    it.count = bi.n_tris;
    for (int k = 0; k < 3; k++) {
        it.lo[k] = f32_unordered(prim_cbounds[6 * p + k]);
        it.hi[k] = f32_unordered(prim_cbounds[6 * p + 3 + k]);
    }
    sah_push(q, 0u, it);
}
__device__ __forceinline__ float wave_min(float v) {
    for (int off = 32; off > 0; off >>= 1)
        v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    for (int off = 32; off > 0; off >>= 1)
        v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float box_area(const float lo[3], const float hi[3]) {
    const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return dx * dy + dy * dz + dz * dx;
}
// One level: the open nodes of queue (kind, d), one group of GW waves each.  src / dst: the two halves of the ping-pong order
// (level d reads the one, writes the other); fin receives every level's arrangement as well -- a position's LAST write is its
// leaf's, so after the last level `fin` is the leaf order.
// GW = 1 (kind 0, sixteen nodes per 1024-thread workgroup, a wave each: the many nodes of <= SAH_BIG triangles) or 16 (kind 1,
// one 1024-thread workgroup per node: up to 47 k triangles on the atrium).  Nodes above SAH_ONE_AXIS triangles bin the widest
// axis of their domain only -- 7 LDS atomics per triangle instead of 21, on a few addresses all 16 waves fight over -- which
// is where the plane of such a node lies anyway (tools/bvh_quality.cpp: +1 % box tests with EVERY node restricted so).
// One triangle per lane into the bins of an axis (bin[k] = count, lo[3], hi[3] in order-preserving int space).  The triangles
// of a node arrive in the order the level above left them in, so most lanes of a wave fall into the same one or two bins, and
// 64 lanes adding to ONE LDS word are 64 atomics the LDS serves one after the other: level 0 of the atrium (24 workgroups of
// 11 k triangles) spent 350 us on 77 k such atomics per workgroup.  Here the wave settles each bin it touches among itself
// first -- the lanes of one bin are counted by a ballot and their boxes reduced by butterflies -- and ONE lane adds the result.
#ifndef SAH_AGG_MIN
#define SAH_AGG_MIN 16
#endif
__device__ __forceinline__ void sah_bin_add(int (*bins)[7], bool valid, uint32_t k, const float (&lo)[3], const float (&hi)[3]) {
    const uint32_t lane = threadIdx.x & 63u;
    unsigned long long todo = __builtin_amdgcn_ballot_w64(valid);
    // bins that hold a quarter of the wave or more are settled inside the wave (at most four rounds); whoever is left -- lanes
    // spread over many bins, the rule below level 0, where a node's triangles are no longer in an order that follows the axis
    // -- adds for itself: sixteen rounds of butterflies cost more than the atomics they replace (level 1: 111 -> 274 us)
    while (todo) {
        const uint32_t leader = (uint32_t)__builtin_ctzll(todo);
        const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)k, (int)leader);
        const bool mine = valid && k == k0;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(mine) & todo;
        if (__builtin_popcountll(m) < SAH_AGG_MIN)
            break;
        float mn[3], mx[3];
#pragma unroll
        for (int x = 0; x < 3; x++) {
            mn[x] = mine ? lo[x] : 3.4e38f;
            mx[x] = mine ? hi[x] : -3.4e38f;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1)
#pragma unroll
            for (int x = 0; x < 3; x++) {
                mn[x] = fminf(mn[x], __shfl_xor(mn[x], off, 64));
                mx[x] = fmaxf(mx[x], __shfl_xor(mx[x], off, 64));
            }
        if (lane == leader) {
            int *bn = bins[k0];
            atomicAdd(bn, (int)__builtin_popcountll(m));
#pragma unroll
            for (int x = 0; x < 3; x++) {
                atomicMin(bn + 1 + x, f32_ordered(mn[x]));
                atomicMax(bn + 4 + x, f32_ordered(mx[x]));
            }
        }
        todo &= ~m;
    }
    if (valid && ((todo >> lane) & 1ull)) {
        int *bn = bins[k];
        atomicAdd(bn, 1);
#pragma unroll
        for (int x = 0; x < 3; x++) {
            atomicMin(bn + 1 + x, f32_ordered(lo[x]));
            atomicMax(bn + 4 + x, f32_ordered(hi[x]));
        }
    }
}

template <typename T, uint32_t GW>
__global__ void __launch_bounds__(1024) k_sah_level(const BlasInfo *blas, uint32_t d, SahQueues q, const uint32_t *src, uint32_t *dst,
                                                                         uint32_t *fin, const float *tri_bounds, BlasNode<T> *inner, T *roots) {
    constexpr uint32_t NPB = GW == 1u ? 16u : 1u, GT = 64u * GW, KIND = GW == 1u ? 0u : 1u; // nodes per workgroup, threads per node
    __shared__ int s_bins[NPB][3][SAH_BINS][7]; // per node: count, lo[3], hi[3] (bounds in order-preserving int space)
    // a big node's sixteen waves bin into copies of their own, merged afterwards: sixteen waves adding to the same sixteen
    // LDS words queue up behind each other (the copies take level 0 of the atrium from 183 to 171 us)
    __shared__ int s_wbins[GW > 1u ? GW : 1u][GW > 1u ? 3u : 1u][GW > 1u ? SAH_BINS : 1u][7];
    __shared__ int s_red[NPB][12];                 // group reductions (GW > 1)
    __shared__ uint32_t s_part[2][GW + 1u];        // per-wave left / right counts of a partition chunk (GW > 1)
    __shared__ uint32_t s_best[4];
    __shared__ uint32_t s_push[2]; // GW == 1: open children of the workgroup's 16 nodes, reserved with ONE global atomic
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint32_t grp = GW == 1u ? wv : 0u, t = GW == 1u ? lane : threadIdx.x;
    const uint32_t w = blockIdx.x * NPB + grp;
    if (GW == 1u) {
        if (threadIdx.x == 0u)
            s_push[0] = 0u;
        __syncthreads();
    }
    // children that stay open, filled by the group's first thread and queued for the next level at the end
    SahItem open_child[2];
    uint32_t n_open = 0;
    const bool active = w < q.count[KIND * (q.levels + 1u) + d];
    if (GW > 1u && !active)
        return;
    if (active) {
    const SahItem it = q.items[KIND][d & 1u][w];
    const uint32_t p = it.prim, slot = it.slot, l = it.start, n = it.count;
    const BlasInfo bi = blas[p];
    auto sync = [&]() {
        if (GW == 1u)
            __builtin_amdgcn_wave_barrier(); // (the LDS operations of a wave execute in order)
        else
            __syncthreads();
    };
    const uint32_t *perm = src + bi.tri_base + l;
    int(*bins)[SAH_BINS][7] = s_bins[grp];
    // nodes above SAH_ONE_AXIS triangles bin the widest axis of their domain only (see above)
    const uint32_t n_axes = (GW > 1u && n > SAH_ONE_AXIS) ? 1u : 3u;
    for (uint32_t qi = t; qi < 3u * SAH_BINS * 7u; qi += GT) {
        const uint32_t f = qi % 7u;
        (&bins[0][0][0])[qi] = f == 0u ? 0 : (f < 4u ? 0x7fffffff : (int)0x80000000);
    }
    if (GW > 1u)
        for (uint32_t qi = t; qi < GW * 3u * SAH_BINS * 7u; qi += GT) {
            const uint32_t f = qi % 7u;
            (&s_wbins[0][0][0][0])[qi] = f == 0u ? 0 : (f < 4u ? 0x7fffffff : (int)0x80000000);
        }
    float cmin[3], ext[3], scale[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        cmin[k] = it.lo[k];
        ext[k] = it.hi[k] - it.lo[k];
        scale[k] = ext[k] > 0.f ? (float)SAH_BINS / ext[k] : 0.f;
    }
    // the axis a big node bins: the widest of its domain (x on ties)
    const uint32_t wide_axis = (ext[1] > ext[0] && ext[1] > ext[2]) ? 1u : (ext[2] > ext[0] && ext[2] > ext[1]) ? 2u : 0u;
    sync(); // the bins are initialised
    // A node of at most UNR elements per thread keeps them in registers from here to the end of the partition (every node
    // below level 1 of the atrium; a wave-per-node group always: SAH_BIG = 4 x 64): its index loads, then its box gathers, are
    // in flight together ONCE.  Read where they are used, they were two dependent round trips to memory per 64 (or 1,024)
    // triangles and pass -- sixteen in a row for a 256-triangle node -- with nothing else for the wave to do meanwhile.
    constexpr uint32_t UNR = 4u;
    // Big nodes only.  The wave-per-node groups of the lower levels come by the thousand: their round trips hide behind each
    // other, and the registers this costs them (two 1,024-thread workgroups per CU need <= 64) cost more -- 60 -> 90 us a level.
    const bool resident = GW > 1u && n <= UNR * GT;
    constexpr bool KEEP_BOXES = true;
    uint32_t rg[UNR];
    float rb[UNR][6];
    auto gather_boxes = [&]() {
#pragma unroll
        for (uint32_t u = 0; u < UNR; u++) {
            const float *bp = tri_bounds + 6 * (size_t)(rg[u] != 0xffffffffu ? rg[u] : 0u);
#pragma unroll
            for (int x = 0; x < 6; x++)
                rb[u][x] = bp[x];
        }
    };
    if (resident) {
#pragma unroll
        for (uint32_t u = 0; u < UNR; u++)
            rg[u] = t + u * GT < n ? perm[t + u * GT] : 0xffffffffu;
        gather_boxes();
    }
    // one element into the bins: a wave-per-node group with plain atomics (its 64 lanes spread over the bins), a big node's
    // waves into their own copies, settling crowded bins inside the wave first (sah_bin_add)
    auto bin_one = [&](bool valid, const float (&bx)[6]) {
        const float lo[3] = {bx[0], bx[1], bx[2]}, hi[3] = {bx[3], bx[4], bx[5]};
        for (uint32_t ai = 0; ai < n_axes; ai++) {
            const uint32_t a = n_axes == 3u ? ai : wide_axis;
            const float c = 0.5f * (lo[a] + hi[a]);
            const uint32_t k = valid ? min((uint32_t)fmaxf((c - cmin[a]) * scale[a], 0.f), SAH_BINS - 1u) : 0u;
            if (GW > 1u) {
                sah_bin_add(s_wbins[wv][ai], valid, k, lo, hi);
            } else if (valid) {
                int *bn = bins[ai][k];
                atomicAdd(bn, 1);
#pragma unroll
                for (int x = 0; x < 3; x++) {
                    atomicMin(bn + 1 + x, f32_ordered(lo[x]));
                    atomicMax(bn + 4 + x, f32_ordered(hi[x]));
                }
            }
        }
    };
    // pass 1: binning
    if (resident) {
#pragma unroll
        for (uint32_t u = 0; u < UNR; u++)
            if (u * GT < n) // (uniform over the group: every lane of a wave goes through sah_bin_add)
                bin_one(rg[u] != 0xffffffffu, rb[u]);
    } else if (GW == 1u) {
        for (uint32_t e = t; e < n; e += GT) {
            const float *bp = tri_bounds + 6 * (size_t)perm[e];
            const float bx[6] = {bp[0], bp[1], bp[2], bp[3], bp[4], bp[5]};
            bin_one(true, bx);
        }
    } else {
        // (a node this big is ONE workgroup walking 10^4 triangles: four elements per thread and round, their loads together)
        for (uint32_t base = 0; base < n; base += UNR * GT) {
            const uint32_t e0 = base + t;
            uint32_t gi[UNR];
            float bb[UNR][6];
#pragma unroll
            for (uint32_t u = 0; u < UNR; u++)
                gi[u] = e0 + u * GT < n ? perm[e0 + u * GT] : 0xffffffffu;
#pragma unroll
            for (uint32_t u = 0; u < UNR; u++) {
                const float *bp = tri_bounds + 6 * (size_t)(gi[u] != 0xffffffffu ? gi[u] : 0u);
#pragma unroll
                for (int x = 0; x < 6; x++)
                    bb[u][x] = bp[x];
            }
#pragma unroll
            for (uint32_t u = 0; u < UNR; u++)
                bin_one(gi[u] != 0xffffffffu, bb[u]);
        }
    }
    if (GW > 1u) { // the waves' copies into the node's bins
        __syncthreads();
        for (uint32_t qi = t; qi < 3u * SAH_BINS * 7u; qi += GT) {
            const uint32_t f = qi % 7u;
            int v = (&s_wbins[0][0][0][0])[qi];
            for (uint32_t wq = 1; wq < GW; wq++) {
                const int o = (&s_wbins[wq][0][0][0])[qi];
                v = f == 0u ? v + o : (f < 4u ? min(v, o) : max(v, o));
            }
            (&bins[0][0][0])[qi] = v;
        }
    }
    sync();
    // the planes, one per lane of the first wave: everything in bins <= plane goes left
    const uint32_t cap = SAH_LEAF << (bi.depth - d - 1u); // what a child's subtree can hold
    uint32_t by_plane_u = 0, axis = 0, plane = 0, nl = n / 2u;
    if (GW == 1u || wv == 0u) {
        float cost = __builtin_inff();
        uint32_t nl_lane = 0;
        if (lane < n_axes * (SAH_BINS - 1u)) {
            const uint32_t ai = lane / (SAH_BINS - 1u), pl = lane % (SAH_BINS - 1u);
            const uint32_t a = n_axes == 3u ? ai : wide_axis;
            float llo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, lhi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
            float rlo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, rhi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
            uint32_t cl = 0, cr = 0;
            for (uint32_t k = 0; k < SAH_BINS; k++) {
                const int *bn = bins[ai][k];
                const uint32_t c = (uint32_t)bn[0];
                if (!c)
                    continue;
                const bool left = k <= pl;
                (left ? cl : cr) += c;
#pragma unroll
                for (int x = 0; x < 3; x++) {
                    const float bl = f32_unordered(bn[1 + x]), bh = f32_unordered(bn[4 + x]);
                    if (left) {
                        llo[x] = fminf(llo[x], bl);
                        lhi[x] = fmaxf(lhi[x], bh);
                    } else {
                        rlo[x] = fminf(rlo[x], bl);
                        rhi[x] = fmaxf(rhi[x], bh);
                    }
                }
            }
            if (ext[a] > 0.f && cl >= 1u && cr >= 1u && cl <= cap && cr <= cap)
                cost = box_area(llo, lhi) * (float)cl + box_area(rlo, rhi) * (float)cr;
            nl_lane = cl;
        }
        // the cheapest plane (the lowest lane among equals)
        float best = cost;
        uint32_t best_lane = lane;
        for (int off = 32; off > 0; off >>= 1) {
            const float oc = __shfl_xor(best, off, 64);
            const uint32_t ol = (uint32_t)__shfl_xor((int)best_lane, off, 64);
            if (oc < best || (oc == best && ol < best_lane)) {
                best = oc;
                best_lane = ol;
            }
        }
        // no admissible plane (coincident centroids, or every plane overfills a side): the object median by position
        by_plane_u = best < __builtin_inff() ? 1u : 0u;
        axis = by_plane_u ? (n_axes == 3u ? best_lane / (SAH_BINS - 1u) : wide_axis) : 0u;
        plane = by_plane_u ? best_lane % (SAH_BINS - 1u) : 0u;
        nl = by_plane_u ? (uint32_t)__shfl((int)nl_lane, (int)best_lane, 64) : n / 2u;
        if (GW > 1u && lane == 0u) {
            s_best[0] = by_plane_u;
            s_best[1] = axis;
            s_best[2] = plane;
            s_best[3] = nl;
        }
    }
    if (GW > 1u) {
        __syncthreads();
        by_plane_u = s_best[0];
        axis = s_best[1];
        plane = s_best[2];
        nl = s_best[3];
    }
    const bool by_plane = by_plane_u != 0u;
    // pass 2: stable partition + the exact boxes of both sides
    float blo[2][3] = {{3.4e38f, 3.4e38f, 3.4e38f}, {3.4e38f, 3.4e38f, 3.4e38f}}, bhi[2][3] = {{-3.4e38f, -3.4e38f, -3.4e38f}, {-3.4e38f, -3.4e38f, -3.4e38f}};
    uint32_t out_l = 0, out_r = 0;
    uint32_t *o1 = dst + bi.tri_base + l, *o2 = fin + bi.tri_base + l;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // one chunk of GT elements (element base + t of the node: triangle g with box b): ranks run on from chunk to chunk
    auto rank_chunk = [&](uint32_t base, bool valid, uint32_t g, const float (&b)[6]) {
        const uint32_t e = base + t;
        bool left = false;
        if (valid) {
            if (by_plane) {
                const float c = 0.5f * (b[axis] + b[3 + axis]);
                left = min((uint32_t)fmaxf((c - cmin[axis]) * scale[axis], 0.f), SAH_BINS - 1u) <= plane;
            } else {
                left = e < nl;
            }
            const int sd = left ? 0 : 1;
#pragma unroll
            for (int x = 0; x < 3; x++) {
                blo[sd][x] = fminf(blo[sd][x], b[x]);
                bhi[sd][x] = fmaxf(bhi[sd][x], b[3 + x]);
            }
        }
        const unsigned long long ml = __builtin_amdgcn_ballot_w64(valid && left), mr = __builtin_amdgcn_ballot_w64(valid && !left);
        uint32_t before_l = 0, before_r = 0, tot_l = (uint32_t)__builtin_popcountll(ml), tot_r = (uint32_t)__builtin_popcountll(mr);
        if (GW > 1u) { // the waves of the group in order: ranks stay those of a sequential pass
            if (lane == 0u) {
                s_part[0][wv] = tot_l;
                s_part[1][wv] = tot_r;
            }
            __syncthreads();
            tot_l = tot_r = 0;
            for (uint32_t qi = 0; qi < GW; qi++) {
                const uint32_t a = s_part[0][qi], c = s_part[1][qi];
                before_l += qi < wv ? a : 0u;
                before_r += qi < wv ? c : 0u;
                tot_l += a;
                tot_r += c;
            }
        }
        if (valid) {
            const uint32_t pos = left ? out_l + before_l + (uint32_t)__builtin_popcountll(ml & lt_mask)
                                      : nl + out_r + before_r + (uint32_t)__builtin_popcountll(mr & lt_mask);
            o1[pos] = g;
            o2[pos] = g;
        }
        out_l += tot_l;
        out_r += tot_r;
        if (GW > 1u)
            __syncthreads(); // s_part is rewritten by the next chunk
    };
    if (resident) {
        if (!KEEP_BOXES)
            gather_boxes();
#pragma unroll
        for (uint32_t u = 0; u < UNR; u++)
            if (u * GT < n)
                rank_chunk(u * GT, rg[u] != 0xffffffffu, rg[u], rb[u]);
    } else if (GW == 1u) {
        for (uint32_t base = 0; base < n; base += GT) {
            const uint32_t e = base + t;
            uint32_t g = 0;
            float bx[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (e < n) {
                g = perm[e];
                const float *bp = tri_bounds + 6 * (size_t)g;
#pragma unroll
                for (int x = 0; x < 6; x++)
                    bx[x] = bp[x];
            }
            rank_chunk(base, e < n, g, bx);
        }
    } else {
        // the chunks are sequential, their loads need not be: while chunk c is ranked, the box of chunk c + 1's element and the
        // index of chunk c + 2's are on their way
        uint32_t g_cur = t < n ? perm[t] : 0u, g_nx1 = t + GT < n ? perm[t + GT] : 0u;
        float b_cur[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (t < n) {
            const float *bp = tri_bounds + 6 * (size_t)g_cur;
#pragma unroll
            for (int x = 0; x < 6; x++)
                b_cur[x] = bp[x];
        }
        for (uint32_t base = 0; base < n; base += GT) {
            const uint32_t e = base + t;
            float b_nx1[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            uint32_t g_nx2 = 0u;
            if (e + GT < n) {
                const float *bp = tri_bounds + 6 * (size_t)g_nx1;
#pragma unroll
                for (int x = 0; x < 6; x++)
                    b_nx1[x] = bp[x];
            }
            if (e + 2u * GT < n)
                g_nx2 = perm[e + 2u * GT];
            rank_chunk(base, e < n, g_cur, b_cur);
            g_cur = g_nx1;
            g_nx1 = g_nx2;
#pragma unroll
            for (int x = 0; x < 6; x++)
                b_cur[x] = b_nx1[x];
        }
    }
#pragma unroll
    for (int sd = 0; sd < 2; sd++)
#pragma unroll
        for (int x = 0; x < 3; x++) {
            blo[sd][x] = wave_min(blo[sd][x]);
            bhi[sd][x] = wave_max(bhi[sd][x]);
        }
    if (GW > 1u) {
        if (t < 12u)
            s_red[grp][t] = (t % 6u) < 3u ? 0x7fffffff : (int)0x80000000;
        __syncthreads();
        if (lane == 0u)
#pragma unroll
            for (int sd = 0; sd < 2; sd++)
#pragma unroll
                for (int x = 0; x < 3; x++) {
                    atomicMin(&s_red[grp][6 * sd + x], f32_ordered(blo[sd][x]));
                    atomicMax(&s_red[grp][6 * sd + 3 + x], f32_ordered(bhi[sd][x]));
                }
        __syncthreads();
#pragma unroll
        for (int sd = 0; sd < 2; sd++)
#pragma unroll
            for (int x = 0; x < 3; x++) {
                blo[sd][x] = f32_unordered(s_red[grp][6 * sd + x]);
                bhi[sd][x] = f32_unordered(s_red[grp][6 * sd + 3 + x]);
            }
    }
    if (t == 0u) {
        const uint32_t nr = n - nl;
        BlasNode<T> nd;
#pragma unroll
        for (int x = 0; x < 3; x++) {
            nd.lo0[x] = round_down<T>(blo[0][x]);
            nd.hi0[x] = round_up<T>(bhi[0][x]);
            nd.lo1[x] = round_down<T>(blo[1][x]);
            nd.hi1[x] = round_up<T>(bhi[1][x]);
        }
        nd.leaf[0] = nl <= SAH_LEAF ? ((l << 3) | nl) : LPRT_CHILD_INNER;
        nd.leaf[1] = nr <= SAH_LEAF ? (((l + nl) << 3) | nr) : LPRT_CHILD_INNER;
        inner[bi.inner_base + slot - 1u] = nd;
        // A child that is a leaf gets a record of its own as well (entry 0: its box and leaf word; entry 1: a far-away point
        // with no triangles), so that the two-level walk can fetch the records of BOTH children of a node without knowing what
        // they are (trace.hpp).  Slots exist for depths < D.  An open child goes to the next level's queue.
        const T far_ = round_down<T>(60000.0f);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            if (nd.leaf[k] != LPRT_CHILD_INNER) {
                if (d + 1u < bi.depth) {
                    BlasNode<T> lr;
#pragma unroll
                    for (int x = 0; x < 3; x++) {
                        lr.lo0[x] = k ? nd.lo1[x] : nd.lo0[x];
                        lr.hi0[x] = k ? nd.hi1[x] : nd.hi0[x];
                        lr.lo1[x] = lr.hi1[x] = far_;
                    }
                    lr.leaf[0] = nd.leaf[k];
                    lr.leaf[1] = 0u;
                    inner[bi.inner_base + 2u * slot + (uint32_t)k - 1u] = lr;
                }
            } else {
                SahItem c{};
                c.prim = p;
                c.slot = 2u * slot + (uint32_t)k;
                c.start = k ? l + nl : l;
                c.count = k ? nr : nl;
#pragma unroll
                for (int x = 0; x < 3; x++) {
                    c.lo[x] = blo[k][x];
                    c.hi[x] = bhi[k][x];
                }
                if (GW == 1u)
                    open_child[n_open++] = c;
                else
                    sah_push(q, d + 1u, c); // a handful of big nodes per level: one atomic each
            }
        }
        if (d == 0u) {
            T *rb = roots + 6 * bi.root_base;
#pragma unroll
            for (int x = 0; x < 3; x++) {
                rb[x] = round_down<T>(fminf(blo[0][x], blo[1][x]));
                rb[3 + x] = round_up<T>(fmaxf(bhi[0][x], bhi[1][x]));
            }
        }
    }
    } // active
    if (GW == 1u) {
        // Tens of thousands of small nodes per level: one device-scope atomic per node on the level's counter serialised into
        // 0.1-0.4 ms per level (rocprofv3 trace of the first form).  The 16 nodes of a workgroup reserve their children's queue
        // slots together: an LDS atomic each, one global atomic for the workgroup.  (Children of small nodes are small.)
        uint32_t local = 0;
        if (lane == 0u && n_open)
            local = atomicAdd(&s_push[0], n_open);
        __syncthreads();
        if (threadIdx.x == 0u && s_push[0])
            s_push[1] = atomicAdd(q.count + d + 1u, s_push[0]);
        __syncthreads();
        if (lane == 0u)
            for (uint32_t k = 0; k < n_open; k++)
                q.items[0][(d + 1u) & 1u][s_push[1] + local + k] = open_child[k];
    }
}
// primitives of <= 3 triangles have no records (the walk tests their triangles directly); their root boxes
template <typename T> __global__ void k_tiny_roots(const BlasInfo *blas, uint32_t n_vaos, const float *tri_bounds, T *roots) {
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n_vaos)
        return;
    const BlasInfo bi = blas[p];
    if (bi.n_tris == 0u || bi.n_tris > 3u)
        return;
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (uint32_t t = 0; t < bi.n_tris; t++)
        for (int x = 0; x < 3; x++) {
            lo[x] = fminf(lo[x], tri_bounds[6 * (size_t)(bi.tri_base + t) + x]);
            hi[x] = fmaxf(hi[x], tri_bounds[6 * (size_t)(bi.tri_base + t) + 3 + x]);
        }
    T *rb = roots + 6 * bi.root_base;
    for (int x = 0; x < 3; x++) {
        rb[x] = round_down<T>(lo[x]);
        rb[3 + x] = round_up<T>(hi[x]);
    }
}

// ---- B2: TLAS in one workgroup ---------------------------------------------------------------------
constexpr int TLAS_MAX = 1024;
template <typename T>
__global__ void __launch_bounds__(256) k_build_tlas(const lprt_aabb *aabbs, uint32_t n, uint32_t D, ObjSlot<T> *slots,
                                                    uint32_t *leaf_words, InnerNode<T> *inner, T *root, SceneSetup<T> su) {
    // The rest of lprt_set_scene rides along (one dispatch per frame instead of four): DataT copies of the lights and the
    // instance constants first, the per-slot object records after the slots exist.
    for (uint32_t i = threadIdx.x; i < su.n_lights; i += 256) {
        LightT<T> l;
        convert_light(su.lights32[i], l);
        su.lights[i] = l;
    }
    for (uint32_t i = threadIdx.x; i < su.n_objects; i += 256) {
        ObjConst<T> o;
        convert_object(su.objects32[i], o);
        su.objects[i] = o;
    }
    if (n == 0)
        return;
    __shared__ unsigned long long keys[TLAS_MAX];
    __shared__ float red[6][256];
    __shared__ float lvl_lo[TLAS_MAX][3], lvl_hi[TLAS_MAX][3]; // boxes of the level being reduced
    const uint32_t tid = threadIdx.x;
    // 1. bounds of the box centres
    float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (uint32_t i = tid; i < n; i += 256)
        for (int a = 0; a < 3; a++) {
            float c = 0.5f * (aabbs[i].lower[a] + aabbs[i].upper[a]);
            lo[a] = fminf(lo[a], c);
            hi[a] = fmaxf(hi[a], c);
        }
    for (int a = 0; a < 3; a++) {
        red[a][tid] = lo[a];
        red[3 + a][tid] = hi[a];
    }
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < (unsigned)s)
            for (int a = 0; a < 3; a++) {
                red[a][tid] = fminf(red[a][tid], red[a][tid + s]);
                red[3 + a][tid] = fmaxf(red[3 + a][tid], red[3 + a][tid + s]);
            }
        __syncthreads();
    }
    // 2. keys = morton << 32 | index, padded with ~0 to a power of two; bitonic sort
    uint32_t n2 = 1;
    while (n2 < n)
        n2 <<= 1;
    for (uint32_t i = tid; i < n2; i += 256) {
        unsigned long long k = ~0ull;
        if (i < n) {
            float q[3];
            for (int a = 0; a < 3; a++) {
                float ext = red[3 + a][0] - red[a][0];
                float c = 0.5f * (aabbs[i].lower[a] + aabbs[i].upper[a]);
                q[a] = ext > 0.0f ? (c - red[a][0]) / ext : 0.0f;
            }
            k = ((unsigned long long)morton30(q[0], q[1], q[2]) << 32) | i;
        }
        keys[i] = k;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= n2; size <<= 1)
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t i = tid; i < n2; i += 256) {
                uint32_t j = i ^ stride;
                if (j > i) {
                    bool up = (i & size) == 0;
                    unsigned long long a = keys[i], b = keys[j];
                    if ((a > b) == up) {
                        keys[i] = b;
                        keys[j] = a;
                    }
                }
            }
            __syncthreads();
        }
    // 3. slots in sorted order (aabbs_W + inscene_objid, scene_bvh.hpp:60-69)
    for (uint32_t s = tid; s < n; s += 256) {
        uint32_t id = (uint32_t)keys[s];
        ObjSlot<T> o;
        for (int a = 0; a < 3; a++) {
            o.lo[a] = round_down<T>(aabbs[id].lower[a]);
            o.hi[a] = round_up<T>(aabbs[id].upper[a]);
        }
        o.inscene_objid = id;
        slots[s] = o;
    }
    // 4. leaves, then levels bottom-up through LDS
    const uint32_t n_leaves = 1u << D;
    for (uint32_t k = tid; k < n_leaves; k += 256) {
        uint32_t l, r;
        leaf_range(k, D, n, l, r);
        leaf_words[k] = (l << 1) | (r - l - 1);
        uint32_t id = (uint32_t)keys[l];
        for (int a = 0; a < 3; a++) {
            lvl_lo[k][a] = aabbs[id].lower[a];
            lvl_hi[k][a] = aabbs[id].upper[a];
        }
        if (r - l == 2) {
            uint32_t id2 = (uint32_t)keys[l + 1];
            for (int a = 0; a < 3; a++) {
                lvl_lo[k][a] = fminf(lvl_lo[k][a], aabbs[id2].lower[a]);
                lvl_hi[k][a] = fmaxf(lvl_hi[k][a], aabbs[id2].upper[a]);
            }
        }
    }
    __syncthreads();
    for (int d = (int)D; d >= 0; d--) { // boxes of depth d are in lvl_*[0 .. 2^d)
        uint32_t cnt = 1u << d;
        float nlo[2][3], nhi[2][3]; // a thread handles at most TLAS_MAX/256/2 = 2 parents per level
        uint32_t np = 0;
        for (uint32_t j = tid; j < cnt; j += 256) {
            uint32_t i = cnt + j; // heap index
            if (i == 1) {
                for (int a = 0; a < 3; a++) {
                    root[a] = round_down<T>(lvl_lo[0][a]);
                    root[3 + a] = round_up<T>(lvl_hi[0][a]);
                }
            } else {
                InnerNode<T> *p = inner + (i >> 1) - 1;
                T *dlo = (i & 1u) ? p->lo1 : p->lo0;
                T *dhi = (i & 1u) ? p->hi1 : p->hi0;
                for (int a = 0; a < 3; a++) {
                    dlo[a] = round_down<T>(lvl_lo[j][a]);
                    dhi[a] = round_up<T>(lvl_hi[j][a]);
                }
            }
        }
        if (d > 0)
            for (uint32_t j = tid; j < (cnt >> 1); j += 256, np++)
                for (int a = 0; a < 3; a++) {
                    nlo[np][a] = fminf(lvl_lo[2 * j][a], lvl_lo[2 * j + 1][a]);
                    nhi[np][a] = fmaxf(lvl_hi[2 * j][a], lvl_hi[2 * j + 1][a]);
                }
        __syncthreads();
        if (d > 0) {
            np = 0;
            for (uint32_t j = tid; j < (cnt >> 1); j += 256, np++)
                for (int a = 0; a < 3; a++) {
                    lvl_lo[j][a] = nlo[np][a];
                    lvl_hi[j][a] = nhi[np][a];
                }
        }
        __syncthreads();
    }
    if (su.blas) { // slots and instance constants were written by this workgroup: make them visible, then gather
        __threadfence_block();
        __syncthreads();
        for (uint32_t s = threadIdx.x; s < n; s += 256) {
            ObjRec<T> r;
            make_objrec(slots[s], su.objects[slots[s].inscene_objid], su.blas, su.roots, su.n_vaos, r);
            su.objrecs[s] = r;
        }
    }
}

// ---- host drivers -------------------------------------------------------------------------------------
static inline uint32_t floor_log2(uint32_t n) {
    uint32_t d = 0;
    while ((2u << d) <= n)
        d++;
    return d;
}

template <typename T> void GeometryStore<T>::upload(hipStream_t st, const lprt_vertex *vbo, uint32_t nv, const uint32_t *ebo, uint32_t ni,
                                                    const lprt_mesh_vao *vaos, uint32_t nvaos) {
    n_vertices = nv;
    n_indices = ni;
    n_tris = ni / 3;
    n_vaos = nvaos;
    host_vaos.assign(vaos, vaos + nvaos);
    d_vbo32.reserve(sizeof(lprt_vertex) * (size_t)std::max(nv, 1u));
    d_vbo.reserve(sizeof(Vertex<T>) * (size_t)std::max(nv, 1u));
    d_ebo.reserve(sizeof(uint32_t) * (size_t)std::max(ni, 1u));
    d_vaos.reserve(sizeof(lprt_mesh_vao) * (size_t)std::max(nvaos, 1u));
    if (nv)
        LPRT_HIP(hipMemcpyAsync(d_vbo32.ptr, vbo, sizeof(lprt_vertex) * (size_t)nv, hipMemcpyHostToDevice, st));
    if (ni)
        LPRT_HIP(hipMemcpyAsync(d_ebo.ptr, ebo, sizeof(uint32_t) * (size_t)ni, hipMemcpyHostToDevice, st));
    if (nvaos)
        LPRT_HIP(hipMemcpyAsync(d_vaos.ptr, vaos, sizeof(lprt_mesh_vao) * (size_t)nvaos, hipMemcpyHostToDevice, st));
    // triangle -> primitive map (host arithmetic on sizes only); the per-primitive tree geometry follows the builder (layout)
    std::vector<uint32_t> tri_prim(std::max(n_tris, 1u), 0xffffffffu);
    host_blas.assign(nvaos, BlasInfo{});
    for (uint32_t p = 0; p < nvaos; p++) {
        BlasInfo &bi = host_blas[p];
        bi.n_tris = vaos[p].ebo_length / 3;
        bi.tri_base = vaos[p].ebo_offset / 3;
        bi.root_base = p;
        for (uint32_t t = 0; t < bi.n_tris; t++)
            tri_prim[bi.tri_base + t] = p;
    }
    layout_kind = -1;
    set_layout(st, builder_hint);
    d_tri_prim.reserve(sizeof(uint32_t) * tri_prim.size());
    LPRT_HIP(hipMemcpyAsync(d_tri_prim.ptr, tri_prim.data(), sizeof(uint32_t) * tri_prim.size(), hipMemcpyHostToDevice, st));
    size_t nt = std::max(n_tris, 1u);
    d_mf32.reserve(sizeof(TriRecF32) * nt);
    d_tri_bounds.reserve(sizeof(float) * 6 * nt);
    d_sorted_bounds.reserve(sizeof(float) * 6 * nt);
    d_prim_cbounds.reserve(sizeof(int) * 6 * (size_t)std::max(nvaos, 1u));
    d_keys[0].reserve(sizeof(unsigned long long) * nt);
    d_keys[1].reserve(sizeof(unsigned long long) * nt);
    d_vals[0].reserve(sizeof(uint32_t) * nt);
    d_vals[1].reserve(sizeof(uint32_t) * nt);
    uint32_t n_blocks = (n_tris + RS_TILE - 1) / RS_TILE;
    d_hist.reserve(sizeof(uint32_t) * 256 * (size_t)std::max(n_blocks, 1u));
    d_tris.reserve(sizeof(TriRec<T>) * nt);
    if (Num<T>::is_half)
        d_tris_f32.reserve(sizeof(TriRecF32) * nt);
    d_roots.reserve(sizeof(T) * 6 * (size_t)std::max(nvaos, 1u));
    LPRT_HIP(hipStreamSynchronize(st)); // host staging vectors above go out of scope
    if (nv)
        hipLaunchKernelGGL(k_convert_vertices<T>, dim3((nv + 255) / 256), dim3(256), 0, st, d_vbo32.as<lprt_vertex>(),
                           d_vbo.as<Vertex<T>>(), nv);
    built = false;
    plan_valid = false;
}

// Depth and record ranges of every primitive's tree for builder `kind` (0 / 1: median or Morton order, a perfect tree of depth
// floor(log2 n) with 1-2 triangles per leaf; 2: surface-area builder, depth BOUND ceil(log2(ceil(n / 2))) + 1, see k_sah_level).
// A function of the triangle counts: runs at upload and again only when a build asks for the other kind.
template <typename T> void GeometryStore<T>::set_layout(hipStream_t st, int kind) {
    if (kind == layout_kind)
        return;
    uint32_t inner_total = 0, leaf_total = 0;
    max_depth = 0;
    max_leaves = 1;
    for (uint32_t p = 0; p < n_vaos; p++) {
        BlasInfo &bi = host_blas[p];
        bi.depth = bi.n_tris ? floor_log2(bi.n_tris) : 0;
        if (kind == 2 && bi.n_tris > 3u) {
            const uint32_t need = (bi.n_tris + SAH_LEAF - 1u) / SAH_LEAF;
            uint32_t lg = 0;
            while ((1u << lg) < need)
                lg++;
            bi.depth = lg + 1u;
        }
        if (bi.depth > 28u) // the walks keep one pending bit per level in 32 bits (two-level walk: 4 bits per 2 levels in 64)
            throw std::string("a primitive with more than 2^28 triangles is not supported");
        bi.inner_base = inner_total;
        bi.leaf_base = leaf_total;
        if (bi.n_tris) {
            inner_total += (1u << bi.depth) - 1;
            leaf_total += 1u << bi.depth;
            max_depth = std::max(max_depth, bi.depth);
            max_leaves = std::max(max_leaves, 1u << bi.depth);
        }
    }
    n_inner = inner_total;
    n_leaves = leaf_total;
    d_blas.reserve(sizeof(BlasInfo) * (size_t)std::max(n_vaos, 1u));
    d_inner.reserve(sizeof(BlasNode<T>) * (size_t)std::max(n_inner, 1u));
    d_leaf.reserve(sizeof(uint32_t) * (size_t)std::max(n_leaves, 1u));
    if (kind == 2) {
        d_sah_nodes.reserve(4096u + 2u * sizeof(SahItem) * ((size_t)n_tris / (SAH_LEAF + 1u) + n_vaos) + 2u * sizeof(SahItem) * ((size_t)n_tris / (SAH_BIG + 1u) + n_vaos));
        d_fin.reserve(sizeof(uint32_t) * (size_t)std::max(n_tris, 1u));
    }
    if (n_vaos) {
        LPRT_HIP(hipMemcpyAsync(d_blas.ptr, host_blas.data(), sizeof(BlasInfo) * (size_t)n_vaos, hipMemcpyHostToDevice, st));
        LPRT_HIP(hipStreamSynchronize(st)); // (host_blas stays alive, but a caller may rebuild at once with other sizes)
    }
    layout_kind = kind;
    plan_valid = false;
    built = false;
}

template <typename T> void GeometryStore<T>::radix_sort(hipStream_t st, const std::vector<int> &shifts, int &cur) {
    const uint32_t n_blocks = (n_tris + RS_TILE - 1) / RS_TILE;
    for (int shift : shifts) {
        hipLaunchKernelGGL(k_radix_hist, dim3(n_blocks), dim3(RS_THREADS), 0, st, d_keys[cur].as<unsigned long long>(), n_tris, shift,
                           d_hist.as<uint32_t>(), n_blocks);
        hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(SCAN_THREADS), 0, st, d_hist.as<uint32_t>(), 256 * n_blocks);
        hipLaunchKernelGGL(k_radix_scatter, dim3(n_blocks), dim3(RS_THREADS), 0, st, d_keys[cur].as<unsigned long long>(),
                           d_vals[cur].as<uint32_t>(), d_keys[cur ^ 1].as<unsigned long long>(), d_vals[cur ^ 1].as<uint32_t>(), n_tris, shift,
                           d_hist.as<uint32_t>(), n_blocks);
        cur ^= 1;
    }
}

// see build_blas: everything about the median build that is a function of the triangle counts, computed and uploaded once
template <typename T> void GeometryStore<T>::make_plan(hipStream_t st, bool use_local) {
        // levels walked by the global sort: until every node of every primitive holds <= LOCAL_MAX triangles
        uint32_t global_levels = max_depth;
        std::vector<LocalJob> jobs;
        if (use_local) {
            global_levels = 0;
            for (const BlasInfo &bi : host_blas) {
                uint32_t d = 0;
                while (d < bi.depth && ((bi.n_tris + (1u << d) - 1) >> d) > LOCAL_MAX)
                    d++;
                global_levels = std::max(global_levels, d);
            }
            for (uint32_t p = 0; p < n_vaos; p++) {
                const BlasInfo &bi = host_blas[p];
                if (bi.depth <= global_levels)
                    continue; // this primitive's ordering is complete after the global levels
                for (uint32_t j = 0; j < (1u << global_levels); j++) {
                    uint32_t l = 0, r = bi.n_tris;
                    for (uint32_t b = 0; b < global_levels; b++) {
                        const uint32_t mid = l + (r - l) / 2;
                        if ((j >> (global_levels - 1 - b)) & 1u)
                            l = mid;
                        else
                            r = mid;
                    }
                    if (r - l > LOCAL_MAX)
                        throw std::string("internal: local BLAS segment larger than LOCAL_MAX");
                    if (r > l)
                        jobs.push_back(LocalJob{p, l, r - l, global_levels});
                }
            }
        }
        // dense node ranks for the key's high part (see k_seg_keys); positions when a finished primitive is too large for 16 bits
        std::vector<uint32_t> rank_bases((size_t)global_levels * n_vaos, 0u);
        std::vector<uint32_t> level_nodes(global_levels, 0u);
        bool dense = global_levels > 0;
        for (uint32_t d = 0; d < global_levels; d++) {
            uint32_t acc = 0;
            for (uint32_t p = 0; p < n_vaos; p++) {
                rank_bases[(size_t)d * n_vaos + p] = acc;
                const BlasInfo &bi = host_blas[p];
                acc += d < bi.depth ? (1u << d) : 1u;
                if (d >= bi.depth && bi.n_tris > 65536u)
                    dense = false;
            }
            level_nodes[d] = acc;
        }
        if (dense) {
            d_rank_bases.reserve(sizeof(uint32_t) * rank_bases.size());
            LPRT_HIP(hipMemcpyAsync(d_rank_bases.ptr, rank_bases.data(), sizeof(uint32_t) * rank_bases.size(), hipMemcpyHostToDevice, st));
        }
        plan_global_levels = global_levels;
        plan_dense = dense;
        plan_level_nodes = level_nodes;
        plan_jobs = (uint32_t)jobs.size();
        if (!jobs.empty()) {
            d_local_jobs.reserve(sizeof(LocalJob) * jobs.size());
            LPRT_HIP(hipMemcpyAsync(d_local_jobs.ptr, jobs.data(), sizeof(LocalJob) * jobs.size(), hipMemcpyHostToDevice, st));
        }
        LPRT_HIP(hipStreamSynchronize(st)); // the host vectors above go out of scope (once per upload, never inside a frame)
        plan_local = use_local;
        plan_valid = true;
}

template <typename T> void GeometryStore<T>::build_blas(hipStream_t st, int kind, bool use_local) {
    if (n_tris == 0 || n_vaos == 0) {
        built = true;
        return;
    }
    const bool use_morton = kind == 1;
    set_layout(st, kind == 2 ? 2 : 0);
    const uint32_t tb = (n_tris + 255) / 256;
    hipLaunchKernelGGL(k_init_bounds, dim3((6 * n_vaos + 255) / 256), dim3(256), 0, st, d_prim_cbounds.as<int>(), n_vaos);
    hipLaunchKernelGGL(k_tri_prepare, dim3((n_tris + 1023u) / 1024u), dim3(1024), 0, st, d_vbo32.as<lprt_vertex>(), d_ebo.as<uint32_t>(), n_tris,
                       d_tri_prim.as<uint32_t>(), d_mf32.as<TriRecF32>(), d_tri_bounds.as<float>(), d_prim_cbounds.as<int>());
    int cur = 0;
    if (kind == 2) {
        // surface-area builder: per level one launch over the queue of open small nodes (a wave each) and, while nodes of more
        // than SAH_BIG triangles can exist, one over the queue of big ones (a 1024-thread workgroup each); the queues are filled
        // on the device, the launches are sized by what a level can hold at most and return at once beyond the queue's end
        SahQueues q{};
        const uint32_t cap_small = n_tris / (SAH_LEAF + 1u) + n_vaos, cap_big = n_tris / (SAH_BIG + 1u) + n_vaos;
        q.levels = max_depth;
        q.count = d_sah_nodes.as<uint32_t>();
        {
            char *base = reinterpret_cast<char *>(d_sah_nodes.ptr) + 256u * ((2u * (max_depth + 1u) * sizeof(uint32_t) + 255u) / 256u);
            for (int par = 0; par < 2; par++) {
                q.items[0][par] = reinterpret_cast<SahItem *>(base);
                base += sizeof(SahItem) * (size_t)cap_small;
            }
            for (int par = 0; par < 2; par++) {
                q.items[1][par] = reinterpret_cast<SahItem *>(base);
                base += sizeof(SahItem) * (size_t)cap_big;
            }
        }
        (void)hipMemsetAsync(d_sah_nodes.ptr, 0, 2u * (max_depth + 1u) * sizeof(uint32_t), st);
        hipLaunchKernelGGL(k_iota, dim3(tb), dim3(256), 0, st, d_vals[0].as<uint32_t>(), n_tris);
        hipLaunchKernelGGL(k_iota, dim3(tb), dim3(256), 0, st, d_fin.as<uint32_t>(), n_tris);
        hipLaunchKernelGGL(k_sah_init, dim3((n_vaos + 255) / 256), dim3(256), 0, st, d_blas.as<BlasInfo>(), n_vaos, d_prim_cbounds.as<int>(), q);
        hipLaunchKernelGGL(k_tiny_roots<T>, dim3((n_vaos + 255) / 256), dim3(256), 0, st, d_blas.as<BlasInfo>(), n_vaos, d_tri_bounds.as<float>(),
                           d_roots.as<T>());
        for (uint32_t d = 0; d < max_depth; d++) {
            const uint64_t level_nodes = (uint64_t)n_vaos << std::min(d, 40u);
            const uint32_t n_small = (uint32_t)std::min<uint64_t>(level_nodes, cap_small), n_big = (uint32_t)std::min<uint64_t>(level_nodes, cap_big);
            // a node at depth d holds at most SAH_LEAF << (D - d) triangles: below that, no node of this level is a big one
            if (((uint64_t)SAH_LEAF << (max_depth - d)) > SAH_BIG)
                hipLaunchKernelGGL((k_sah_level<T, 16u>), dim3(n_big), dim3(1024), 0, st, d_blas.as<BlasInfo>(), d, q, d_vals[d & 1u].as<uint32_t>(),
                                   d_vals[(d & 1u) ^ 1u].as<uint32_t>(), d_fin.as<uint32_t>(), d_tri_bounds.as<float>(), d_inner.as<BlasNode<T>>(),
                                   d_roots.as<T>());
            hipLaunchKernelGGL((k_sah_level<T, 1u>), dim3((n_small + 15u) / 16u), dim3(1024), 0, st, d_blas.as<BlasInfo>(), d, q, d_vals[d & 1u].as<uint32_t>(),
                               d_vals[(d & 1u) ^ 1u].as<uint32_t>(), d_fin.as<uint32_t>(), d_tri_bounds.as<float>(), d_inner.as<BlasNode<T>>(),
                               d_roots.as<T>());
        }
        hipLaunchKernelGGL(k_emit_triangles<T>, dim3(tb), dim3(256), 0, st, d_fin.as<uint32_t>(), d_mf32.as<TriRecF32>(), d_tri_bounds.as<float>(),
                           d_tri_prim.as<uint32_t>(), d_vaos.as<lprt_mesh_vao>(), n_tris, d_tris.as<TriRec<T>>(),
                           Num<T>::is_half ? d_tris_f32.as<TriRecF32>() : nullptr, d_sorted_bounds.as<float>());
        sorted_buf = 2; // read-back: the leaf order is d_fin
        LPRT_HIP(hipGetLastError());
        built = true;
        return;
    }
    if (use_morton) {
        hipLaunchKernelGGL(k_morton_keys, dim3(tb), dim3(256), 0, st, d_tri_bounds.as<float>(), d_tri_prim.as<uint32_t>(),
                           d_prim_cbounds.as<int>(), n_tris, d_keys[0].as<unsigned long long>(), d_vals[0].as<uint32_t>());
        // radix passes over the 30 Morton bits + the primitive bits (key = prim << 32 | morton)
        uint32_t prim_bits = 0;
        while ((1u << prim_bits) < n_vaos)
            prim_bits++;
        std::vector<int> shifts = {0, 8, 16, 24};
        for (uint32_t b = 0; b < prim_bits; b += 8)
            shifts.push_back(32 + (int)b);
        radix_sort(st, shifts, cur);
    } else {
        uint32_t pos_bits = 1;
        while ((1ull << pos_bits) < n_tris)
            pos_bits++;
        std::vector<int> shifts = {0, 8};
        for (uint32_t b = 0; b < pos_bits; b += 8)
            shifts.push_back(16 + (int)b);
        d_seg_bounds.reserve(sizeof(int) * 6 * (size_t)n_tris);
        hipLaunchKernelGGL(k_iota, dim3(tb), dim3(256), 0, st, d_vals[cur].as<uint32_t>(), n_tris);
        // The plan of the build -- how many levels the global sort walks, the subtrees handed to k_local_levels, the dense node
        // ranks of every level -- depends on the triangle COUNTS only: it is made once per geometry upload (and builder
        // option) and kept on the device, so a per-frame rebuild (lprt_frame_settings.rebuild_blas) enqueues kernels and
        // nothing else: no host staging, no stream synchronisation inside the frame.
        if (!plan_valid || plan_local != use_local)
            make_plan(st, use_local);
        const uint32_t global_levels = plan_global_levels;
        const bool dense = plan_dense;
        const std::vector<uint32_t> &level_nodes = plan_level_nodes;
        for (uint32_t d = 0; d < global_levels; d++) {
            if (dense) {
                uint32_t bits = 1;
                while ((1u << bits) < level_nodes[d])
                    bits++;
                shifts = {0, 8};
                for (uint32_t b = 0; b < bits; b += 8)
                    shifts.push_back(16 + (int)b);
            }
            hipLaunchKernelGGL(k_init_bounds, dim3((6 * n_tris + 255) / 256), dim3(256), 0, st, d_seg_bounds.as<int>(), n_tris);
            hipLaunchKernelGGL(k_seg_bounds, dim3(tb), dim3(256), 0, st, d_vals[cur].as<uint32_t>(), d_tri_bounds.as<float>(),
                               d_tri_prim.as<uint32_t>(), d_blas.as<BlasInfo>(), n_tris, d, d_seg_bounds.as<int>());
            hipLaunchKernelGGL(k_seg_keys, dim3(tb), dim3(256), 0, st, d_vals[cur].as<uint32_t>(), d_tri_bounds.as<float>(),
                               d_tri_prim.as<uint32_t>(), d_blas.as<BlasInfo>(), n_tris, d, d_seg_bounds.as<int>(),
                               d_keys[cur].as<unsigned long long>(), dense ? d_rank_bases.as<uint32_t>() + (size_t)d * n_vaos : nullptr);
            radix_sort(st, shifts, cur);
        }
        if (plan_jobs)
            hipLaunchKernelGGL(k_local_levels, dim3(plan_jobs), dim3(256), 0, st, d_vals[cur].as<uint32_t>(), d_tri_bounds.as<float>(),
                               d_blas.as<BlasInfo>(), d_local_jobs.as<LocalJob>());
    }
    sorted_buf = cur;
    hipLaunchKernelGGL(k_emit_triangles<T>, dim3(tb), dim3(256), 0, st, d_vals[cur].as<uint32_t>(), d_mf32.as<TriRecF32>(),
                       d_tri_bounds.as<float>(), d_tri_prim.as<uint32_t>(), d_vaos.as<lprt_mesh_vao>(), n_tris,
                       d_tris.as<TriRec<T>>(), Num<T>::is_half ? d_tris_f32.as<TriRecF32>() : nullptr,
                       d_sorted_bounds.as<float>());
    hipLaunchKernelGGL(k_leaf_boxes<T>, dim3((max_leaves + 255) / 256, n_vaos), dim3(256), 0, st, d_blas.as<BlasInfo>(),
                       d_sorted_bounds.as<float>(), d_leaf.as<uint32_t>(), d_inner.as<BlasNode<T>>(), d_roots.as<T>());
    for (int d = (int)max_depth - 1; d >= 0; d--)
        hipLaunchKernelGGL(k_inner_level<T>, dim3(((1u << d) + 255) / 256, n_vaos), dim3(256), 0, st, d_blas.as<BlasInfo>(),
                           (uint32_t)d, d_inner.as<BlasNode<T>>(), d_roots.as<T>());
    LPRT_HIP(hipGetLastError());
    built = true;
}

// aabbs: DEVICE pointer (the caller has already enqueued the copy on st)
template <typename T> void TlasStore<T>::build(hipStream_t st, const lprt_aabb *aabbs, uint32_t n, const SceneSetup<T> &su) {
    n_objects = n;
    depth = n ? floor_log2(n) : 0;
    d_slots.reserve(sizeof(ObjSlot<T>) * (size_t)std::max(n, 1u));
    d_leaf.reserve(sizeof(uint32_t) * (size_t)(1u << depth));
    d_inner.reserve(sizeof(InnerNode<T>) * (size_t)std::max((1u << depth) - 1, 1u));
    d_root.reserve(sizeof(T) * 6);
    hipLaunchKernelGGL(k_build_tlas<T>, dim3(1), dim3(256), 0, st, aabbs, n, depth, d_slots.as<ObjSlot<T>>(), d_leaf.as<uint32_t>(),
                       d_inner.as<InnerNode<T>>(), d_root.as<T>(), su);
    LPRT_HIP(hipGetLastError());
}

template struct GeometryStore<float>;
template struct GeometryStore<h16>;
template struct TlasStore<float>;
template struct TlasStore<h16>;

} // namespace lprt
