// Ray-casting kernels: host ray batches (lprt_trace_rays), primary rays -> G-buffer (G1, G2),
// GI rays (TG) and shadow rays (TD).
#include "kernels.hpp"
#include <algorithm>
#include "scene_dev.hpp"
#include "shade.hpp"
#include "trace.hpp"

namespace lprt {

// ---- lprt_trace_rays: cast_ray<find_any,false> over a batch (cuda.hpp:622-631) -------------------------
LPRT_VAR_BEGIN
template <typename T, bool ANY, bool COUNT, bool WIDE>
__global__ void __launch_bounds__(256) k_trace_rays(SceneDev<T> sc, const lprt_ray *rays, uint32_t n, lprt_hit *hits, Counters *counters) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    Cnt<COUNT> cnt;
    if (i < n) {
        const lprt_ray in = rays[i];
        RayT<T> ray;
        ray.o = mk3(cvt<T>(in.source[0]), cvt<T>(in.source[1]), cvt<T>(in.source[2]));
        ray.d = mk3(cvt<T>(in.direction[0]), cvt<T>(in.direction[1]), cvt<T>(in.direction[2]));
        Isect<T> is = cast_ray<T, ANY, COUNT, WIDE>(sc, ray, in.skip_vao, in.skip_inobject_offset, cvt<T>(in.min_distance),
                                              cvt<T>(in.max_distance), cnt);
        lprt_hit h;
        h.hit = is.hit ? 1u : 0u;
        h.inscene_objid = is.inscene_objid;
        h.inobject_offset = is.inobject_offset;
        h.t = (float)is.t;
        h.barycentric[0] = (float)is.bary.x;
        h.barycentric[1] = (float)is.bary.y;
        h.barycentric[2] = (float)is.bary.z;
        hits[i] = h;
    }
    flush_counters<COUNT>(cnt, counters);
}

template <typename T, bool ANY, bool WIDE>
static void launch_trace_rays_w(hipStream_t st, const SceneDev<T> &sc, const lprt_ray *rays, uint32_t n, lprt_hit *hits, Counters *counters) {
    dim3 grid((n + 255) / 256), block(256);
    if (counters)
        hipLaunchKernelGGL((k_trace_rays<T, ANY, true, WIDE>), grid, block, 0, st, sc, rays, n, hits, counters);
    else
        hipLaunchKernelGGL((k_trace_rays<T, ANY, false, WIDE>), grid, block, 0, st, sc, rays, n, hits, counters);
}
template <typename T>
void launch_trace_rays(hipStream_t st, const SceneDev<T> &sc, const lprt_ray *rays, uint32_t n, bool any, bool wide, lprt_hit *hits, Counters *counters) {
    if (n == 0)
        return;
    if (any && wide)
        launch_trace_rays_w<T, true, true>(st, sc, rays, n, hits, counters);
    else if (any)
        launch_trace_rays_w<T, true, false>(st, sc, rays, n, hits, counters);
    else if (wide)
        launch_trace_rays_w<T, false, true>(st, sc, rays, n, hits, counters);
    else
        launch_trace_rays_w<T, false, false>(st, sc, rays, n, hits, counters);
}
#ifndef LPRT_REFSLAB_TU
template void launch_trace_rays<float>(hipStream_t, const SceneDev<float> &, const lprt_ray *, uint32_t, bool, bool, lprt_hit *, Counters *);
#endif
template void launch_trace_rays<h16>(hipStream_t, const SceneDev<h16> &, const lprt_ray *, uint32_t, bool, bool, lprt_hit *, Counters *);
LPRT_VAR_END

#ifndef LPRT_REFSLAB_TU // everything below exists once

// ---- material conversion (CUDAMaterial in DataT); objects and lights are converted by k_build_tlas (scene_dev.hpp) ----
template <typename T> __global__ void k_convert_materials(const lprt_material *in, MaterialT<T> *out, uint32_t n) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    MaterialT<T> m;
    for (int k = 0; k < 3; k++) {
        m.color[k] = cvt<T>(in[i].color[k]);
        m.emission[k] = cvt<T>(in[i].emission[k]);
    }
    m.metallic = cvt<T>(in[i].metallic);
    m.roughness = cvt<T>(in[i].roughness);
    m.double_sided = in[i].double_sided;
    m.uv_color = in[i].uv_color;
    m.tex_color = in[i].tex_color;
    m.pad = 0;
    out[i] = m;
}
// ObjRec[s] for TLAS slot s, gathered once per scene/BLAS change
template <typename T>
__global__ void k_make_objrecs(const ObjSlot<T> *slots, const ObjConst<T> *objects, const BlasInfo *blas, const T *roots, uint32_t n, uint32_t n_vaos,
                               ObjRec<T> *out) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n)
        return;
    ObjRec<T> r;
    make_objrec(slots[s], objects[slots[s].inscene_objid], blas, roots, n_vaos, r);
    out[s] = r;
}
template <typename T>
void launch_make_objrecs(hipStream_t st, const ObjSlot<T> *slots, const ObjConst<T> *objects, const BlasInfo *blas, const T *roots, uint32_t n,
                         uint32_t n_vaos, ObjRec<T> *out) {
    if (n)
        hipLaunchKernelGGL(k_make_objrecs<T>, dim3((n + 63) / 64), dim3(64), 0, st, slots, objects, blas, roots, n, n_vaos, out);
}

template <typename T> void launch_convert_materials(hipStream_t st, const lprt_material *in, MaterialT<T> *out, uint32_t n) {
    if (n)
        hipLaunchKernelGGL(k_convert_materials<T>, dim3((n + 63) / 64), dim3(64), 0, st, in, out, n);
}
// blockIdx.y = job; 16-byte words when both ends and the length allow it, bytes otherwise (uint8 planes, odd widths)
__global__ void k_copy_jobs(CopyJobs jobs) {
    const uint32_t j = blockIdx.y;
    const char *src = static_cast<const char *>(jobs.src[j]);
    char *dst = static_cast<char *>(jobs.dst[j]);
    const size_t n = jobs.bytes[j];
    const bool wide = ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | n) & 15u) == 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wide) {
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src);
        uint4 *d4 = reinterpret_cast<uint4 *>(dst);
        for (size_t i = i0; i < n / 16; i += stride)
            d4[i] = s4[i];
    } else {
        for (size_t i = i0; i < n; i += stride)
            dst[i] = src[i];
    }
}
void launch_copy_jobs(hipStream_t st, const CopyJobs &jobs, size_t longest) {
    const uint32_t bx = (uint32_t)std::min<size_t>((longest / 16 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(k_copy_jobs, dim3(bx, jobs.n), dim3(256), 0, st, jobs);
}

// test hook behind lprt_debug_sample_texture: the same fetch shade() makes
__global__ void k_sample_texture(const Texture *textures, const float *lut, uint32_t handle, const float *uv, uint32_t n, float *rgba) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float4 c = tex2d(textures, lut, handle, uv[2 * i], uv[2 * i + 1]);
    rgba[4 * i + 0] = c.x;
    rgba[4 * i + 1] = c.y;
    rgba[4 * i + 2] = c.z;
    rgba[4 * i + 3] = c.w;
}
void launch_sample_texture(hipStream_t st, const Texture *textures, const float *lut, uint32_t handle, const float *uv, uint32_t n, float *rgba) {
    hipLaunchKernelGGL(k_sample_texture, dim3((n + 255) / 256), dim3(256), 0, st, textures, lut, handle, uv, n, rgba);
}
__global__ void k_fmath(int op, const float *in, uint32_t n, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float x = in[i];
    out[i] = op == 0 ? lprt_logf(x) : op == 1 ? lprt_sinf(x) : op == 2 ? lprt_cosf(x) : lprt_tanf(x);
}
void launch_fmath(hipStream_t st, int op, const float *in, uint32_t n, float *out) {
    hipLaunchKernelGGL(k_fmath, dim3((n + 255) / 256), dim3(256), 0, st, op, in, n, out);
}

#define INST(T)                                                                                                        \
    template void launch_make_objrecs<T>(hipStream_t, const ObjSlot<T> *, const ObjConst<T> *, const BlasInfo *, const T *, uint32_t, uint32_t,    \
                                         ObjRec<T> *);                                                                 \
    template void launch_convert_materials<T>(hipStream_t, const lprt_material *, MaterialT<T> *, uint32_t);
INST(float)
INST(h16)
#undef INST

#endif // LPRT_REFSLAB_TU

} // namespace lprt
