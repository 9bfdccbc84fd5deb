// Owners of the static geometry / BLAS set and of the per-frame TLAS.
#pragma once
#include "ctx.hpp"
#include "scene_dev.hpp"
#include <algorithm>

namespace lprt {

template <typename T> struct GeometryStore {
    uint32_t n_vertices = 0, n_indices = 0, n_tris = 0, n_vaos = 0, n_inner = 0, n_leaves = 0, max_depth = 0, max_leaves = 1;
    bool built = false;
    int sorted_buf = 0;
    std::vector<lprt_mesh_vao> host_vaos;
    std::vector<BlasInfo> host_blas;
    DevBuf d_vbo32, d_vbo, d_ebo, d_vaos, d_tri_prim, d_blas;
    DevBuf d_mf32, d_tri_bounds, d_sorted_bounds, d_prim_cbounds, d_keys[2], d_vals[2], d_hist;
    DevBuf d_tris, d_tris_f32, d_inner, d_leaf, d_roots, d_seg_bounds, d_local_jobs, d_rank_bases, d_sah_nodes, d_fin;
    int layout_kind = -1;   // which builder host_blas / d_blas are laid out for (set_layout)
    int builder_hint = 0;   // the kind upload() lays out for (the context's builder option at upload time)

    // RDResource::{objects_vbo, objects_ebo, objects_vao} + vertex conversion (memory.hpp:568-619)
    void upload(hipStream_t st, const lprt_vertex *vbo, uint32_t nv, const uint32_t *ebo, uint32_t ni, const lprt_mesh_vao *vaos,
                uint32_t nvaos);
    // compute_m + ObjectBVH for every primitive, all on the device
    void build_blas(hipStream_t st, int kind = 0, bool use_local = true); // kind 0 median, 1 Morton, 2 surface-area (LPRT_OPT_BLAS_BUILDER); LPRT_OPT_BLAS_LOCAL
    void set_layout(hipStream_t st, int kind);
    const uint32_t *leaf_order() const { return sorted_buf == 2 ? d_fin.as<uint32_t>() : d_vals[sorted_buf].template as<uint32_t>(); }
    void radix_sort(hipStream_t st, const std::vector<int> &shifts, int &cur);
    // the plan of the median build (a function of the triangle counts and of use_local): made once, kept on the device
    bool plan_valid = false, plan_local = true, plan_dense = false;
    uint32_t plan_global_levels = 0, plan_jobs = 0;
    std::vector<uint32_t> plan_level_nodes;
    void make_plan(hipStream_t st, bool use_local);
};

template <typename T> struct TlasStore {
    uint32_t n_objects = 0, depth = 0;
    DevBuf d_slots, d_leaf, d_inner, d_root;
    void build(hipStream_t st, const lprt_aabb *aabbs_dev, uint32_t n, const SceneSetup<T> &setup);
};

} // namespace lprt
