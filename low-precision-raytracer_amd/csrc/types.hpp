// Device data layouts (HBM) shared by the kernels.  See DESIGN.md "Data layout in HBM".
#pragma once
#include "dmath.hpp"
#include "../../include/lprt.h"

namespace lprt {

// ---- geometry -----------------------------------------------------------------------------------
// RDVertex<DataT> (rt/rtrt/memory.hpp:26-41): 16 scalars = 64 B (f32) / 32 B (f16), loaded as 16-B vectors.
template <typename T> struct alignas(sizeof(T) * 4) Vertex {
    T position[3], normal[3], tangent[3], color[3], uv0[2], uv1[2];
};

// One triangle of a BLAS, stored in Morton-sorted leaf order.  Holds what ray_triangle_intersection
// gathers through three levels of indirection in the reference (vaos -> ebo -> vbo, M_shift[offset/3];
// rt/rtrt/cuda.hpp:199-209): the rows of M_shift, the position of vertex 2 and inobject_offset.
template <typename T> struct TriRec;
template <> struct alignas(16) TriRec<float> { // 64 B
    float m0[3], v2x, m1[3], v2y, m2[3], v2z;
    uint32_t inobject_offset, pad[3];
};
template <> struct alignas(16) TriRec<h16> { // 32 B
    h16 m0[3], v2x, m1[3], v2y, m2[3], v2z;
    uint32_t inobject_offset, pad;
};
// fp32 edge-zone re-test operands for the half instantiation (M_shift_f32, position_f32; cuda.hpp:271-276)
struct alignas(16) TriRecF32 { // 48 B
    float m0[3], v2x, m1[3], v2y, m2[3], v2z;
};

// ---- acceleration structures -------------------------------------------------------------------
// TLAS node: perfect binary tree over the Morton-sorted instances, heap order (root = 1, children 2i / 2i+1), all
// leaves at depth D = floor(log2 n); a leaf owns 1 or 2 consecutive sorted instances (leaf-word array).  An inner node
// stores the boxes of BOTH children, so one fetch feeds two slab tests.
template <typename T> struct alignas(sizeof(T) * 4) InnerNode { // 48 B (f32) / 24 B (f16)
    T lo0[3], hi0[3], lo1[3], hi1[3];
};
// One node of a BLAS: the boxes of both children and, per child, a LEAF WORD -- LPRT_CHILD_INNER when the child is an inner
// node (its own record sits at heap slot 2i + k: no child pointers), else (first << 3) | count: the child is a leaf holding
// `count` (0..7; 0 = no child, box inverted) triangles from `first` on, relative to the primitive's first TriRec, in leaf
// order.  Leaves can sit at any depth <= D (the surface-area builder stops as soon as a node holds <= 2 triangles), and a
// leaf needs no record of its own: what a ray needs to test its triangles arrives with the parent's record, in the fetch that
// brought the leaf's box.  32 B (f16) / 64 B (f32): one or two 32-byte sectors, never straddling a 128-byte line.
#define LPRT_CHILD_INNER 0xffffffffu
template <typename T> struct BlasNode;
template <> struct alignas(32) BlasNode<h16> {
    h16 lo0[3], hi0[3], lo1[3], hi1[3];
    uint32_t leaf[2];
};
template <> struct alignas(64) BlasNode<float> {
    float lo0[3], hi0[3], lo1[3], hi1[3];
    uint32_t leaf[2], pad[2];
};
static_assert(sizeof(BlasNode<h16>) == 32 && sizeof(BlasNode<float>) == 64, "BlasNode layout");
struct BlasInfo {
    uint32_t n_tris;     // 0: empty primitive
    uint32_t depth;      // D: no node lies deeper (median builder: every leaf at exactly D; surface-area builder: an upper bound)
    uint32_t tri_base;   // first TriRec of this primitive (sorted order)
    uint32_t inner_base; // first BlasNode (2^D - 1 heap slots: nodes of depth 0 .. D-1)
    uint32_t leaf_base;  // first leaf word (2^D of them): (first_sorted_tri << 1) | (count - 1)
    uint32_t root_base;  // index of the root box in root_boxes
    uint32_t pad[2];
};
// TLAS leaf slot: one object instance in sorted order (aabbs_W + inscene_objid, scene_bvh.hpp:46-47)
template <typename T> struct alignas(16) ObjSlot {
    T lo[3], hi[3];
    uint32_t inscene_objid;
};
// Everything a ray needs to enter one object instance, in TLAS (Morton) slot order: the world box (aabbs_W), the
// world->local transform and the BLAS header + root box.  The reference reaches these through four dependent
// fetches (aabbs_W[i], inscene_objid[i] -> objects_constant[id] -> objects_bvh[vao] -> nodes[0]; cuda.hpp:544-563, 421-437).
// The head is what a ray needs to ENTER the instance (BLAS header, root box, flags): 64 / 48 bytes fetched in one round trip
// (trace.hpp::load_obj_head); the world box is read by the instance scan, the matrix only when it is not the identity.
template <typename T> struct alignas(16) ObjHead {
    uint32_t inscene_objid, vao_index, n_tris, depth, tri_base, inner_base;
    uint32_t pad[2];          // pad[0] bit 0: W2L is exactly the identity (make_objrec)
    T root_lo[3], root_hi[3]; // BLAS root box (object space)
    T hpad[2];
};
template <typename T> struct alignas(16) ObjRec : ObjHead<T> {
    T lo[3], hi[3]; // world-space box
    M4<T> W2L;
};
static_assert(sizeof(ObjHead<float>) == 64 && sizeof(ObjHead<h16>) == 48, "ObjHead is a whole number of 16-byte vectors");
// CUDAObjectConstantData<DataT> (memory.hpp:279-301)
template <typename T> struct alignas(16) ObjConst {
    M4<T> L2W, W2L;
    uint32_t materialid, objectid, pad[2];
};
// CUDAMaterial<DataT> (memory.hpp:303-332), fields read by shade()
template <typename T> struct alignas(16) MaterialT {
    T color[3], emission[3];
    T metallic, roughness;
    uint32_t double_sided, uv_color, tex_color, pad;
};
template <typename T> struct alignas(16) LightT { // RDLight<DataT> (memory.hpp:43-51)
    T direction[3], intensity[3], position[3];
    uint32_t type;
};
struct Texture { // RDTexture (memory.hpp:334-369): RGBA8, bilinear, wrap
    const uint32_t *texels;
    uint32_t width, height, srgb, pad;
};

template <typename T> struct SceneDev {
    // TLAS
    const InnerNode<T> *tlas_inner;
    const uint32_t *tlas_leaf;
    const T *tlas_root; // 6 scalars
    const ObjSlot<T> *tlas_slots;
    const ObjRec<T> *objrecs; // n_objects records in slot order
    uint32_t tlas_depth, n_objects;
    uint32_t tlas_mode;   // 0: tree, 1: flat scan + per-lane candidate list (<= 32 objects)
    uint32_t axis_reject; // 1 (default): trace.hpp's stationary-axis rejection on; 0: the reference's plain slab test
    uint32_t stream_refill_min, stream_node_min; // ray-stream tunables (stream.hpp), lane counts
    uint32_t stream_sched;                       // 1: one phase per pass, the one most lanes wait for; 0: all phases every pass
    uint32_t stream_xcd;                         // 1: eight item ranges with a counter each, drained XCD by XCD (stream.hpp); 0: one counter
    const ObjConst<T> *objects;
    // BLAS
    const BlasInfo *blas;
    const T *blas_roots; // 6 scalars per primitive
    const BlasNode<T> *blas_inner;
    const uint32_t *blas_leaf;
    const TriRec<T> *tris;
    const TriRecF32 *tris_f32; // half instantiation only
    // attributes
    const Vertex<T> *vbo;
    const uint32_t *ebo;
    const lprt_mesh_vao *vaos;
    const MaterialT<T> *materials;
    const LightT<T> *lights;
    const Texture *textures;
    const float *srgb_lut; // 256 entries: 8-bit sRGB -> linear
    uint32_t n_lights, n_vaos;
    // camera (RDCamera<DataT>, memory.hpp:382-405)
    M4<T> cam_L2W;
    T cam_fovy;
    // skybox
    uint32_t sky_valid, sky_texture;
    T sky_dx, sky_dy, sky_exposure;
    // array extents, used only by LPRT_BOUNDS builds (index validation instead of a GPU fault)
    uint32_t lim_vertices, lim_indices, lim_tris, lim_inner, lim_leaves, lim_objects, lim_materials, lim_pad;
    uint32_t *debug; // [16] error code, then operands
};

#ifdef LPRT_BOUNDS
#define LPRT_CHECK(sc, cond, code, a, b)                                                                               \
    do {                                                                                                               \
        if (!(cond)) {                                                                                                 \
            if ((sc).debug && atomicCAS((sc).debug, 0u, (uint32_t)(code)) == 0u) {                                     \
                (sc).debug[1] = (uint32_t)(a);                                                                         \
                (sc).debug[2] = (uint32_t)(b);                                                                         \
            }                                                                                                          \
            return;                                                                                                    \
        }                                                                                                              \
    } while (0)
#define LPRT_CHECK_RET(sc, cond, code, a, b, ret)                                                                      \
    do {                                                                                                               \
        if (!(cond)) {                                                                                                 \
            if ((sc).debug && atomicCAS((sc).debug, 0u, (uint32_t)(code)) == 0u) {                                     \
                (sc).debug[1] = (uint32_t)(a);                                                                         \
                (sc).debug[2] = (uint32_t)(b);                                                                         \
            }                                                                                                          \
            return ret;                                                                                                \
        }                                                                                                              \
    } while (0)
#else
#define LPRT_CHECK(sc, cond, code, a, b) ((void)0)
#define LPRT_CHECK_RET(sc, cond, code, a, b, ret) ((void)0)
#endif
#ifdef LPRT_BOUNDS_A
#define LPRT_CHECKA_RET(sc, cond, code, a, b, ret)                                                                     \
    do {                                                                                                               \
        if (!(cond)) {                                                                                                 \
            if ((sc).debug && atomicCAS((sc).debug, 0u, (uint32_t)(code)) == 0u) {                                     \
                (sc).debug[1] = (uint32_t)(a);                                                                         \
                (sc).debug[2] = (uint32_t)(b);                                                                         \
            }                                                                                                          \
            return ret;                                                                                                \
        }                                                                                                              \
    } while (0)
#else
#define LPRT_CHECKA_RET(sc, cond, code, a, b, ret) LPRT_CHECK_RET(sc, cond, code, a, b, ret)
#endif

struct Ray32 { // host-facing ray record == lprt_ray
    float o[3], d[3], tmin, tmax;
    uint32_t skip_vao, skip_off;
};

template <typename T> struct RayT {
    V3<T> o, d;
};
template <typename T> struct Isect { // Intersection<DataT>, cuda.hpp:50-57
    uint32_t inscene_objid, inobject_offset;
    V3<T> bary;
    T t;
    bool hit;
};

struct Counters {
    unsigned long long rays, tlas_nodes, aabbw, objects, blas_nodes, tris, fallbacks;
    unsigned long long wave_steps[3], lane_steps[3]; // lane occupancy per phase of the walk (lprt_counters)
};

// image window: rows [row_begin, row_end) of a W x H frame (one GPU's strip + halo)
struct Window {
    uint32_t W, H, row_begin, row_end;
    __host__ __device__ uint32_t rows() const { return row_end - row_begin; }
    __host__ __device__ uint32_t npix() const { return W * (row_end - row_begin); }
};

} // namespace lprt
