// Per-frame scene conversion shared by the stand-alone convert kernels and the fused scene set-up in k_build_tlas.
#pragma once
#include "types.hpp"

namespace lprt {

// everything lprt_set_scene needs done on the device besides the TLAS itself
template <typename T> struct SceneSetup {
    const lprt_object *objects32;
    ObjConst<T> *objects;
    uint32_t n_objects;
    const lprt_light *lights32;
    LightT<T> *lights;
    uint32_t n_lights;
    const BlasInfo *blas; // null: BLAS not built yet, object records are made later
    const T *roots;
    uint32_t n_vaos;
    ObjRec<T> *objrecs;
};

template <typename T> LPRT_DEV void convert_object(const lprt_object &in, ObjConst<T> &o) {
    for (int k = 0; k < 16; k++) {
        o.L2W.m[k] = cvt<T>(in.transform_L2W[k]);
        o.W2L.m[k] = cvt<T>(in.transform_W2L[k]);
    }
    o.materialid = in.materialid;
    o.objectid = in.objectid;
    o.pad[0] = o.pad[1] = 0;
}
template <typename T> LPRT_DEV void convert_light(const lprt_light &in, LightT<T> &l) {
    for (int k = 0; k < 3; k++) {
        l.direction[k] = cvt<T>(in.direction[k]);
        l.intensity[k] = cvt<T>(in.intensity[k]);
        l.position[k] = cvt<T>(in.position[k]);
    }
    l.type = in.type;
}
// ObjRec for one TLAS slot
template <typename T> LPRT_DEV void make_objrec(const ObjSlot<T> &slot, const ObjConst<T> &oc, const BlasInfo *blas, const T *roots, uint32_t n_vaos, ObjRec<T> &r) {
    for (int k = 0; k < 3; k++) {
        r.lo[k] = slot.lo[k];
        r.hi[k] = slot.hi[k];
    }
    r.W2L = oc.W2L;
    // pad[0] bit 0: W2L is exactly the identity in DataT -> the instance's rays are the world rays (x*1 + y*0 + z*0 + 0 is exact
    // and w == 1 skips the division: transform_ray would return the ray unchanged), traversal skips the transform, the
    // three reciprocals and the 64 bytes of matrix
    bool ident = true;
    for (int k = 0; k < 16; k++)
        ident = ident && oc.W2L.m[k] == ((k % 5 == 0) ? (T)1 : (T)0);
    r.pad[0] = ident ? 1u : 0u;
    r.pad[1] = 0;
    r.inscene_objid = slot.inscene_objid;
    r.vao_index = oc.objectid;
    BlasInfo bi{};
    if (oc.objectid < n_vaos)
        bi = blas[oc.objectid];
    r.n_tris = bi.n_tris;
    r.depth = bi.depth;
    r.tri_base = bi.tri_base;
    r.inner_base = bi.inner_base;
    for (int k = 0; k < 3; k++) {
        r.root_lo[k] = bi.n_tris ? roots[6 * bi.root_base + k] : (T)0;
        r.root_hi[k] = bi.n_tris ? roots[6 * bi.root_base + 3 + k] : (T)0;
    }
}

} // namespace lprt
