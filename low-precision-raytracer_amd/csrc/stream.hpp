// Ray streams: persistent waves that keep their lanes full.
//
// The per-pixel traversal kernels (pipeline.hip) give every lane one ray and the wave runs until its LONGEST ray is done.
// In a large scene that is the whole cost: rocprofv3 on the 262 k-triangle atrium (profiles/r02a_*) shows 26-29 % of the
// lanes active per vector instruction in k_trace_gi / k_trace_di against 97 % in k_shade -- the BVH walk of an incoherent
// ray is 3-4x longer for the unlucky lane than for the average one.  Here a launch is a STREAM of work items (one per
// potential ray, numbered so that 64 consecutive items are the rays of one 8x8-pixel block, per light slot for shadow
// rays) and a fixed set of waves that pull from it: a wave takes 64 items at a time (one atomic add), prepares their rays with all of
// its lanes and parks them in a small LDS ring; a lane whose ray has finished takes the next parked ray (ballot + prefix
// count), so the lanes of a wave stay busy until the stream runs dry.  The walk is the same arithmetic as trace.hpp (slab tests, triangle test, culls, skip rule, the one- and
// two-level steps), rewritten as a resumable per-lane state machine:
//     fetch  ->  [ next candidate instance | TLAS step ]  ->  node steps (while-while)  ->  leaf triangles  ->  ...
// Results go to planes (visibility byte per light slot, hit record per pixel); the per-pixel stages that used to follow in
// the same thread (shade #2, accumulate + compose) run as streaming kernels of their own (pipeline.hip: k_shade2, k_accum_di), at full lane use.
//
// Exit condition reached by every wave: the stream counter only grows; a wave leaves when a fetch found the stream
// exhausted and none of its lanes holds a ray.  Replaces the launch structure of trace_gi / trace_di_light
// (rt/rtrt/cuda.hpp:2058-2141); what a ray reports is unchanged (tests: hit records and frames against the per-pixel
// kernels, bit for bit except exact-t ties, and against the oracle).
#pragma once
#include "trace.hpp"
#include <type_traits>

namespace lprt {


// box test of the walk: trace.hpp's slab test + culls, the stationary-axis rejection as a run-time (wave-skipped) branch
template <typename T>
LPRT_DEV bool stream_box(const T *lo, const T *hi, const RaySlab<T> &rs, float best_t, float min_d, float max_d, float &tnear, bool wave_slow) {
    float x;
    bool h;
    if (wave_slow) { // rare: some lane's ray does not move along an axis (see trace.hpp); wave-uniform, evaluated once per node loop
        h = rs.slow ? slab_test_general<T, false>(lo, hi, rs, tnear, x) : slab_test_moving<T, false>(lo, hi, rs, tnear, x);
        return h && tnear < best_t && tnear < max_d && x > min_d;
    }
#ifndef LPRT_REFSLAB_TU
    if (Num<T>::is_half)
        return box_hit_moving<T>(lo, hi, rs, SlabOps<T>::mn(best_t, max_d), cull_floor(min_d), tnear); // the culls folded in (trace.hpp)
#endif
    h = slab_test_moving<T, false>(lo, hi, rs, tnear, x);
    return h && tnear < best_t && tnear < max_d && x > min_d;
}

// A ray made ready for traversal: what the fetch of one item yields (ray, window, skip record, reciprocals) and the result of
// its world-box scan.  Rays are PREPARED 64 at a time, by all lanes of the wave (the scan of the instances' world boxes is
// ~30 instructions per box: done by the few lanes that happen to be idle it cost more than the refill saved), parked in a
// per-wave LDS ring, and handed to lanes as they fall idle -- a pop is a dozen instructions, so a lane waits for work for
// one iteration of the loop, not until three quarters of the wave have run dry.
template <typename T> struct PrepRay { // 48 B (f32) / 32 B (f16)
    V3<T> o, d;
    T tmin, tmax;
    uint32_t skip_vao, skip_off, item, cand; // cand: candidate-instance bits (flat scan) or "the TLAS root box is hit" (tree)
};
// What a lane needs of its ray BETWEEN instances only -- the world-space ray and its reciprocals, the skipped mesh, the item
// number, the state of the top-level walk -- lives in a per-lane LDS record, not in registers: the walk inside an instance
// (node and leaf phases, where the time goes) never touches it.  Held in registers, these ~15 words were what the compiler
// spilled to scratch at the 128-register budget (64 B per lane in k_stream_gi): 3 GB of scratch written and re-read per launch
// through the same L2 that should hold the scene (WRITE_SIZE 2.9 GB for a kernel whose results are 0.27 GB), and 2 MB of
// scratch lines per XCD competing with the BVH for 4 MB of L2.
template <typename T> struct WorldRay { // 60 B (f32) / 48 B (f16) per lane
    V3<T> o, d;
    float ix, iy, iz;
    uint32_t slow;
    uint32_t skip_vao, item;
    uint32_t top;   // flat scan: candidate-instance bits still to visit; tree: the TLAS node the walk stands on
    uint32_t tpend; // tree: deferred siblings, one bit per level
    uint32_t tcl;   // tree: depth | leaf cursor << 8 | walk alive << 16
};
#ifndef LPRT_STREAM_RING
#define LPRT_STREAM_RING 128u // prepared rays a wave can park (power of two, >= 128: a batch of 64 is prepared when 64 slots are free)
#endif

// Src: load(item, ray, tmin, tmax, skip_vao, skip_off) -> bool (false: the item carries no ray), store(item, Isect).
// Ray order (pipeline.hip "ray sort"): with `order` the stream's j-th work item is order[j] instead of j -- the items that carry
// a ray, counting-sorted by (cell of the ray's origin, direction octant) -- and the stream is *n_sorted long: 64 consecutive
// items are then rays that start in one cell and head the same way, whatever pixels they belong to, and an XCD's range of
// the stream is one region of the scene.  Results do not depend on the order (a ray's record goes to its own item).
struct StreamOrder {
    const uint32_t *order;    // null: identity
    const uint32_t *n_sorted; // device word: items in `order`
};
template <typename T, bool ANY, bool COUNT, bool WIDE, typename Src>
LPRT_DEV void trace_stream(const SceneDev<T> &sc, Src &src, uint32_t n_items_all, const StreamOrder &so, uint32_t *stream_counter, Cnt<COUNT> &cnt) {
    const uint32_t n_items = so.order ? *so.n_sorted : n_items_all;
    __shared__ PrepRay<T> s_ring[4][LPRT_STREAM_RING];
    __shared__ WorldRay<T> s_world[256];
    PrepRay<T> *ring = s_ring[threadIdx.x >> 6];
    WorldRay<T> &world = s_world[threadIdx.x];
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    bool has_ray = false, in_obj = false, exhausted = false;
    uint32_t head = 0, tail = 0; // ring positions (wave-uniform): [tail, head) are parked
    T tmin = (T)0, tmax = (T)0;
    uint32_t skip_off = 0;
    // current instance
    RayT<T> lray{zero3<T>(), zero3<T>()};
    RaySlab<T> lrs{};
    const BlasNode<T> *inner = nullptr;
    uint32_t D = 0, n_tris = 0, tri_base = 0, obj_id = 0;
    bool skip_here = false;
    // inside an instance: (i, c) the node to visit next while node_live; lf != 0: triangles to test first, (first << 3) | count
    // (types.hpp: BlasNode -- leaves arrive with their parent's record and are never deferred); pend: deferred inner siblings
    uint32_t i = 1, c = 0, lf = 0, lf2 = 0; // (lf2: the second child record's leaves in the two-level walk)
    bool node_live = false;
    // (one bit per level in the one-level walk: 32 bits hold every tree the builders make -- depth <= 31 --; the two-level walk
    // keeps four bits per two levels and needs 64)
    typedef typename std::conditional<WIDE, unsigned long long, uint32_t>::type pend_t;
    pend_t pend = 0;
    Isect<T> is;
    is.hit = false;
    is.t = (T)0;
    is.inscene_objid = is.inobject_offset = 0;
    is.bary = zero3<T>();
    const bool flat = sc.tlas_mode == 1u;
    // XCD-local order (SceneDev::stream_xcd): range r of 8 covers the items [r * range, (r + 1) * range) -- whole batches, a band
    // of the image -- and has its own counter; a wave starts with the range of its workgroup's XCD group (blockIdx % 8: the
    // dispatcher deals workgroups round-robin over the XCDs) and moves on to the next range when a grant finds one drained.
    // `ranges_left` only falls, every counter only grows: a wave is exhausted after at most 8 failed grants.
    const uint32_t n_ranges = sc.stream_xcd ? 8u : 1u;
    const uint32_t range = sc.stream_xcd ? (((n_items + 7u) / 8u + 63u) & ~63u) : n_items;
    uint32_t ranges_tried = 0;

    for (;;) {
        // ---- prepare: when the ring has room for a batch, all 64 lanes fetch and scan one item each ---------------------
        if (!exhausted && head - tail <= LPRT_STREAM_RING - 64u) {
            uint32_t base = 0, limit = 0;
            while (ranges_tried < n_ranges) {
                const uint32_t r = ((blockIdx.x & 7u) + ranges_tried) & (n_ranges - 1u);
                const uint32_t lo = r * range, len = lo < n_items ? (n_items - lo < range ? n_items - lo : range) : 0u;
                uint32_t b = len; // an empty range: nothing to ask for
                if (len != 0u) {
                    if (lane == 0u)
                        b = atomicAdd(stream_counter + 16u * r, 64u);
                    b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
                }
                if (b + 64u >= len)
                    ranges_tried++; // wave-uniform: nothing is left in this range behind this grant
                if (b < len) {
                    base = lo + b;
                    limit = lo + len;
                    break;
                }
            }
            if (ranges_tried >= n_ranges)
                exhausted = true;
            PrepRay<T> pr;
            RayT<T> nray{zero3<T>(), zero3<T>()};
            const uint32_t j = base + lane;
            pr.item = (so.order && j < limit) ? so.order[j] : j;
            const bool ok = j < limit && src.load(pr.item, nray, pr.tmin, pr.tmax, pr.skip_vao, pr.skip_off);
            if (ok) {
                if (COUNT)
                    cnt.rays++;
                const RaySlab<T> nrs = make_slab(nray, sc.axis_reject);
                pr.o = nray.o;
                pr.d = nray.d;
                pr.cand = 0;
                if (sc.n_objects != 0u) {
                    if (flat) {
                        // every world box against the ray (the boxes are wave-uniform addresses: scalar loads)
                        for (uint32_t k = 0; k < sc.n_objects; k++) {
                            const ObjRec<T> *ob = sc.objrecs + k;
                            float a;
                            if (COUNT)
                                cnt.aabbw++;
                            if (scene_box(ob->lo, ob->hi, nrs, (float)pr.tmin, (float)pr.tmax, scene_floor((float)pr.tmin), a))
                                pr.cand |= 1u << k;
                        }
                    } else {
                        float a;
                        if (COUNT)
                            cnt.tlas_nodes++;
                        pr.cand = scene_box(sc.tlas_root, sc.tlas_root + 3, nrs, (float)pr.tmin, (float)pr.tmax, scene_floor((float)pr.tmin), a) ? 1u : 0u;
                    }
                }
            }
            const unsigned long long okm = __builtin_amdgcn_ballot_w64(ok);
            if (ok)
                ring[(head + (uint32_t)__builtin_popcountll(okm & lt_mask)) & (LPRT_STREAM_RING - 1u)] = pr;
            head += (uint32_t)__builtin_popcountll(okm);
            // the ring is written by some lanes and read by others of the SAME wave: a wave barrier (no instruction, a
            // scheduling fence) keeps the compiler from moving the hand-out's LDS reads above these writes
            __builtin_amdgcn_wave_barrier();
        }
        // ---- hand out: idle lanes take parked rays ----------------------------------------------------------------------
        {
            const unsigned long long idle = __builtin_amdgcn_ballot_w64(!has_ray);
            const uint32_t avail = head - tail, n_idle = (uint32_t)__builtin_popcountll(idle);
            if (avail != 0u && (n_idle >= sc.stream_refill_min || n_idle == 64u)) {
                const uint32_t n_take = avail < n_idle ? avail : n_idle;
                const uint32_t rank = (uint32_t)__builtin_popcountll(idle & lt_mask);
                if (!has_ray && rank < n_take) {
                    const PrepRay<T> pr = ring[(tail + rank) & (LPRT_STREAM_RING - 1u)];
                    const RaySlab<T> prs = make_slab(RayT<T>{pr.o, pr.d}, sc.axis_reject); // the reciprocals again: cheaper than parking them
                    world.o = pr.o;
                    world.d = pr.d;
                    world.ix = prs.ix;
                    world.iy = prs.iy;
                    world.iz = prs.iz;
                    world.slow = prs.slow;
                    world.skip_vao = pr.skip_vao;
                    world.item = pr.item;
                    world.top = flat ? pr.cand : 1u;
                    world.tpend = 0u;
                    world.tcl = (!flat && pr.cand != 0u) ? (1u << 16) : 0u;
                    tmin = pr.tmin;
                    tmax = pr.tmax;
                    skip_off = pr.skip_off;
                    has_ray = true;
                    in_obj = false;
                    is.hit = false;
                    is.t = cvt<T>(1e5f); // +inf in half (cuda.hpp:54)
                    is.inscene_objid = is.inobject_offset = 0;
                    is.bary = zero3<T>();
                }
                tail += n_take;
                __builtin_amdgcn_wave_barrier(); // ring slots read above may be overwritten by the next prepare
            }
        }
        if (!__builtin_amdgcn_ballot_w64(has_ray)) {
            if (exhausted && head == tail)
                break;
            continue;
        }
        // ---- which phase runs: the one most lanes are waiting for ---------------------------------------------------------
        // A lane with a ray is in exactly one of three states: between instances (entry), on an inner node (node), on a leaf
        // (leaf).  Default (mode 0): all three phases every pass, the node phase repeated while >= stream_node_min lanes descend
        // ("while-while").  Mode 1 (LPRT_OPT_STREAM_SCHED) runs ONE phase per pass, the one with the most lanes: measured slower
        // on the atrium (47.3 vs 44.5 ms a frame) at almost unchanged lane occupancy (node 0.47 vs 0.45) -- what keeps the
        // occupancy at half is not lanes waiting for each other's phase but the stream's tail: a persistent wave gets ~500 rays
        // and its last ones run 200-400 steps with the lanes of the finished ones empty.
        const bool majority = sc.stream_sched != 0u;
        uint32_t n_entry = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(has_ray && !in_obj));
        uint32_t n_node = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in_obj && node_live && (lf | lf2) == 0u));
        uint32_t n_leaf = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(in_obj && (lf | lf2) != 0u));
        const bool run_node = !majority || (n_node >= n_entry && n_node >= n_leaf);
        const bool run_leaf = !majority || (!run_node && n_leaf >= n_entry);
        const bool run_entry = !majority || (!run_node && !run_leaf);
        // ---- top level: lanes between instances pick the next candidate, or retire their ray ---------------------------
        if (run_entry && has_ray && !in_obj) {
            cnt.step(0);
            uint32_t k = 0xffffffffu; // slot of the next instance to enter
            // the world-space ray and the top-level walk come out of the lane's LDS record and go back into it below
            RayT<T> wray{world.o, world.d};
            RaySlab<T> wrs{};
            wrs.o = wray.o;
#ifdef LPRT_REFSLAB_TU
            wrs.d = wray.d;
#endif
            wrs.ix = world.ix;
            wrs.iy = world.iy;
            wrs.iz = world.iz;
            wrs.slow = world.slow;
            wrs.reject = sc.axis_reject;
            wrs.par = sc.axis_reject ? wrs.slow : 0u;
            if (!(ANY && is.hit)) {
                if (flat) {
                    uint32_t cand = world.top;
                    if (cand) {
                        k = (uint32_t)__builtin_ctz(cand);
                        cand &= cand - 1u;
                        world.top = cand;
                    }
                } else {
                    uint32_t ti = world.top, tpend = world.tpend, tc = world.tcl & 255u, tleaf = (world.tcl >> 8) & 255u;
                    bool tlas_live = (world.tcl >> 16) != 0u;
                    // resume the TLAS walk (cast_ray's tree loop, trace.hpp) until it yields an instance whose world box passes
                    while (tlas_live && k == 0xffffffffu) {
                        if (tc < sc.tlas_depth) {
                            const InnerNode<T> nd = sc.tlas_inner[ti - 1];
                            if (COUNT)
                                cnt.tlas_nodes += 2;
                            float n0, n1;
                            const bool h0 = scene_box(nd.lo0, nd.hi0, wrs, (float)tmin, (float)tmax, scene_floor((float)tmin), n0);
                            const bool h1 = scene_box(nd.lo1, nd.hi1, wrs, (float)tmin, (float)tmax, scene_floor((float)tmin), n1);
                            if (h0 | h1) {
                                const uint32_t first = (h0 && (!h1 || n0 <= n1)) ? 0u : 1u;
                                if (h0 & h1)
                                    tpend |= 1u << (tc + 1);
                                ti = 2 * ti + first;
                                tc++;
                                tleaf = 0;
                                continue;
                            }
                        } else {
                            const uint32_t lw = sc.tlas_leaf[ti - (1u << sc.tlas_depth)];
                            const uint32_t first = lw >> 1, count = (lw & 1u) + 1u;
                            if (tleaf < count) {
                                const uint32_t s = first + tleaf;
                                tleaf++;
                                const ObjRec<T> *ob = sc.objrecs + s;
                                float a;
                                if (COUNT)
                                    cnt.aabbw++;
                                if (scene_box(ob->lo, ob->hi, wrs, (float)tmin, (float)tmax, scene_floor((float)tmin), a))
                                    k = s;
                                continue;
                            }
                        }
                        if (!tpend) {
                            tlas_live = false;
                            break;
                        }
                        const uint32_t d = 31u - (uint32_t)__builtin_clz(tpend);
                        tpend ^= 1u << d;
                        ti = (ti >> (tc - d)) ^ 1u;
                        tc = d;
                        tleaf = 0;
                    }
                    world.top = ti;
                    world.tpend = tpend;
                    world.tcl = tc | (tleaf << 8) | (tlas_live ? (1u << 16) : 0u);
                }
            }
            if (k == 0xffffffffu) {
                src.store(world.item, is); // the ray is done
                has_ray = false;
            } else {
                const ObjHead<T> ob = load_obj_head(sc.objrecs + k); // one round trip; the matrix only when it is one
                if (COUNT)
                    cnt.objects++;
                if (ob.n_tris != 0u) {
                    if (ob.pad[0] & 1u) { // identity W2L (make_objrec): the instance's ray is the world ray
                        lray = wray;
                        lrs = wrs;
                    } else {
                        lray = transform_ray(wray, sc.objrecs[k].W2L);
                        lrs = make_slab(lray, sc.axis_reject);
                    }
                    float tn;
                    if (COUNT)
                        cnt.blas_nodes++;
                    if (stream_box(ob.root_lo, ob.root_hi, lrs, (float)is.t, (float)tmin, (float)tmax, tn, __builtin_amdgcn_ballot_w64(lrs.slow != 0u) != 0ull)) {
                        inner = sc.blas_inner + ob.inner_base;
                        D = ob.depth;
                        n_tris = ob.n_tris;
                        tri_base = ob.tri_base;
                        obj_id = ob.inscene_objid;
                        skip_here = ob.vao_index == world.skip_vao;
                        i = 1;
                        c = 0;
                        pend = 0;
                        lf = lf2 = 0;
                        node_live = true;
                        in_obj = true;
                        if (D <= 1u) { // 1-3 triangles: no boxes worth testing (trace.hpp)
                            for (uint32_t q = 0; q < n_tris; q++) {
                                const uint32_t rec = tri_base + q;
                                tri_test<T, COUNT>(sc, rec, lray, obj_id, is, tmin, tmax, cnt, skip_here, skip_off);
                                if (ANY && is.hit)
                                    break;
                            }
                            in_obj = false;
                        }
                    }
                }
            }
        }
        // ---- node steps: "while-while", left early when few lanes are still descending ----------------------------------
        // deferred sibling -> (i, c); false: nothing is pending
        auto pop = [&]() -> bool {
            if (!pend)
                return false;
            if (WIDE) {
                const uint32_t bit = 63u - (uint32_t)__builtin_clzll((unsigned long long)pend);
                pend ^= (pend_t)(1ull << bit);
                const uint32_t k2 = (bit >> 2) * 2u, j = bit & 3u; // depth of the node the step started from, grandchild slot
                i = ((i >> (c - k2)) << 2) | j;
                c = k2 + 2u;
            } else {
                const uint32_t d = 31u - (uint32_t)__builtin_clz((uint32_t)pend);
                pend ^= (pend_t)1u << d;
                i = (i >> (c - d)) ^ 1u;
                c = d;
            }
            return true;
        };
        // (a lane changes its ray only in the entry phase: whether some lane of the wave holds a ray with a stationary axis is
        // invariant across the node loop)
        const bool wave_slow = __builtin_amdgcn_ballot_w64(in_obj && lrs.slow != 0u) != 0ull;
        while (run_node && in_obj && node_live && (lf | lf2) == 0u) {
            cnt.step(1);
            if (WIDE && c + 2u <= D) {
                // two levels per step (trace.hpp): the records of both children, whatever the children are
                const BlasNode<T> ra = inner[2 * i - 1], rb = inner[2 * i];
                if (COUNT)
                    cnt.blas_nodes += 4;
                float n[4];
                const float bt = (float)is.t, mn = (float)tmin, mx = (float)tmax;
                const bool h0 = stream_box(ra.lo0, ra.hi0, lrs, bt, mn, mx, n[0], wave_slow), h1 = stream_box(ra.lo1, ra.hi1, lrs, bt, mn, mx, n[1], wave_slow);
                const bool h2 = stream_box(rb.lo0, rb.hi0, lrs, bt, mn, mx, n[2], wave_slow), h3 = stream_box(rb.lo1, rb.hi1, lrs, bt, mn, mx, n[3], wave_slow);
                const bool l0 = ra.leaf[0] != LPRT_CHILD_INNER, l1 = ra.leaf[1] != LPRT_CHILD_INNER;
                const bool l2 = rb.leaf[0] != LPRT_CHILD_INNER, l3 = rb.leaf[1] != LPRT_CHILD_INNER;
                if (h0 && l0)
                    lf = ra.leaf[0];
                if (h1 && l1)
                    lf = lf ? lf + (ra.leaf[1] & 7u) : ra.leaf[1];
                if (h2 && l2)
                    lf2 = rb.leaf[0];
                if (h3 && l3)
                    lf2 = lf2 ? lf2 + (rb.leaf[1] & 7u) : rb.leaf[1];
                const uint32_t g = ((h0 && !l0) ? 1u : 0u) | ((h1 && !l1) ? 2u : 0u) | ((h2 && !l2) ? 4u : 0u) | ((h3 && !l3) ? 8u : 0u);
                if (g) {
                    uint32_t first = 0;
                    float best = __builtin_inff();
                    bool have = false;
#pragma unroll
                    for (uint32_t q = 0; q < 4; q++)
                        if (((g >> q) & 1u) && (!have || n[q] < best)) {
                            best = n[q];
                            first = q;
                            have = true;
                        }
                    pend |= (pend_t)((unsigned long long)(g & ~(1u << first)) << (2 * c));
                    i = 4 * i + first;
                    c += 2;
                } else {
                    node_live = pop();
                }
            } else {
                const BlasNode<T> nd = inner[i - 1];
                if (COUNT)
                    cnt.blas_nodes += 2;
                float n0, n1;
                const float bt = (float)is.t, mn = (float)tmin, mx = (float)tmax;
                const bool h0 = stream_box(nd.lo0, nd.hi0, lrs, bt, mn, mx, n0, wave_slow);
                const bool h1 = stream_box(nd.lo1, nd.hi1, lrs, bt, mn, mx, n1, wave_slow);
                const bool l0 = nd.leaf[0] != LPRT_CHILD_INNER, l1 = nd.leaf[1] != LPRT_CHILD_INNER;
                if (h0 && l0)
                    lf = nd.leaf[0];
                if (h1 && l1)
                    lf = lf ? lf + (nd.leaf[1] & 7u) : nd.leaf[1]; // sibling leaves are adjacent in leaf order
                const bool g0 = h0 && !l0 && !WIDE, g1 = h1 && !l1 && !WIDE; // (two-level walk: this is the last level, all leaves)
                if (g0 | g1) {
                    const uint32_t first = (g0 && (!g1 || n0 <= n1)) ? 0u : 1u;
                    if (g0 & g1)
                        pend |= (pend_t)1u << (c + 1);
                    i = 2 * i + first;
                    c += 1;
                } else {
                    node_live = pop();
                }
            }
            if (majority) {
                // stay while the descending lanes are still the largest group (lanes that got triangles to test or left their
                // instance during this step are counted where they now are)
                const uint32_t now_node = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(node_live && (lf | lf2) == 0u));
                const uint32_t moved = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) - now_node;
                n_leaf += moved; // upper bound for either of the other groups: cheap, and errs towards leaving early
                if (now_node < n_leaf || now_node < n_entry + moved)
                    break;
            } else if ((uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(true)) < sc.stream_node_min) {
                break;
            }
        }
        // ---- leaf: the wave tests triangles -----------------------------------------------------------------------------
        if (run_leaf && in_obj && (lf | lf2) != 0u) {
            cnt.step(2);
            if (lf == 0u) { // (only the two-level walk fills lf2)
                lf = lf2;
                lf2 = 0;
            }
            const uint32_t l = lf >> 3, r = l + (lf & 7u);
            lf = WIDE ? lf2 : 0u; // the second range, if any, in the next pass
            lf2 = 0;
            for (uint32_t q = l; q < r; q++) {
                const uint32_t rec = tri_base + q;
                tri_test<T, COUNT>(sc, rec, lray, obj_id, is, tmin, tmax, cnt, skip_here, skip_off);
                if (ANY && is.hit)
                    break;
            }
            if (ANY && is.hit) {
                node_live = false;
                lf = 0;
            }
        }
        if (in_obj && !node_live && (lf | lf2) == 0u)
            in_obj = false; // this instance is done: back to the top level
    }
}

} // namespace lprt
