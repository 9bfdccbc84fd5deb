// CPU experiment (development tool, not part of the product or of the oracle): how many box tests, node steps and triangle
// tests does a ray of the atrium cost under different BLAS builders?  Emulates the GPU walk (csrc/trace.hpp: two child boxes
// per node record, near child first, deferred siblings popped deepest first, the cull by the running closest t) over
//   median   the round-2 builder: widest axis of the node, object median of the centroids, perfect tree, 1-2 triangles per leaf
//   sah      binned surface-area heuristic with the builder's constraints: depth <= Dmax, leaves of <= L triangles
// and prints the averages for primary rays, one-bounce GI rays and shadow rays of the atrium camera.
//   python tools/export_scene_bin.py /tmp/scene.bin && g++ -O2 -fopenmp tools/bvh_quality.cpp -o /tmp/bq && /tmp/bq /tmp/scene.bin
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

struct V3 {
    float x, y, z;
    float operator[](int k) const { return k == 0 ? x : k == 1 ? y : z; }
};
static V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V3 operator*(V3 a, float k) { return {a.x * k, a.y * k, a.z * k}; }
static float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V3 norm(V3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

struct Box {
    float lo[3] = {1e30f, 1e30f, 1e30f}, hi[3] = {-1e30f, -1e30f, -1e30f};
    void add(V3 p) {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], p[k]);
            hi[k] = std::max(hi[k], p[k]);
        }
    }
    void add(const Box &b) {
        for (int k = 0; k < 3; k++) {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    float area() const {
        float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (d[0] < 0)
            return 0;
        return 2 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
};
struct Tri {
    V3 v[3];
    Box box() const {
        Box b;
        for (auto &p : v)
            b.add(p);
        return b;
    }
    V3 centroid() const {
        Box b = box();
        return {(b.lo[0] + b.hi[0]) * 0.5f, (b.lo[1] + b.hi[1]) * 0.5f, (b.lo[2] + b.hi[2]) * 0.5f};
    }
};
struct Node {
    Box cb[2];
    int child[2];     // >= 0: inner node index; < 0: leaf
    int first[2], cnt[2];
};
struct Blas {
    std::vector<Tri> tris; // leaf order
    std::vector<Node> nodes;
    Box root;
    int root_leaf_cnt = -1; // >= 0: the whole primitive is one leaf
    int max_depth = 0;
    double sah = 0;
};

static int g_variant = 0; // bit 0: longest centroid axis only; bit 1: bins over the node's box instead of the centroid bounds
struct Params {
    int mode;  // 0 median, 1 sah
    int L;     // max triangles per leaf (sah)
    int extra; // Dmax = floor(log2(n / L)) + 1 + extra  (sah)
    int bins;
};

static int build_rec(Blas &b, std::vector<Tri> &t, int l, int r, int d, int D, const Params &pm, Box &out_box, int &out_first, int &out_cnt) {
    Box nb;
    for (int i = l; i < r; i++)
        nb.add(t[i].box());
    out_box = nb;
    b.max_depth = std::max(b.max_depth, d);
    const int n = r - l;
    bool leaf = pm.mode != 1 ? d == D : n <= pm.L;
    if (leaf) {
        out_first = l;
        out_cnt = n;
        return -1;
    }
    int split = l + n / 2, axis = 0;
    if (pm.mode == 0) {
        float w[3] = {nb.hi[0] - nb.lo[0], nb.hi[1] - nb.lo[1], nb.hi[2] - nb.lo[2]};
        axis = (w[1] > w[0] && w[1] > w[2]) ? 1 : (w[2] > w[0] && w[2] > w[1]) ? 2 : 0;
        std::stable_sort(t.begin() + l, t.begin() + r, [&](const Tri &a, const Tri &c) { return a.centroid()[axis] < c.centroid()[axis]; });
    } else {
        // capacity of a child subtree: L * 2^(levels below the child); mode 2 (perfect tree): every leaf needs at least one
        const long long cap = (long long)(pm.mode == 2 ? 2 : pm.L) << std::max(0, D - d - 1);
        const long long need = pm.mode == 2 ? (1ll << std::max(0, D - d - 1)) : 1;
        Box cbx;
        for (int i = l; i < r; i++)
            cbx.add(t[i].centroid());
        double best = 1e300;
        int best_axis = -1, best_bin = -1;
        const int NB = pm.bins;
        int longest = 0;
        {
            float w[3] = {cbx.hi[0] - cbx.lo[0], cbx.hi[1] - cbx.lo[1], cbx.hi[2] - cbx.lo[2]};
            longest = (w[1] > w[0] && w[1] > w[2]) ? 1 : (w[2] > w[0] && w[2] > w[1]) ? 2 : 0;
        }
        if (g_variant & 2)
            cbx = nb;
        for (int ax = 0; ax < 3; ax++) {
            if ((g_variant & 1) && ax != longest)
                continue;
            const float lo = cbx.lo[ax], ext = cbx.hi[ax] - lo;
            if (!(ext > 0))
                continue;
            std::vector<Box> bb(NB);
            std::vector<int> bc(NB, 0);
            for (int i = l; i < r; i++) {
                int k = std::min(NB - 1, (int)((t[i].centroid()[ax] - lo) / ext * NB));
                bb[k].add(t[i].box());
                bc[k]++;
            }
            std::vector<Box> suf(NB);
            std::vector<int> sufc(NB, 0);
            Box acc;
            int c = 0;
            for (int k = NB - 1; k >= 0; k--) {
                acc.add(bb[k]);
                c += bc[k];
                suf[k] = acc;
                sufc[k] = c;
            }
            Box pre;
            int pc = 0;
            for (int k = 0; k + 1 < NB; k++) {
                pre.add(bb[k]);
                pc += bc[k];
                const int rc = sufc[k + 1];
                if (pc < need || rc < need || pc > cap || rc > cap)
                    continue;
                const double cost = pre.area() * (double)pc + suf[k + 1].area() * (double)rc;
                if (cost < best) {
                    best = cost;
                    best_axis = ax;
                    best_bin = k;
                }
            }
        }
        if (best_axis >= 0) {
            axis = best_axis;
            const float lo = cbx.lo[axis], ext = cbx.hi[axis] - lo;
            auto mid = std::stable_partition(t.begin() + l, t.begin() + r, [&](const Tri &a) {
                return std::min(NB - 1, (int)((a.centroid()[axis] - lo) / ext * NB)) <= best_bin;
            });
            split = (int)(mid - t.begin());
        } else { // no admissible plane (all centroids equal, or the capacity excludes every plane): object median
            float w[3] = {nb.hi[0] - nb.lo[0], nb.hi[1] - nb.lo[1], nb.hi[2] - nb.lo[2]};
            axis = (w[1] > w[0] && w[1] > w[2]) ? 1 : (w[2] > w[0] && w[2] > w[1]) ? 2 : 0;
            std::stable_sort(t.begin() + l, t.begin() + r, [&](const Tri &a, const Tri &c) { return a.centroid()[axis] < c.centroid()[axis]; });
            split = l + n / 2;
        }
    }
    const int me = (int)b.nodes.size();
    b.nodes.push_back({});
    Node nd;
    nd.child[0] = build_rec(b, t, l, split, d + 1, D, pm, nd.cb[0], nd.first[0], nd.cnt[0]);
    nd.child[1] = build_rec(b, t, split, r, d + 1, D, pm, nd.cb[1], nd.first[1], nd.cnt[1]);
    b.sah += nd.cb[0].area() + nd.cb[1].area();
    b.nodes[me] = nd;
    return me;
}

static Blas build(const std::vector<Tri> &src, const Params &pm) {
    Blas b;
    b.tris = src;
    const int n = (int)src.size();
    int D = 0;
    if (pm.mode != 1) {
        while ((2 << D) <= n)
            D++; // floor(log2 n)
    } else {
        int need = (n + pm.L - 1) / pm.L, lg = 0;
        while ((1 << lg) < need)
            lg++;
        D = lg + pm.extra;
    }
    int first, cnt;
    int r = build_rec(b, b.tris, 0, n, 0, D, pm, b.root, first, cnt);
    if (r < 0)
        b.root_leaf_cnt = cnt;
    return b;
}

struct Stats {
    double rays = 0, boxes = 0, steps = 0, tris = 0, inst = 0, leafs = 0;
    void add(const Stats &o) {
        rays += o.rays, boxes += o.boxes, steps += o.steps, tris += o.tris, inst += o.inst, leafs += o.leafs;
    }
};
struct Hit {
    float t = 1e30f;
    int prim = -1, tri = -1;
};
static bool slab(const Box &b, V3 o, const float inv[3], float best, float mn, float mx, float &tn) {
    float t1 = -1e30f, t2 = 1e30f;
    for (int k = 0; k < 3; k++) {
        float a = (b.lo[k] - o[k]) * inv[k], c = (b.hi[k] - o[k]) * inv[k];
        t1 = std::max(t1, std::min(a, c));
        t2 = std::min(t2, std::max(a, c));
    }
    tn = t1;
    return t1 <= t2 * 1.001953f && t2 >= 0 && t1 < best && t1 < mx && t2 > mn;
}
static bool tri_hit(const Tri &tr, V3 o, V3 d, float mn, float mx, float &t) {
    V3 e1 = tr.v[1] - tr.v[0], e2 = tr.v[2] - tr.v[0], p = cross(d, e2);
    float det = dot(e1, p);
    if (std::fabs(det) < 1e-12f)
        return false;
    float inv = 1 / det;
    V3 s = o - tr.v[0];
    float u = dot(s, p) * inv;
    if (u < 0 || u > 1)
        return false;
    V3 q = cross(s, e1);
    float v = dot(d, q) * inv;
    if (v < 0 || u + v > 1)
        return false;
    float tt = dot(e2, q) * inv;
    if (tt > mn && tt < mx) {
        t = tt;
        return true;
    }
    return false;
}
static void walk(const Blas &b, int prim, V3 o, V3 d, float mn, float mx, bool any, Hit &h, Stats &st) {
    float inv[3] = {1 / d.x, 1 / d.y, 1 / d.z}, tn;
    if (!slab(b.root, o, inv, h.t, mn, mx, tn)) // (the instance scan: not counted as a BLAS box test)
        return;
    st.inst++;
    st.boxes++;
    auto leaf = [&](int first, int cnt) {
        st.leafs++;
        for (int i = first; i < first + cnt; i++) {
            float t;
            st.tris++;
            if (tri_hit(b.tris[i], o, d, mn, std::min(mx, h.t), t)) {
                h.t = t;
                h.prim = prim;
                h.tri = i;
                if (any)
                    return true;
            }
        }
        return false;
    };
    if (b.root_leaf_cnt >= 0) {
        leaf(0, b.root_leaf_cnt);
        return;
    }
    int stack[128], sp = 0, cur = 0;
    for (;;) {
        const Node &nd = b.nodes[cur];
        st.steps++;
        st.boxes += 2;
        float n0, n1;
        bool h0 = slab(nd.cb[0], o, inv, h.t, mn, mx, n0), h1 = slab(nd.cb[1], o, inv, h.t, mn, mx, n1);
        int order[2] = {0, 1};
        if (h0 && h1 && n1 < n0)
            order[0] = 1, order[1] = 0;
        int next = -1;
        // the far child is deferred; leaves are entered at once in near-far order, as the GPU walk does when it reaches them
        bool hh[2] = {h0, h1};
        int deferred = -1;
        for (int q = 0; q < 2; q++) {
            int c = order[q];
            if (!hh[c])
                continue;
            if (next < 0 && deferred < 0) {
                next = c;
            } else {
                deferred = c;
            }
        }
        // process: near first
        int go = -1;
        if (next >= 0) {
            if (deferred >= 0)
                stack[sp++] = (cur << 1) | deferred; // revisit: the box is re-tested against the then-current best t on pop
            if (nd.child[next] < 0) {
                if (leaf(nd.first[next], nd.cnt[next]) && any)
                    return;
            } else {
                go = nd.child[next];
            }
        }
        while (go < 0) {
            if (!sp)
                return;
            int e = stack[--sp], pn = e >> 1, c = e & 1;
            const Node &p = b.nodes[pn];
            // (the GPU does not re-test the deferred box; it re-tests its children.  A popped LEAF child is tested as is.)
            if (p.child[c] < 0) {
                if (leaf(p.first[c], p.cnt[c]) && any)
                    return;
            } else {
                go = p.child[c];
            }
        }
        cur = go;
    }
}

int main(int argc, char **argv) {
    if (argc < 2)
        return 1;
    FILE *f = fopen(argv[1], "rb");
    uint32_t hdr[3];
    if (!f || fread(hdr, 4, 3, f) != 3)
        return 1;
    std::vector<float> pos(hdr[0] * 3), nrm(hdr[0] * 3);
    std::vector<uint32_t> idx(hdr[1]), vaos(hdr[2] * 4);
    if (fread(pos.data(), 4, pos.size(), f) != pos.size() || fread(nrm.data(), 4, nrm.size(), f) != nrm.size() ||
        fread(idx.data(), 4, idx.size(), f) != idx.size() || fread(vaos.data(), 4, vaos.size(), f) != vaos.size())
        return 1;
    fclose(f);
    const int np = (int)hdr[2];
    std::vector<std::vector<Tri>> prims(np);
    for (int p = 0; p < np; p++) {
        uint32_t off = vaos[4 * p + 2], len = vaos[4 * p + 3];
        for (uint32_t i = 0; i + 2 < len; i += 3) {
            Tri t;
            for (int k = 0; k < 3; k++) {
                uint32_t v = idx[off + i + k];
                t.v[k] = {pos[3 * v], pos[3 * v + 1], pos[3 * v + 2]};
            }
            prims[p].push_back(t);
        }
    }
    if (argc > 2)
        g_variant = atoi(argv[2]);
    if (argc > 3 && atoi(argv[3]) == 1) { // one BLAS over all primitives (identity instances merged)
        std::vector<Tri> all;
        for (auto &pr : prims)
            all.insert(all.end(), pr.begin(), pr.end());
        prims.assign(1, all);
    }
    const int np2 = (int)prims.size();
    std::vector<Params> cfgs = {{0, 2, 0, 0}, {1, 2, 1, 16}};
    const int W = 256, H = 144;
    for (const Params &pm : cfgs) {
        std::vector<Blas> bl(np2);
#pragma omp parallel for schedule(dynamic, 1)
        for (int p = 0; p < np2; p++)
            bl[p] = build(prims[p], pm);
        int maxd = 0;
        double nodes = 0;
        for (auto &b : bl) {
            maxd = std::max(maxd, b.max_depth);
            nodes += b.nodes.size();
        }
        auto cast = [&](V3 o, V3 d, float mn, float mx, bool any, Stats &st) {
            Hit h;
            st.rays++;
            for (int p = 0; p < np2; p++) {
                walk(bl[p], p, o, d, mn, mx, any, h, st);
                if (any && h.prim >= 0)
                    break;
            }
            return h;
        };
        Stats sp, sg, ss;
        std::vector<Stats> tp(W * H);
        // camera at (-11, 2, 0) looking +x, yfov 1.0
#pragma omp parallel for schedule(dynamic, 64)
        for (int i = 0; i < W * H; i++) {
            Stats a, g, s;
            std::mt19937 rng(i * 7919 + 13);
            std::uniform_real_distribution<float> U(0, 1);
            int x = i % W, y = i / W;
            float my = std::tan(0.5f), ny = (2 * (y + 0.5f) / H - 1) * my, nx = (2 * (x + 0.5f) / W - 1) * my * W / H;
            V3 o{-11, 2, 0}, d = norm(V3{1, ny, nx});
            Hit h = cast(o, d, 0, 1e30f, false, a);
            if (h.prim >= 0) {
                const Tri &tr = bl[h.prim].tris[h.tri];
                V3 n = norm(cross(tr.v[1] - tr.v[0], tr.v[2] - tr.v[0]));
                if (dot(n, d) > 0)
                    n = n * -1.0f;
                V3 p = o + d * h.t;
                // uniform direction on the hemisphere (normalised Gaussians, flipped: SURVEY.md Q10)
                std::normal_distribution<float> N(0, 1);
                V3 gd = norm(V3{N(rng), N(rng), N(rng)});
                if (dot(gd, n) < 0)
                    gd = gd * -1.0f;
                Hit h2 = cast(p + n * 1e-3f, gd, 1e-4f, 1e30f, false, g);
                if (h2.prim >= 0) {
                    V3 p2 = p + n * 1e-3f + gd * h2.t;
                    V3 L{7.0f, 8.5f, 1.0f};
                    V3 ld = L - p2;
                    float dist = std::sqrt(dot(ld, ld));
                    cast(p2 + gd * -1e-3f, ld * (1 / dist), 1e-3f, dist, true, s);
                }
            }
#pragma omp critical
            {
                sp.add(a);
                sg.add(g);
                ss.add(s);
            }
        }
        printf("mode=%d L=%d extra=%d bins=%d | max depth %d, %.0f nodes\n", pm.mode, pm.L, pm.extra, pm.bins, maxd, nodes);
        for (auto pr : {std::make_pair("primary", &sp), std::make_pair("gi", &sg), std::make_pair("shadow(any)", &ss)}) {
            Stats &s = *pr.second;
            printf("  %-12s rays %8.0f  boxes/ray %6.1f  steps/ray %6.1f  tris/ray %5.1f  leaves/ray %4.1f  instances/ray %4.1f\n", pr.first, s.rays,
                   s.boxes / s.rays, s.steps / s.rays, s.tris / s.rays, s.leafs / s.rays, s.inst / s.rays);
        }
        fflush(stdout);
    }
    return 0;
}
