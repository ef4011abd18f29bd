// bvh_build.cpp -- the traversal structure the fast kernels use on the GPU.
//
// The reference's tree (src/KDTree.hs:79-90: spatial-median split, leaves of up
// to 19 triangles, fp64 boxes, both children always visited) is kept verbatim
// for the exhaustive/validation kernels.  It is a poor fit for a 64-lane
// machine: large leaves mean many fp64 triangle tests, and 64-byte fp64 boxes
// double the gather traffic.  Its boxes only decide WHICH triangles get tested;
// the result of a query is "closest triangle hit, ties by reference order", so
// any conservative culling structure yields the same answer.  This file builds
// one that suits CDNA4:
//   * binary BVH, binned surface-area heuristic, leaves of at most 3
//     triangles (triangle tests are the expensive fp64 part);
//   * a node stores BOTH children's boxes as fp32, rounded outward and inflated
//     by 2^-18 of the mesh extent, 64 bytes per node = one record per visit;
//   * triangles are re-ordered into BVH leaf order; each keeps its reference
//     rank so that exact-t ties resolve as in the reference
//     (src/Geometry.hs:54-57, src/KDTree.hs:113-115).
#include "bvh_build.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <limits>
#include <string>
#include <vector>

namespace rayhs {

namespace {

struct Box3 {
    double lo[3], hi[3];
    void reset() { for (int k = 0; k < 3; k++) { lo[k] = std::numeric_limits<double>::infinity(); hi[k] = -lo[k]; } }
    void grow(const double p[3]) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
    void grow(const Box3 &b) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    double half_area() const {
        double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0;
        return dx * dy + dy * dz + dz * dx;
    }
};

constexpr int kBins = 64;  // 16 bins: 1.2 % more box tests on the dragon frame
// Leaves of at most 3 triangles (2..6 measure within 2 % of each other, profiles/r04/experiments/bvh_leaf_sweep.txt); SAH splits
// only while depth + ceil(log2 n) stays below the depth limit, median splits (which halve n) otherwise, so inner nodes never sit
// deeper than that.  Builds made with -DRAYHS_DEV_KNOBS (build_exp/) can move both: RAYHS_BVH_LEAF=1..7, RAYHS_BVH_DEPTH_LIMIT.
#ifdef RAYHS_DEV_KNOBS
static int max_leaf() {
    static int v = [] { const char *e = std::getenv("RAYHS_BVH_LEAF"); int n = e ? std::atoi(e) : 3; return n < 1 ? 1 : n > 7 ? 7 : n; }();
    return v;
}
static int depth_limit() {
    static int v = [] { const char *e = std::getenv("RAYHS_BVH_DEPTH_LIMIT"); int n = e ? std::atoi(e) : 28; return n < 4 ? 4 : n > 28 ? 28 : n; }();
    return v;
}
#define kMaxLeaf (max_leaf())
#define kDepthLimit (depth_limit())
#else
constexpr int kMaxLeaf = 3, kDepthLimit = 28;
#endif

struct Builder {
    const std::vector<Box3> &tb;       // per-triangle bounds
    const std::vector<double> &cen;    // per-triangle centroid, 3 each
    std::vector<int> order;            // triangle permutation being partitioned in place
    std::vector<FNodeBuild> nodes;
    int max_depth = 0;

    Builder(const std::vector<Box3> &b, const std::vector<double> &c) : tb(b), cen(c) {}

    Box3 bounds(int a, int b) const {
        Box3 r; r.reset();
        for (int i = a; i < b; i++) r.grow(tb[(size_t)order[(size_t)i]]);
        return r;
    }

    // returns child reference: >= 0 inner node, < 0 leaf ~((first << 3) | count)
    int build(int a, int b, int depth) {
        max_depth = std::max(max_depth, depth);
        const int n = b - a;
        if (n <= kMaxLeaf) return ~((a << 3) | n);
        // centroid bounds
        double clo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, chi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
        for (int i = a; i < b; i++)
            for (int k = 0; k < 3; k++) {
                double c = cen[3 * (size_t)order[(size_t)i] + (size_t)k];
                clo[k] = std::min(clo[k], c); chi[k] = std::max(chi[k], c);
            }
        int mid = -1;
        int lg = 0;
        while ((1 << lg) < n) lg++;
        if (depth + lg + 1 < kDepthLimit) {
            double best = DBL_MAX; int best_axis = -1, best_bin = -1;
            for (int ax = 0; ax < 3; ax++) {
                double ext = chi[ax] - clo[ax];
                if (!(ext > 0)) continue;
                Box3 bb[kBins]; int cnt[kBins];
                for (int k = 0; k < kBins; k++) { bb[k].reset(); cnt[k] = 0; }
                double scale = kBins / ext;
                for (int i = a; i < b; i++) {
                    int t = order[(size_t)i];
                    int k = std::min(kBins - 1, std::max(0, (int)((cen[3 * (size_t)t + (size_t)ax] - clo[ax]) * scale)));
                    bb[k].grow(tb[(size_t)t]); cnt[k]++;
                }
                double right_area[kBins]; int right_cnt[kBins];
                Box3 acc; acc.reset(); int c = 0;
                for (int k = kBins - 1; k > 0; k--) { acc.grow(bb[k]); c += cnt[k]; right_area[k] = acc.half_area(); right_cnt[k] = c; }
                acc.reset(); c = 0;
                for (int k = 0; k < kBins - 1; k++) {
                    acc.grow(bb[k]); c += cnt[k];
                    if (c == 0 || right_cnt[k + 1] == 0) continue;
                    double cost = acc.half_area() * c + right_area[k + 1] * right_cnt[k + 1];
                    if (cost < best) { best = cost; best_axis = ax; best_bin = k; }
                }
            }
            if (best_axis >= 0) {
                double ext = chi[best_axis] - clo[best_axis], scale = kBins / ext;
                auto it = std::partition(order.begin() + a, order.begin() + b, [&](int t) {
                    int k = std::min(kBins - 1, std::max(0, (int)((cen[3 * (size_t)t + (size_t)best_axis] - clo[best_axis]) * scale)));
                    return k <= best_bin;
                });
                mid = (int)(it - order.begin());
            }
        }
        if (mid <= a || mid >= b) {  // no useful SAH split: median along the widest centroid axis
            int ax = 0;
            for (int k = 1; k < 3; k++) if (chi[k] - clo[k] > chi[ax] - clo[ax]) ax = k;
            mid = a + n / 2;
            std::nth_element(order.begin() + a, order.begin() + mid, order.begin() + b, [&](int x, int y) {
                return cen[3 * (size_t)x + (size_t)ax] < cen[3 * (size_t)y + (size_t)ax];
            });
        }
        int id = (int)nodes.size();
        nodes.emplace_back();
        Box3 b0 = bounds(a, mid), b1 = bounds(mid, b);
        int c0 = build(a, mid, depth + 1);
        int c1 = build(mid, b, depth + 1);
        FNodeBuild &nd = nodes[(size_t)id];
        std::memcpy(nd.lo0, b0.lo, sizeof nd.lo0); std::memcpy(nd.hi0, b0.hi, sizeof nd.hi0);
        std::memcpy(nd.lo1, b1.lo, sizeof nd.lo1); std::memcpy(nd.hi1, b1.hi, sizeof nd.hi1);
        nd.c0 = c0; nd.c1 = c1;
        return id;
    }
};

}  // namespace

float round_down_f32(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafter(f, -std::numeric_limits<float>::infinity());
    return f;
}
float round_up_f32(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafter(f, std::numeric_limits<float>::infinity());
    return f;
}

void build_fast_bvh(const double *tri_pos, const std::vector<int> &tri_ids, FastBvh &out) {
    const int count = (int)tri_ids.size();
    out.nodes.clear(); out.order.clear(); out.depth = 0;
    Box3 all; all.reset();
    std::vector<Box3> tb((size_t)count);
    std::vector<double> cen(3 * (size_t)count);
    for (int i = 0; i < count; i++) {
        const double *p = tri_pos + 9 * (size_t)tri_ids[(size_t)i];
        Box3 &b = tb[(size_t)i]; b.reset();
        for (int v = 0; v < 3; v++) b.grow(p + 3 * v);
        for (int k = 0; k < 3; k++) {
            // a NaN vertex (the reference can produce them, e.g. rotate about the z axis) must not
            // poison the tree: give the triangle an empty box.  The kernels never report a hit on
            // such a triangle, whereas the reference does return NaN hits for it -- the documented
            // deviation of DESIGN.md ("Known deviation"), pinned by tests/test_gpu_parity.py
            if (!(b.lo[k] <= b.hi[k])) { b.lo[k] = 0; b.hi[k] = 0; }
            cen[3 * (size_t)i + (size_t)k] = 0.5 * (b.lo[k] + b.hi[k]);
        }
        all.grow(b);
    }
    for (int k = 0; k < 3; k++) { out.lo[k] = count ? all.lo[k] : 0; out.hi[k] = count ? all.hi[k] : 0; }
    Builder bl(tb, cen);
    bl.order.resize((size_t)count);
    for (int i = 0; i < count; i++) bl.order[(size_t)i] = i;
    int root;
    if (count == 0) root = INT32_MIN;
    else if (count <= kMaxLeaf) {
        // a single leaf still needs a node so that the kernel has two child slots to test
        bl.nodes.emplace_back();
        FNodeBuild &nd = bl.nodes[0];
        std::memcpy(nd.lo0, all.lo, sizeof nd.lo0); std::memcpy(nd.hi0, all.hi, sizeof nd.hi0);
        // ... and the second slot is an EMPTY leaf (no triangles) with the same box: every stored box stays finite
        std::memcpy(nd.lo1, all.lo, sizeof nd.lo1); std::memcpy(nd.hi1, all.hi, sizeof nd.hi1);
        nd.c0 = ~((0 << 3) | count); nd.c1 = ~0;
        root = 0;
    } else root = bl.build(0, count, 0);
    out.root = root;
    out.depth = bl.max_depth;
    out.nodes.swap(bl.nodes);
    out.order.swap(bl.order);
#ifdef RAYHS_DEV_KNOBS
    // development (profiles/r05/experiments/bvh_node_order.log): the builder emits nodes in depth-first order (a node, its left
    // subtree, its right subtree).  RAYHS_BVH_ORDER = bfs | top<k> | veb<h> re-numbers them: breadth-first throughout; the top k
    // levels breadth-first and every subtree below depth-first; treelets of h levels, breadth-first inside, recursively.
    if (const char *mode = std::getenv("RAYHS_BVH_ORDER")) {
        const int n = (int)out.nodes.size();
        if (out.root >= 0 && n > 1) {
            std::vector<int> seq;  // old ids in their new order
            seq.reserve((size_t)n);
            auto kids = [&](int id, int *k) { int m = 0; if (out.nodes[(size_t)id].c0 >= 0) k[m++] = out.nodes[(size_t)id].c0; if (out.nodes[(size_t)id].c1 >= 0) k[m++] = out.nodes[(size_t)id].c1; return m; };
            std::function<void(int)> dfs = [&](int id) { seq.push_back(id); int k[2]; const int m = kids(id, k); for (int i = 0; i < m; i++) dfs(k[i]); };
            // breadth-first over `levels` levels below `id`; the nodes hanging off the last level are returned in `frontier`
            auto bfs_levels = [&](int id, int levels, std::vector<int> &frontier) {
                std::vector<int> cur{id};
                for (int l = 0; l < levels && !cur.empty(); l++) {
                    std::vector<int> next;
                    for (int x : cur) { seq.push_back(x); int k[2]; const int m = kids(x, k); for (int i = 0; i < m; i++) next.push_back(k[i]); }
                    cur.swap(next);
                }
                frontier = cur;
            };
            const std::string md = mode;
            if (md == "bfs") { std::vector<int> f; bfs_levels(out.root, 1 << 30, f); }
            else if (md.rfind("top", 0) == 0) { std::vector<int> f; bfs_levels(out.root, std::max(1, std::atoi(mode + 3)), f); for (int x : f) dfs(x); }
            else if (md.rfind("veb", 0) == 0) {
                const int h = std::max(1, std::atoi(mode + 3));
                std::function<void(int)> treelet = [&](int id) { std::vector<int> f; bfs_levels(id, h, f); for (int x : f) treelet(x); };
                treelet(out.root);
            } else dfs(out.root);
            if ((int)seq.size() == n) {
                std::vector<int> fresh((size_t)n);
                for (int i = 0; i < n; i++) fresh[(size_t)seq[(size_t)i]] = i;
                std::vector<FNodeBuild> re((size_t)n);
                for (int i = 0; i < n; i++) {
                    FNodeBuild nd = out.nodes[(size_t)seq[(size_t)i]];
                    if (nd.c0 >= 0) nd.c0 = fresh[(size_t)nd.c0];
                    if (nd.c1 >= 0) nd.c1 = fresh[(size_t)nd.c1];
                    re[(size_t)i] = nd;
                }
                out.nodes.swap(re);
                out.root = fresh[(size_t)out.root];
            }
        }
    }
#endif
}

}  // namespace rayhs
