// trace_kernel.hip -- the per-pixel trace loop of rayhs as a persistent HIP
// kernel for gfx950 (MI355X).
//
// What it computes (reference, paths relative to /root/reference):
//   rayTrace / tracePixel / traceRay        src/RayHs.hs:149-166
//   distributedTracePixel (spp > 1)         src/RayHs.hs:169-188
//   rayFromPixel / unproject                src/Projection.hs:22-46
//   closestIntersection / shadowIntersection src/RayHs.hs:58-87
//   plane / sphere tests                    src/Geometry.hs:68-96
//   rayInterBox / rayInter                  src/KDTree.hs:39-56, 96-115
//   triangleIntersection                    src/Mesh.hs:59-82
//   irradiance / accumDiffuse / specular    src/RayHs.hs:89-147
//   lightAt, fresnel, diffuse, colorAt      src/Light.hs:12-17, src/Material.hs:22-33, src/ColorMap.hs:18-58
//
// How it runs here (not how the reference runs):
//   * Persistent waves: a fixed grid sized to the chip; each 64-lane wave pulls
//     8x8-pixel tiles from one atomic ticket counter until the frame is done.
//   * The recursion of irradiance/specular is an explicit per-lane state
//     machine.  Every lane, whatever its shading state, funnels into ONE scene
//     query site per iteration, so primary, shadow, reflection and refraction
//     rays of different pixels share the traversal loop of a wave.
//   * Tree traversal uses a per-lane node stack in LDS (level-major, so a
//     wave's accesses are bank-conflict free), near-child-first ordering by the
//     split axis, and pruning against the closest hit so far; shadow queries
//     stop at the first occluder.  The reference visits both children always;
//     the winner is the same because ties are resolved by the reference's rule
//     (first triangle in a leaf, right-most leaf, first scene object).
//   * All arithmetic is fp64 like the reference.  Built twice: "exact"
//     (-ffp-contract=off, same operations as the reference/oracle) and "fast"
//     (fused multiply-add allowed).
//
// Continuation frames of suspended irradiance calls live in a global workspace
// (14 doubles per level per lane, slot-major so lanes coalesce).
#include <hip/hip_runtime.h>

#include "device_scene.h"

#ifndef RAYHS_KERNEL_SUFFIX
#define RAYHS_KERNEL_SUFFIX fast
#endif
#define RAYHS_CAT2(a, b) a##_##b
#define RAYHS_CAT(a, b) RAYHS_CAT2(a, b)

namespace rayhs {

struct V3 { double x, y, z; };
#define DEV static __device__ __forceinline__

DEV V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV V3 ld3(const double *p) { return mk(p[0], p[1], p[2]); }
DEV V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
DEV V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
DEV V3 mulv(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
DEV V3 neg(V3 a) { return mk(-a.x, -a.y, -a.z); }
DEV V3 scl(double l, V3 a) { return mk(l * a.x, l * a.y, l * a.z); }
DEV double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
DEV V3 cross(V3 a, V3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV V3 unit(V3 a) { return scl(1 / sqrt(dot(a, a)), a); }                    // src/Vec.hs:124-126
DEV V3 reflect(V3 v, V3 n) { return sub(v, scl(2 * dot(v, n), n)); }        // src/Vec.hs:128-130
// Haskell Ord Double: min x y = if x <= y then x else y; max x y = if x <= y then y else x
DEV double hs_min(double x, double y) { return x <= y ? x : y; }
DEV double hs_max(double x, double y) { return x <= y ? y : x; }

constexpr double kEps = 0.000001;                 // src/Geometry.hs:31-32
constexpr double kPiInv = 1 / 3.141592653589793;  // src/Math.hs:12-13

// src/Vec.hs:132-140
DEV bool refract(V3 i, V3 n, double n1, double n2, V3 &out) {
    double n1n2 = n1 / n2;
    double cos0 = -(dot(i, n));
    double sin20 = n1n2 * n1n2 * (1 - cos0 * cos0);
    if (sin20 > 1) return false;
    double coeff = n1n2 * cos0 - sqrt(1.0 - sin20);
    out = add(scl(n1n2, i), scl(coeff, n));
    return true;
}

// Data.Fixed.mod' (src/ColorMap.hs:22,57-58): n - fromInteger(floor(n/d exactly)) * d.
// fma gives the exact sign of n - q*d, which corrects a floor taken on the rounded quotient.
DEV double mod_floor(double n, double d) {
    double q = floor(n / d);
    if (isfinite(q) && d != 0) {
        for (int it = 0; it < 4; it++) {
            double rem = __builtin_fma(-q, d, n);
            if (d > 0) {
                if (rem < 0) q -= 1; else if (rem >= d) q += 1; else break;
            } else {
                if (rem > 0) q -= 1; else if (rem <= d) q += 1; else break;
            }
        }
    }
    return n - q * d;
}

DEV long long imod(long long a, long long m) { long long r = a % m; if (r != 0 && ((r < 0) != (m < 0))) r += m; return r; }

// colorAt (src/ColorMap.hs:18-58)
DEV V3 color_at(const DObject &o, const double *texels, double u, double v) {
    if (o.cmap_kind == 0) return ld3(o.c1);
    if (o.cmap_kind == 1) {
        double s = o.size;
        if ((mod_floor(u, s) - (0.5 * s)) * (mod_floor(v, s) - (0.5 * s)) < 0) return ld3(o.c1);
        return ld3(o.c2);
    }
    long long w = o.tex_w, h = o.tex_h;
    double pu = mod_floor(u, 1) * (double)w;
    double pv = mod_floor(v, 1) * (double)h;
    long long ui = (long long)rint(pu), vi = (long long)rint(pv);  // Haskell round: ties to even
    long long x0 = imod(ui - 1, w), x1 = imod(ui, w), y0 = imod(vi - 1, h), y1 = imod(vi, h);
    double lx = pu - (double)(ui - 1) - 0.5;
    double ly = pv - (double)(vi - 1) - 0.5;
    const double *t = texels + 3 * o.tex_off;
    V3 c0 = ld3(t + 3 * (x0 + w * y0)), c1 = ld3(t + 3 * (x1 + w * y0));
    V3 c2 = ld3(t + 3 * (x0 + w * y1)), c3 = ld3(t + 3 * (x1 + w * y1));
    V3 cx0 = add(scl(lx, c1), scl(1 - lx, c0));
    V3 cx1 = add(scl(lx, c3), scl(1 - lx, c2));
    return add(scl(ly, cx1), scl(1 - ly, cx0));
}

// fresnel (src/Material.hs:26-29); r = r0 1.0 ior precomputed; (^5) = (x*x)*(x*x)*x as GHC's (^)
DEV double fresnel(double r, double cos0) {
    double x = 1 - cos0;
    double x2 = x * x;
    return r + (1 - r) * ((x2 * x2) * x);
}

// splitmix64 finaliser: the build-defined counter sampler (include/rayhs.h, rayhs_sample01)
DEV uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
DEV double sample01(uint64_t seed_mixed, uint64_t pixel, uint32_t sample, uint32_t dim, uint32_t spp) {
    uint64_t ctr = (pixel * (uint64_t)spp + sample) * 2u + dim;
    uint64_t x = mix64(ctr + seed_mixed);
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

struct Ray { V3 o, d; };

// rayFromPixel (src/Projection.hs:22-39); camera constants precomputed on the host
DEV Ray ray_from_pixel(const DCamera &c, double px, double py) {
    double vx = c.apw * (px - c.half_w) / c.w;
    double vy = c.aph * ((-py) + c.half_h) / c.h;
    V3 o, d;
    if (c.proj_kind == 0) { o = mk(vx, vy, 0); d = mk(0, 0, 1); }
    else { d = unit(mk(vx, vy, c.f)); o = mk(0, 0, 0); }
    Ray r;
    r.o = add(o, ld3(c.pos));
    r.d = mk(c.m[0] * d.x + c.m[1] * d.y + c.m[2] * d.z, c.m[3] * d.x + c.m[4] * d.y + c.m[5] * d.z,
             c.m[6] * d.x + c.m[7] * d.y + c.m[8] * d.z);
    return r;
}

// lane states between scene queries
enum : int { ST_NEW_SAMPLE = 0, ST_HIT, ST_SHADOW, ST_EXIT, ST_DONE };
// frame kinds
enum : int { FR_PLASTIC = 0, FR_MIRROR, FR_TRANS_WAIT_RADIANCE, FR_TRANS_WAIT_SPEC };
// query kinds
enum : int { Q_NONE = 0, Q_CLOSEST, Q_SHADOW };

struct Counters { unsigned long long c[10]; };

template <bool EXH, bool COUNT>
struct Tracer {
    const DScene &sc;
    int *stack;      // this lane's LDS column: stack[level * TRACE_BLOCK]
    Counters &cnt;

    // closest-hit record
    double bt, bu, bv;
    int bobj, btri, bleaf;
    int bprim;  // winning plane index i, or sphere index as -(i + 2); -1 for a triangle / miss

    __device__ Tracer(const DScene &s, int *st, Counters &c) : sc(s), stack(st), cnt(c) {}

    // ---- analytic primitives (src/Geometry.hs:68-96) ----
    __device__ __forceinline__ bool plane_t(const DPlane &pl, const Ray &r, double &t) {
        double dDotn = dot(r.d, ld3(pl.n));
        double time = dot(ld3(pl.n), sub(ld3(pl.p), r.o)) / dDotn;
        t = time;
        return fabs(dDotn) > 0 && time > 0;
    }
    __device__ __forceinline__ bool sphere_t(const DSphere &s, const Ray &r, double a, double &t) {
        V3 oc = sub(r.o, ld3(s.c));
        double b = 2.0 * dot(r.d, oc);
        double c = dot(oc, oc) - s.r2;
        double delta = b * b - 4.0 * a * c;
        if (delta < 0.0) return false;
        double sq = sqrt(delta);
        double t0 = 0.5 * ((-b) - sq) / a;
        if (t0 > 0) { t = t0; return true; }
        double t1 = 0.5 * ((-b) + sq) / a;
        if (t1 > 0) { t = t1; return true; }
        return false;
    }

    // rayInterBox (src/KDTree.hs:39-56); returns pass and tmin
    __device__ __forceinline__ bool box_test(const DNode &nd, const Ray &r, V3 inv, double &tmin_out) {
        double t1 = inv.x * (nd.lo[0] - r.o.x), t2 = inv.x * (nd.hi[0] - r.o.x);
        double t3 = inv.y * (nd.lo[1] - r.o.y), t4 = inv.y * (nd.hi[1] - r.o.y);
        double t5 = inv.z * (nd.lo[2] - r.o.z), t6 = inv.z * (nd.hi[2] - r.o.z);
        double tmin = hs_max(hs_max(hs_min(t1, t2), hs_min(t3, t4)), hs_min(t5, t6));
        double tmax = hs_min(hs_min(hs_max(t1, t2), hs_max(t3, t4)), hs_max(t5, t6));
        tmin_out = tmin;
        return !(tmax < 0 || tmin > tmax);
    }

    // triangleIntersection (src/Mesh.hs:59-72)
    __device__ __forceinline__ bool tri_test(const DTri &tr, const Ray &r, double &t, double &u, double &v) {
        V3 e1 = ld3(tr.e1), e2 = ld3(tr.e2);
        V3 p = cross(r.d, e2);
        double det = dot(e1, p);
        double idet = 1 / det;
        V3 t0 = sub(r.o, ld3(tr.p0));
        u = idet * dot(t0, p);
        V3 q = cross(t0, e1);
        v = idet * dot(r.d, q);
        t = idet * dot(e2, q);
        return !(fabs(det) < kEps || u < 0 || u > 1 || v < 0 || (u + v) > 1 || t < kEps);
    }

    // One scene query for this lane.  kind == Q_CLOSEST fills the b* record
    // (bobj = -1 on a miss); kind == Q_SHADOW returns whether any non-Emmit
    // object occludes the light (src/RayHs.hs:74-87): for a point light a hit
    // occludes iff sqrDist origin light > sqrDist origin hit.
    __device__ bool query(int kind, const Ray &r, bool point_light, double light_d2, double light_dist) {
        const bool active = kind != Q_NONE;
        const bool shadow = kind == Q_SHADOW;
        bool occluded = false;
        bt = __builtin_inf(); bu = 0; bv = 0; bobj = -1; btri = -1; bleaf = -1; bprim = -1;

        // ---- planes and spheres: wave-uniform loops, scene constants via scalar loads
        for (int i = 0; i < sc.n_planes; i++) {
            const DPlane &pl = sc.planes[i];
            if (!active || (shadow && !EXH && (occluded || !pl.occluder))) continue;
            if (COUNT) cnt.c[CNT_PRIMS]++;
            double t;
            if (plane_t(pl, r, t)) {
                if (shadow) {
                    if (pl.occluder) {
                        bool front = true;
                        if (point_light) { V3 pos = add(r.o, scl(t, r.d)); V3 dd = sub(r.o, pos); front = light_d2 > dot(dd, dd); }
                        occluded = occluded || front;
                    }
                } else if (t < bt || (t == bt && pl.obj < bobj)) { bt = t; bobj = pl.obj; bprim = i; }
            }
        }
        const double a = dot(r.d, r.d);
        for (int i = 0; i < sc.n_spheres; i++) {
            const DSphere &s = sc.spheres[i];
            if (!active || (shadow && !EXH && (occluded || !s.occluder))) continue;
            if (COUNT) cnt.c[CNT_PRIMS]++;
            double t;
            if (sphere_t(s, r, a, t)) {
                if (shadow) {
                    if (s.occluder) {
                        bool front = true;
                        if (point_light) { V3 pos = add(r.o, scl(t, r.d)); V3 dd = sub(r.o, pos); front = light_d2 > dot(dd, dd); }
                        occluded = occluded || front;
                    }
                } else if (t < bt || (t == bt && s.obj < bobj)) { bt = t; bobj = s.obj; bprim = -(i + 2); }
            }
        }

        // ---- meshes: per-lane traversal with the LDS stack
        if (sc.n_meshes > 0) {
            V3 inv = mk(1 / r.d.x, 1 / r.d.y, 1 / r.d.z);  // src/KDTree.hs:46-48
            for (int m = 0; m < sc.n_meshes; m++) {
                const DMesh &me = sc.meshes[m];
                int node = me.root;
                if (!active || (shadow && !EXH && (occluded || !me.occluder))) node = -1;
                int sp = 0;
                while (true) {
                    if (node < 0) {
                        if (sp == 0) break;
                        sp--;
                        node = stack[sp * TRACE_BLOCK];
                        continue;
                    }
                    const DNode nd = sc.nodes[node];
                    if (COUNT) cnt.c[CNT_NODES]++;
                    double tmin;
                    bool pass = box_test(nd, r, inv, tmin);
                    if (!EXH) {
                        // prune: nothing behind the best hit / the light can matter.  The slack keeps
                        // the decision conservative against rounding of tmin vs. a triangle's t.
                        double lim = shadow ? light_dist : bt;
                        if (tmin > lim + 1e-9 * (1 + fabs(lim))) pass = false;
                    }
                    int next = -1;
                    if (pass) {
                        if (nd.flags & DNODE_LEAF) {
                            const int first = nd.a, last = nd.a + nd.b;
                            for (int k = first; k < last; k++) {
                                if (!EXH && shadow && occluded) break;
                                const DTri tr = sc.tris[k];
                                if (COUNT) cnt.c[CNT_TRIS]++;
                                double t, u, v;
                                if (tri_test(tr, r, t, u, v)) {
                                    if (shadow) {
                                        if (me.occluder) {
                                            bool front = true;
                                            if (point_light) { V3 pos = add(r.o, scl(t, r.d)); V3 dd = sub(r.o, pos); front = light_d2 > dot(dd, dd); }
                                            occluded = occluded || front;
                                        }
                                    } else {
                                        // tie rule: first triangle inside a leaf, right-most leaf inside a
                                        // tree (src/Geometry.hs:54-57, src/KDTree.hs:113-115), first object
                                        // in the scene (src/RayHs.hs:67-71)
                                        bool take = t < bt;
                                        if (t == bt) {
                                            if (me.obj != bobj) take = me.obj < bobj;
                                            else take = first > bleaf;
                                        }
                                        if (take) { bt = t; bu = u; bv = v; bobj = me.obj; btri = k; bleaf = first; bprim = -1; }
                                    }
                                }
                            }
                        } else {
                            int l = nd.a, rr = nd.b;
                            if (!EXH) {
                                // near child first: left holds barycentres below the split (src/KDTree.hs:84-87)
                                int ax = nd.flags & 3;
                                double dax = ax == 0 ? r.d.x : ax == 1 ? r.d.y : r.d.z;
                                if (dax < 0) { int tmp = l; l = rr; rr = tmp; }
                            }
                            if (l >= 0 && rr >= 0) { stack[sp * TRACE_BLOCK] = rr; sp++; next = l; }
                            else next = l >= 0 ? l : rr;
                        }
                    }
                    if (!EXH && shadow && occluded) { sp = 0; next = -1; }
                    node = next;
                }
            }
        }
        return occluded;
    }
};

DEV void frame_store(double *ws, int stride, int level, int slot, double v) { ws[(size_t)(level * FRAME_SLOTS + slot) * (size_t)stride] = v; }
DEV double frame_load(const double *ws, int stride, int level, int slot) { return ws[(size_t)(level * FRAME_SLOTS + slot) * (size_t)stride]; }
DEV void frame_store3(double *ws, int stride, int level, int slot, V3 v) {
    frame_store(ws, stride, level, slot, v.x); frame_store(ws, stride, level, slot + 1, v.y); frame_store(ws, stride, level, slot + 2, v.z);
}
DEV V3 frame_load3(const double *ws, int stride, int level, int slot) {
    return mk(frame_load(ws, stride, level, slot), frame_load(ws, stride, level, slot + 1), frame_load(ws, stride, level, slot + 2));
}
// frame slots
enum : int { FS_KIND = 0, FS_DEPTH = 1, FS_F = 2, FS_S = 3, FS_K = 4, FS_A = 5, FS_RO = 8, FS_RD = 11 };

template <bool EXH, bool COUNT>
static __device__ void trace_body(const DScene &sc, const DCamera &cam, const DParams &prm, void *out,
                                  unsigned long long *counters, unsigned int *ticket, double *frames) {
    extern __shared__ int lds_stack[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int gtid = blockIdx.x * TRACE_BLOCK + tid;
    Counters cnt;
    for (int i = 0; i < 10; i++) cnt.c[i] = 0;
    Tracer<EXH, COUNT> tr(sc, lds_stack + tid, cnt);
    double *ws = frames + gtid;
    const int stride = prm.frame_stride;
    const uint64_t seed_mixed = mix64(prm.seed + 0x9e3779b97f4a7c15ULL);

    // per-lane path state
    int state = ST_DONE;
    int px = 0, lrow = 0, yrow = 0, sample = 0;
    V3 sum = mk(0, 0, 0);
    int depth = 0, fsp = 0;
    Ray ray; ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1);
    int qkind = Q_NONE;
    bool q_point = false; double q_d2 = 0, q_dist = 0;
    // shading context kept across shadow / exit queries
    V3 c_p = mk(0, 0, 0), c_n = mk(0, 0, 0), c_v = mk(0, 0, 0), c_cd = mk(0, 0, 0), c_A = mk(0, 0, 0), c_term = mk(0, 0, 0);
    int c_obj = 0, c_li = 0;
    bool occluded = false;
    bool have_pixel = false;

    while (true) {
        // ---- tile fetch: the whole wave takes the next 8x8 tile when every lane is idle
        if (__ballot(state != ST_DONE) == 0ull) {
            unsigned int tile = 0;
            if (lane == 0) tile = atomicAdd(ticket, 1u);
            tile = __shfl(tile, 0);
            if (tile >= (unsigned)prm.n_tiles) break;
            int ty = (int)tile / prm.tiles_x, tx = (int)tile - ty * prm.tiles_x;
            px = tx * 8 + (lane & 7);
            lrow = ty * 8 + (lane >> 3);
            have_pixel = px < prm.width && lrow < prm.n_rows;
            if (have_pixel) {
                // local row -> image row (stripe interleave, include/rayhs.h)
                int s_local = lrow / prm.stripe_rows, within = lrow - s_local * prm.stripe_rows;
                yrow = prm.row0 + (s_local * prm.stripe_count + prm.stripe_rank) * prm.stripe_rows + within;
                sample = 0;
                sum = mk(0, 0, 0);
                state = ST_NEW_SAMPLE;
            }
        }

        // ---- advance: consume the last query's answer, run shading up to the next query
        qkind = Q_NONE;
        if (state != ST_DONE) {
            V3 result = mk(0, 0, 0);
            // sub-steps inside advance
            enum : int { A_DISPATCH, A_LIGHT_NEXT, A_SPEC, A_RETURN, A_QUERY_READY };
            int step = A_DISPATCH;
            bool spec_with_A = false;
            while (step != A_QUERY_READY && state != ST_DONE) {
                if (step == A_DISPATCH) {
                    if (state == ST_NEW_SAMPLE) {
                        // pixelCoord (src/Image.hs:31-32) + optional sample offset (src/RayHs.hs:178-188)
                        double fx = (double)px, fy = (double)yrow;
                        if (prm.spp > 1) {
                            uint64_t pixel = (uint64_t)px + (uint64_t)prm.width * (uint64_t)yrow;
                            fx = fx + (sample01(seed_mixed, pixel, (uint32_t)sample, 0, (uint32_t)prm.spp) - 0.5);
                            fy = fy + (sample01(seed_mixed, pixel, (uint32_t)sample, 1, (uint32_t)prm.spp) - 0.5);
                        }
                        ray = ray_from_pixel(cam, fx, fy);
                        depth = 0; fsp = 0;
                        if (COUNT) cnt.c[CNT_PRIMARY]++;
                        qkind = Q_CLOSEST; state = ST_HIT; step = A_QUERY_READY;
                    } else if (state == ST_HIT) {
                        // traceRay (src/RayHs.hs:149-154)
                        if (tr.bobj < 0) { result = mk(0, 0, 0); step = A_RETURN; }
                        else {
                            const DObject &ob = sc.objects[tr.bobj];
                            if (COUNT) cnt.c[CNT_SHADED]++;
                            c_obj = tr.bobj;
                            c_v = ray.d;
                            c_p = add(ray.o, scl(tr.bt, ray.d));  // rayAt (src/Geometry.hs:27-29)
                            double hu = 0, hv = 0;
                            const bool need_uv = ((ob.mat_kind == 1 || ob.mat_kind == 2) && ob.cmap_kind != 0) || ob.mat_kind == 6;
                            if (tr.btri >= 0) {
                                const DTriAttr &at = sc.attrs[tr.btri];
                                double u = tr.bu, v = tr.bv, w = 1 - u - v;
                                // barycentricInterp u n1 v n2 (1-u-v) n0 (src/Mesh.hs:53-57,73-74)
                                c_n = add(add(scl(u, ld3(at.n1)), scl(v, ld3(at.n2))), scl(w, ld3(at.n0)));
                                hu = u * at.uv1[0] + v * at.uv2[0] + w * at.uv0[0];
                                hv = u * at.uv1[1] + v * at.uv2[1] + w * at.uv0[1];
                            } else {
                                // re-derive the analytic hit from t
                                if (tr.bprim >= 0) {
                                    const DPlane &pl = sc.planes[tr.bprim];
                                    c_n = ld3(pl.n);
                                    if (need_uv) { V3 rel = sub(c_p, ld3(pl.p)); hu = dot(ld3(pl.t), rel); hv = dot(ld3(pl.b), rel); }
                                } else {
                                    const DSphere &s = sc.spheres[-tr.bprim - 2];
                                    c_n = unit(sub(c_p, ld3(s.c)));
                                    if (need_uv) { hu = kPiInv * atan(c_n.z / c_n.x); hv = kPiInv * acos(c_n.y); }  // polar, src/Geometry.hs:96
                                }
                            }
                            // irradiance (src/RayHs.hs:107-147)
                            switch (ob.mat_kind) {
                            case 1: case 2:  // Diffuse, Plastic
                                c_cd = color_at(ob, sc.texels, hu, hv);
                                c_A = mk(0, 0, 0); c_li = 0;
                                step = A_LIGHT_NEXT;
                                break;
                            case 0:  // Mirror
                                spec_with_A = false; step = A_SPEC;
                                break;
                            case 3:  // Emmit
                                result = ld3(ob.ce); step = A_RETURN;
                                break;
                            case 4: {  // Transparent: try to transmit (src/RayHs.hs:136-143)
                                V3 refdir;
                                if (depth != prm.max_depth && refract(c_v, c_n, 1.0, ob.ior, refdir)) {
                                    c_cd = refdir;  // keep refDir across the exit query
                                    ray.o = add(c_p, scl(kEps, refdir)); ray.d = refdir;  // rayEps
                                    if (COUNT) cnt.c[CNT_REFRACT_EXIT]++;
                                    qkind = Q_CLOSEST; state = ST_EXIT; step = A_QUERY_READY;
                                } else { spec_with_A = false; step = A_SPEC; }
                                break;
                            }
                            case 5: result = c_n; step = A_RETURN; break;           // ShowNormal
                            default: result = mk(hu, hv, 0); step = A_RETURN; break; // ShowUV
                            }
                        }
                    } else if (state == ST_SHADOW) {
                        // accumDiffuse fold step (src/RayHs.hs:90-96)
                        c_A = add(c_A, occluded ? mk(0, 0, 0) : c_term);
                        c_li++;
                        step = A_LIGHT_NEXT;
                    } else {  // ST_EXIT: the transmitted ray's exit point (src/RayHs.hs:140-143)
                        const DObject &ob = sc.objects[c_obj];
                        bool have = false;
                        if (tr.bobj >= 0) {
                            V3 outp = add(ray.o, scl(tr.bt, ray.d));
                            V3 outn;
                            if (tr.btri >= 0) {
                                const DTriAttr &at = sc.attrs[tr.btri];
                                double u = tr.bu, v = tr.bv, w = 1 - u - v;
                                outn = add(add(scl(u, ld3(at.n1)), scl(v, ld3(at.n2))), scl(w, ld3(at.n0)));
                            } else {
                                if (tr.bprim >= 0) outn = ld3(sc.planes[tr.bprim].n);
                                else outn = unit(sub(outp, ld3(sc.spheres[-tr.bprim - 2].c)));
                            }
                            V3 outdir;
                            if (refract(c_cd, neg(outn), ob.ior, 1.0, outdir)) {
                                have = true;
                                // suspend this irradiance call: radiance child first, specular child later
                                V3 rf = reflect(c_v, c_n);
                                frame_store(ws, stride, fsp, FS_KIND, (double)FR_TRANS_WAIT_RADIANCE);
                                frame_store(ws, stride, fsp, FS_DEPTH, (double)depth);
                                frame_store(ws, stride, fsp, FS_F, fresnel(ob.fr, dot(c_n, neg(c_v))));
                                frame_store(ws, stride, fsp, FS_S, dot(rf, c_n));
                                frame_store(ws, stride, fsp, FS_K, ob.one_m_r0);
                                frame_store3(ws, stride, fsp, FS_RO, add(c_p, scl(kEps, rf)));
                                frame_store3(ws, stride, fsp, FS_RD, rf);
                                fsp++;
                                ray.o = add(outp, scl(kEps, outdir)); ray.d = outdir;
                                depth = depth + 1;
                                if (COUNT) cnt.c[CNT_REFRACT]++;
                                qkind = Q_CLOSEST; state = ST_HIT; step = A_QUERY_READY;
                            }
                        }
                        if (!have) { spec_with_A = false; step = A_SPEC; }
                    }
                } else if (step == A_LIGHT_NEXT) {
                    const DObject &ob = sc.objects[c_obj];
                    if (c_li < sc.n_lights) {
                        // lightAt (src/Light.hs:12-17) + diffuse (src/Material.hs:31-33)
                        const DLight &L = sc.lights[c_li];
                        V3 ld, lc;
                        if (L.kind == 0) { ld = ld3(L.v); lc = ld3(L.c); q_point = false; q_d2 = 0; q_dist = __builtin_inf(); }
                        else {
                            V3 dv = sub(ld3(L.v), c_p);
                            double d = sqrt(dot(dv, dv));
                            double den = 1.0 + d / L.radius;
                            double falloff = 1.0 / (den * den);
                            ld = scl(1 / d, dv);
                            lc = scl(falloff, ld3(L.c));
                            q_point = true; q_dist = d;
                        }
                        c_term = scl(hs_max(dot(ld, c_n), 0) * kPiInv, mulv(c_cd, lc));
                        ray.o = add(c_p, scl(kEps, ld)); ray.d = ld;  // rayEps p ld
                        if (q_point) { V3 dd = sub(ray.o, ld3(L.v)); q_d2 = dot(dd, dd); }  // sqrDist origin p (src/RayHs.hs:87)
                        if (COUNT) cnt.c[CNT_SHADOW]++;
                        qkind = Q_SHADOW; state = ST_SHADOW; step = A_QUERY_READY;
                    } else if (ob.mat_kind == 1) {
                        result = add(scl(0.2, c_cd), c_A);  // Diffuse (src/RayHs.hs:111-114)
                        step = A_RETURN;
                    } else { spec_with_A = true; step = A_SPEC; }  // Plastic (src/RayHs.hs:117-120)
                } else if (step == A_SPEC) {
                    // fresnel * specular (src/RayHs.hs:99-104,119,124,132)
                    const DObject &ob = sc.objects[c_obj];
                    double F = fresnel(ob.fr, dot(c_n, neg(c_v)));
                    if (depth < prm.max_depth) {
                        V3 rf = reflect(c_v, c_n);
                        frame_store(ws, stride, fsp, FS_KIND, (double)(spec_with_A ? FR_PLASTIC : FR_MIRROR));
                        frame_store(ws, stride, fsp, FS_DEPTH, (double)depth);
                        frame_store(ws, stride, fsp, FS_F, F);
                        frame_store(ws, stride, fsp, FS_S, dot(rf, c_n));
                        if (spec_with_A) frame_store3(ws, stride, fsp, FS_A, c_A);
                        fsp++;
                        ray.o = add(c_p, scl(kEps, rf)); ray.d = rf;
                        depth = depth + 1;
                        if (COUNT) cnt.c[CNT_REFLECT]++;
                        qkind = Q_CLOSEST; state = ST_HIT; step = A_QUERY_READY;
                    } else {
                        V3 spec = scl(F, mk(0, 0, 0));
                        result = spec_with_A ? add(c_A, spec) : spec;
                        step = A_RETURN;
                    }
                } else {  // A_RETURN: hand `result` to the suspended caller, or finish the sample
                    if (fsp == 0) {
                        sum = add(sum, result);  // foldl (+) black (src/RayHs.hs:171)
                        sample++;
                        if (sample < prm.spp) { state = ST_NEW_SAMPLE; step = A_DISPATCH; }
                        else {
                            V3 c = prm.spp == 1 ? result : scl(1.0 / (double)prm.spp, sum);
                            size_t o = ((size_t)lrow * (size_t)prm.width + (size_t)px) * 3;
                            if (prm.out_format == 0) { float *f = (float *)out; f[o] = (float)c.x; f[o + 1] = (float)c.y; f[o + 2] = (float)c.z; }
                            else if (prm.out_format == 1) { double *f = (double *)out; f[o] = c.x; f[o + 1] = c.y; f[o + 2] = c.z; }
                            else {
                                // toIntC (src/Image.hs:54-55): truncate (255 * min c 1), saturated to int16
                                short *q = (short *)out;
                                double v0 = 255 * hs_min(c.x, 1), v1 = 255 * hs_min(c.y, 1), v2 = 255 * hs_min(c.z, 1);
                                q[o] = (short)(v0 < -32768.0 ? -32768 : (int)v0);
                                q[o + 1] = (short)(v1 < -32768.0 ? -32768 : (int)v1);
                                q[o + 2] = (short)(v2 < -32768.0 ? -32768 : (int)v2);
                            }
                            if (COUNT) cnt.c[CNT_PIXELS]++;
                            state = ST_DONE;
                        }
                    } else {
                        int lv = fsp - 1;
                        int kind = (int)frame_load(ws, stride, lv, FS_KIND);
                        double F = frame_load(ws, stride, lv, FS_F), S = frame_load(ws, stride, lv, FS_S);
                        if (kind == FR_PLASTIC) {
                            result = add(frame_load3(ws, stride, lv, FS_A), scl(F, scl(S, result)));
                            fsp = lv;
                        } else if (kind == FR_MIRROR) {
                            result = scl(F, scl(S, result));
                            fsp = lv;
                        } else if (kind == FR_TRANS_WAIT_RADIANCE) {
                            frame_store3(ws, stride, lv, FS_A, result);
                            frame_store(ws, stride, lv, FS_KIND, (double)FR_TRANS_WAIT_SPEC);
                            ray.o = frame_load3(ws, stride, lv, FS_RO);
                            ray.d = frame_load3(ws, stride, lv, FS_RD);
                            depth = (int)frame_load(ws, stride, lv, FS_DEPTH) + 1;
                            if (COUNT) cnt.c[CNT_REFLECT]++;
                            qkind = Q_CLOSEST; state = ST_HIT; step = A_QUERY_READY;
                        } else {
                            double K = frame_load(ws, stride, lv, FS_K);
                            result = add(scl(K, frame_load3(ws, stride, lv, FS_A)), scl(F, scl(S, result)));
                            fsp = lv;
                        }
                    }
                }
            }
        }

        // ---- the single scene-query site shared by every lane of the wave
        occluded = tr.query(qkind, ray, q_point, q_d2, q_dist);
    }

    if (COUNT) {
        for (int i = 0; i < 10; i++) {
            unsigned long long v = cnt.c[i];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) atomicAdd(&counters[i], v);
        }
    }
}

}  // namespace rayhs

using namespace rayhs;

#define RAYHS_DEFINE_KERNEL(NAME, EXH, COUNT)                                                                     \
    extern "C" __global__ void __launch_bounds__(TRACE_BLOCK)                                                     \
    RAYHS_CAT(NAME, RAYHS_KERNEL_SUFFIX)(DScene sc, DCamera cam, DParams prm, void *out,                          \
                                         unsigned long long *counters, unsigned int *ticket, double *frames) {    \
        trace_body<EXH, COUNT>(sc, cam, prm, out, counters, ticket, frames);                                      \
    }

RAYHS_DEFINE_KERNEL(rayhs_trace, false, false)
RAYHS_DEFINE_KERNEL(rayhs_trace_count, false, true)
RAYHS_DEFINE_KERNEL(rayhs_trace_exh_count, true, true)

// Host-side launcher of this translation unit's kernels (called from abi.cpp).
namespace rayhs {
hipError_t RAYHS_CAT(launch_trace, RAYHS_KERNEL_SUFFIX)(int variant, int grid, size_t lds_bytes, hipStream_t stream,
                                                        const DScene &sc, const DCamera &cam, const DParams &prm, void *out,
                                                        unsigned long long *counters, unsigned int *ticket, double *frames) {
    dim3 g((unsigned)grid), b(TRACE_BLOCK);
    switch (variant) {
    case 0: hipLaunchKernelGGL(RAYHS_CAT(rayhs_trace, RAYHS_KERNEL_SUFFIX), g, b, lds_bytes, stream, sc, cam, prm, out, counters, ticket, frames); break;
    case 1: hipLaunchKernelGGL(RAYHS_CAT(rayhs_trace_count, RAYHS_KERNEL_SUFFIX), g, b, lds_bytes, stream, sc, cam, prm, out, counters, ticket, frames); break;
    default: hipLaunchKernelGGL(RAYHS_CAT(rayhs_trace_exh_count, RAYHS_KERNEL_SUFFIX), g, b, lds_bytes, stream, sc, cam, prm, out, counters, ticket, frames); break;
    }
    return hipGetLastError();
}
int RAYHS_CAT(occupancy_trace, RAYHS_KERNEL_SUFFIX)(int variant, size_t lds_bytes) {
    int n = 0;
    const void *f = variant == 0 ? (const void *)RAYHS_CAT(rayhs_trace, RAYHS_KERNEL_SUFFIX)
                  : variant == 1 ? (const void *)RAYHS_CAT(rayhs_trace_count, RAYHS_KERNEL_SUFFIX)
                                 : (const void *)RAYHS_CAT(rayhs_trace_exh_count, RAYHS_KERNEL_SUFFIX);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, TRACE_BLOCK, lds_bytes) != hipSuccess) return 0;
    return n;
}
}  // namespace rayhs
