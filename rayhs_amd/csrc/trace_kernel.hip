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
//   * Persistent waves: a fixed grid sized to the chip.  Work items are (pixel, sample) pairs in
//     Morton order inside 8x8-pixel tiles; a wave draws a chunk of 64 consecutive items from the ticket
//     counter of its XCD's band of the frame (8 bands, requested one chunk ahead, next band when dry),
//     works on it in lock-step -- its lanes hold samples of the same few pixels -- and draws the next
//     one when all 64 are done; a wave that finds every band dry -- it learns that from a line of eight flags
//     in its own XCD's L2, not from the counters -- leaves.  Tiles are handed out slowest-first: by the chunk
//     times earlier frames of the same geometry measured, or by a static estimate for the first frame
//     (rayhs_static_costs, rayhs_order_tiles).  When spp divides 64 a pixel's samples sit in one chunk
//     and are folded in sample order (the reference's foldl) inside the wave; any other spp > 1 goes
//     through a sample buffer and rayhs_resolve.
//   * The recursion of irradiance/specular is an explicit per-lane state machine.  Every lane, whatever
//     its shading state, funnels into ONE scene query site per iteration, so primary, shadow,
//     reflection and refraction rays share the traversal code of a wave.  The sections of the state
//     machine hand their vectors over through the lane's LDS context slots, not through registers.
//   * Planes and spheres are wave-uniform loops over records fetched through the scalar cache, one
//     batch of loads and one wave-level branch per primitive; planes are grouped by the axis of their
//     normal so that an axis-aligned wall costs one product per dot product.
//   * Mesh traversal uses the BVH of bvh_build.cpp: 64-byte nodes holding both children's boxes as
//     mesh-centred fp32 planes (one fused multiply-add per plane, culling only).  A wave whose
//     entering rays share an octant or form a narrow cone walks it as ONE packet (one node, one stack,
//     scalar fetches for the whole wave); otherwise every lane walks on its own with a stack of
//     inner-node indices in LDS (level-major: bank-conflict free), nearest child first, leaves
//     postponed so that the fp64 triangle tests run with many lanes active.  Both prune against the
//     closest hit and leave on any hit for shadow queries.  Boxes only decide which triangles are
//     tested; the winner is the reference's because every triangle test is the reference's fp64
//     Moller-Trumbore and ties are resolved by its rule (first triangle in a leaf, right-most leaf,
//     first scene object).  Where the reference's own fp64 boxes are NOT a conservative cull its
//     behaviour is replayed: rays with a (near-)zero direction component -- and rays with a huge or non-finite
//     component, which the fp32 boxes cannot hold -- walk the reference tree (meshes_reference), near-edge
//     hits are checked against the reference's box path (reference_path_admits).
//   * The exhaustive kernels walk the reference's own tree for every ray (fp64 boxes, both children
//     always) and visit exactly its node set.
//   * Everything that enters a result is fp64 like the reference.  Built twice: "exact"
//     (-ffp-contract=off, same operations as the reference/oracle, the default) and "fast" (fused
//     multiply-add allowed, RAYHS_FLAG_FMA).
//   * What bounds it is instruction throughput (profiles/r05/micro/issue_rates.txt): a CU's ONE scalar unit
//     (1.05 cycles per scalar instruction of any of its 16 waves) and its four vector pipes (2 cycles per wave64 fp32 add / mul / fma,
//     4 per fp64 add / mul / fma, integer, compare, select, convert; 16 per fp64 rcp / rsq / sqrt) are both about two thirds busy.
//   * To run 4 waves per SIMD the kernel must fit 128 VGPRs and 40 KiB of LDS per workgroup: the
//     shading context of a hit (p, n, cd, A and, when LDS has room, the light term) lives in LDS; the
//     view direction and the continuation frames of suspended irradiance calls in a global workspace
//     (slot-major so lanes coalesce); kernel arguments are read from the kernarg segment where they
//     are used instead of being held in SGPRs (KArgs) -- the fields a code path reads together as ONE
//     wide scalar load (ParamsHot, PointersHot, the scene head in begin(), the head of a plane record).
// Development switches: RAYHS_PROFILE_LOOPS / RAYHS_PROFILE_TIME (diagnostic builds with trip counters
// or s_memtime stamps, read with RAYHS_DEBUG_COUNTERS=1), RAYHS_WAVE_ENDS (per-wave end times from the
// production kernel, tools/wave_ends.py), RAYHS_PROBE_PLANES2 / RAYHS_PROBE_CULL2 (a section run twice: what
// it costs), RAYHS_NO_PACKET / RAYHS_PACKET_ALWAYS, RAYHS_NO_SETPRIO.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "device_scene.h"
#include "short_div.h"

#ifndef RAYHS_KERNEL_SUFFIX
#define RAYHS_KERNEL_SUFFIX fast
#endif
#define RAYHS_CAT2(a, b) a##_##b
#define RAYHS_CAT(a, b) RAYHS_CAT2(a, b)

namespace rayhs {

struct V3 { double x, y, z; };
#define DEV static __device__ __forceinline__

// Everything the kernel is given.  It is passed BY VALUE as the one kernel parameter and never touched through that
// parameter: the code reads it from the kernarg segment (constant address space -> scalar loads) at the point of use,
// through a pointer that is laundered at every mention so that the compiler cannot hoist those loads to the top of the
// kernel.  Preloaded up front, the ~110 dwords of scene pointers, camera and launch parameters would have to live in
// SGPRs across the whole main loop; gfx950 has 102 of them, and the overflow is spilled into VGPR lanes and read
// back with v_readlane (a VALU instruction) at every use -- a quarter of the kernel's static VALU count before this.
struct KArgs {
    DScene sc;
    DCamera cam;
    DParams prm;
    void *out;
    unsigned long long *counters, *ticket;
    double *frames;
};

// Scene tables (planes, spheres, meshes, objects, lights) are read through the constant address
// space: a wave-uniform index then becomes a scalar load into SGPRs instead of 64 identical
// per-lane loads into VGPRs.  LDS columns are addressed through explicit local-address-space
// pointers so that they compile to ds_read/ds_write rather than flat accesses.
template <class T> using cptr = const T __attribute__((address_space(4))) *;
template <class T> static __device__ __forceinline__ cptr<T> as_const(const T *p) { return (cptr<T>)(unsigned long long)p; }
static __device__ __forceinline__ cptr<KArgs> kargs() {
    cptr<KArgs> p = (cptr<KArgs>)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
#define SC (kargs()->sc)
#define PRM (kargs()->prm)
// What a wave reads of the launch parameters when it draws a chunk of work: the first 64 bytes of DParams, fetched with ONE
// sixteen-dword scalar load.  (Read field by field where they are used, every one of them was a scalar load of its own with a
// wait behind it, a chain of about twenty round trips through the scalar cache per chunk.)
struct ParamsHot {
    int spp, spp_shift, width, n_rows, tiles_x; float inv_tiles_x; int n_tiles, wave_resolve, out_format, stripe_count, row0, stripe_shift;
    uint64_t seed; double inv_spp;
};
static_assert(sizeof(ParamsHot) == 64 && offsetof(DParams, inv_spp) == 56 && offsetof(DParams, seed) == 48 && offsetof(DParams, stripe_shift) == 44 &&
              offsetof(DParams, inv_tiles_x) == 20 && offsetof(DParams, wave_resolve) == 28, "ParamsHot mirrors the head of DParams");
// ... and the pointers it follows: the tail of DParams and the four pointers behind it in KArgs, 64 bytes, one load
struct PointersHot { double *sbuf; const int32_t *tile_order, *stripe_map; float *tile_cost; void *out; unsigned long long *counters, *ticket; double *frames; };
static_assert(sizeof(PointersHot) == 64 && offsetof(KArgs, out) == offsetof(KArgs, prm) + sizeof(DParams) && offsetof(KArgs, frames) == offsetof(KArgs, out) + 24,
              "PointersHot mirrors the tail of DParams and the pointers behind it");
static __device__ __forceinline__ PointersHot pointers_hot() {
    typedef int i16v __attribute__((ext_vector_type(16)));
    const i16v raw = *(cptr<i16v>)&kargs()->prm.sbuf;
    auto ptr = [&](int k) { return ((unsigned long long)(unsigned)raw[2 * k + 1] << 32) | (unsigned long long)(unsigned)raw[2 * k]; };
    PointersHot h;
    h.sbuf = (double *)ptr(0); h.tile_order = (const int32_t *)ptr(1); h.stripe_map = (const int32_t *)ptr(2); h.tile_cost = (float *)ptr(3);
    h.out = (void *)ptr(4); h.counters = (unsigned long long *)ptr(5); h.ticket = (unsigned long long *)ptr(6); h.frames = (double *)ptr(7);
    return h;
}
static __device__ __forceinline__ ParamsHot params_hot() {
    typedef int i16v __attribute__((ext_vector_type(16)));
    const i16v raw = *(cptr<i16v>)&kargs()->prm;
    ParamsHot h;
    h.spp = raw[0]; h.spp_shift = raw[1]; h.width = raw[2]; h.n_rows = raw[3]; h.tiles_x = raw[4]; h.inv_tiles_x = __builtin_bit_cast(float, (int)raw[5]);
    h.n_tiles = raw[6]; h.wave_resolve = raw[7]; h.out_format = raw[8]; h.stripe_count = raw[9]; h.row0 = raw[10]; h.stripe_shift = raw[11];
    h.seed = ((uint64_t)(unsigned)raw[13] << 32) | (uint64_t)(unsigned)raw[12];
    h.inv_spp = __hiloint2double(raw[15], raw[14]);
    return h;
}
typedef double __attribute__((address_space(3))) *lds_dptr;

DEV V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
template <class P> DEV V3 ld3(P p) { return mk(p[0], p[1], p[2]); }
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

// (rcp_in_range, div_in_range, sqrt_in_range: short_div.h -- IEEE division and square root without their range fix-ups, where the range is known)
#ifdef RAYHS_NO_SHORT_DIV
constexpr bool kShortDiv = false;
#else
constexpr bool kShortDiv = true;
#endif

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
template <class OB> DEV V3 color_at(const OB &o, const double *texels, double u, double v) {
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

// fresnel (src/Material.hs:26-29); r = r0 1.0 ior precomputed.  (^ (5 :: Int)) is the flat product u*u*u*u*u that
// GHC's rewrite RULES "^5/Int" (GHC.Real) produce with optimisation on (build.sh:1: -O2), not square-and-multiply
// (RAYHS_FLAG_POW5_SQMUL selects the un-rewritten definition instead, wave-uniformly: DParams.flags_dev bit 1)
DEV double fresnel(double r, double cos0, bool sqmul) {
    double x = 1 - cos0;
    const double x2 = x * x;
    const double p5 = sqmul ? (x2 * x2) * x : ((x2 * x) * x) * x;
    return r + (1 - r) * p5;
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

// final colour of a pixel in the caller's format
DEV void store_pixel(void *out, int out_format, size_t pix, V3 c) {
    const size_t o = pix * 3;
    if (out_format == 0) { float *f = (float *)out; f[o] = (float)c.x; f[o + 1] = (float)c.y; f[o + 2] = (float)c.z; }
    else if (out_format == 1) { double *f = (double *)out; f[o] = c.x; f[o + 1] = c.y; f[o + 2] = c.z; }
    else {
        // toIntC (src/Image.hs:54-55): truncate (255 * min c 1), saturated to int16
        short *q = (short *)out;
        double v0 = 255 * hs_min(c.x, 1), v1 = 255 * hs_min(c.y, 1), v2 = 255 * hs_min(c.z, 1);
        q[o] = (short)(v0 < -32768.0 ? -32768 : (int)v0);
        q[o + 1] = (short)(v1 < -32768.0 ? -32768 : (int)v1);
        q[o + 2] = (short)(v2 < -32768.0 ? -32768 : (int)v2);
    }
}

// one channel of it (o = 3 * pixel + channel)
DEV void store_channel(void *out, int out_format, size_t o, double c) {
    if (out_format == 0) ((float *)out)[o] = (float)c;
    else if (out_format == 1) ((double *)out)[o] = c;
    else { const double v = 255 * hs_min(c, 1); ((short *)out)[o] = (short)(v < -32768.0 ? -32768 : (int)v); }
}

// rare paths kept out of line so that their temporaries do not inflate the kernel's register budget
struct UV2 { double u, v; };
static __device__ __attribute__((noinline)) UV2 polar_uv(V3 n) {  // polar, src/Geometry.hs:96 (by value: out-references put u and v of EVERY hit into scratch)
    UV2 r;
    r.u = kPiInv * atan(n.z / n.x);
    r.v = kPiInv * acos(n.y);
    return r;
}
static __device__ __attribute__((noinline)) V3 color_at_slow(cptr<DObject> objects, int obj, const double *texels, double u, double v) {
    return color_at(objects[obj], texels, u, v);
}

struct Ray { V3 o, d; };

// The reference reaches a triangle only through the fp64 slab tests of its leaf and of every ancestor
// (src/KDTree.hs:39-56, 96-107).  Those tests are exact up to rounding, so a ray that hits the triangle's
// interior passes them; a ray through an edge or a vertex -- which lie ON the box faces -- may fail one by
// an ulp, and the reference then never sees the hit.  For such near-edge candidates of the fast BVH the
// reference's own path is replayed.  Out of line: it runs for about one hit in a million.
static __device__ __attribute__((noinline)) bool reference_path_admits(const DNode *nodes, int leaf_node, V3 o, V3 d) {
    const V3 inv = mk(1 / d.x, 1 / d.y, 1 / d.z);
    for (int n = leaf_node; n >= 0;) {
        const DNode nd = nodes[n];
        double t1 = inv.x * (nd.lo[0] - o.x), t2 = inv.x * (nd.hi[0] - o.x);
        double t3 = inv.y * (nd.lo[1] - o.y), t4 = inv.y * (nd.hi[1] - o.y);
        double t5 = inv.z * (nd.lo[2] - o.z), t6 = inv.z * (nd.hi[2] - o.z);
        double tmin = hs_max(hs_max(hs_min(t1, t2), hs_min(t3, t4)), hs_min(t5, t6));
        double tmax = hs_min(hs_min(hs_max(t1, t2), hs_max(t3, t4)), hs_max(t5, t6));
        if (tmax < 0 || tmin > tmax) return false;
        n = nd.parent;
    }
    return true;
}
constexpr double kEdgeBand = 1e-6;  // barycentric distance from an edge below which the path is replayed

// rayFromPixel (src/Projection.hs:22-39); camera constants precomputed on the host
template <class CAM> DEV Ray ray_from_pixel(const CAM &c, double px, double py) {
    const double nx = c.apw * (px - c.half_w), ny = c.aph * ((-py) + c.half_h);
    // (the two divisions and the normalisation in their short forms -- short_div.h -- when the launcher found the camera's constants in range,
    // DCamera::short_ok: 18 vector instructions fewer per chunk of primary rays.  A zero numerator -- the centre column or row -- is +0 here,
    // x - x, times a positive extent: the short quotient of +0 by a positive divisor is the +0 of the full one)
    const bool short_ok = kShortDiv && c.short_ok != 0;   // (wave-uniform)
    double vx, vy;
    if (short_ok) { vx = div_in_range(nx, c.w); vy = div_in_range(ny, c.h); }
    else { vx = nx / c.w; vy = ny / c.h; }
    V3 o, d;
    if (c.proj_kind == 0) { o = mk(vx, vy, 0); d = mk(0, 0, 1); }
    else {
        const V3 v = mk(vx, vy, c.f);
        const double dd = dot(v, v);
        d = scl(short_ok ? rcp_in_range(sqrt_in_range(dd)) : 1 / sqrt(dd), v);   // unit, src/Vec.hs:124-126
        o = mk(0, 0, 0);
    }
    Ray r;
    r.o = add(o, ld3(c.pos));
    r.d = mk(c.m[0] * d.x + c.m[1] * d.y + c.m[2] * d.z, c.m[3] * d.x + c.m[4] * d.y + c.m[5] * d.z,
             c.m[6] * d.x + c.m[7] * d.y + c.m[8] * d.z);
    return r;
}

// lane states between scene queries
enum : int { ST_HIT = 1, ST_SHADOW, ST_EXIT, ST_DONE };
// frame kinds
enum : int { FR_PLASTIC = 0, FR_MIRROR, FR_TRANS_WAIT_RADIANCE, FR_TRANS_WAIT_SPEC };
// query kinds
enum : int { Q_NONE = 0, Q_CLOSEST, Q_SHADOW, Q_SHADOW_POINT };  // (a shadow query towards a point light is a kind of its own: a yes/no carried beside the kind
                                                                // across the shading sections cost a wave mask and three scalar instructions at every join)

#ifdef RAYHS_PROFILE_LOOPS
constexpr int kCounterSlots = CNT_SLOTS_DEV;  // + (active lanes, wave executions) pairs of the shading sections and query loops
#else
constexpr int kCounterSlots = 16;
#endif
struct Counters { unsigned long long c[kCounterSlots]; };
// development counters (slots 10..15, -DRAYHS_PROFILE_LOOPS): lane- and wave-level trip counts of the hot loops
// -DRAYHS_PROFILE_TIME: the counting kernel is built like the production kernel (no work counters, same
// occupancy) but lane 0 of every wave accumulates shader-clock cycles per phase into slots 10..15.
#ifdef RAYHS_PROFILE_TIME
#define TIME_DECL unsigned long long t_last_ = __builtin_amdgcn_s_memtime()
#define TIME_ADD(slot) do { unsigned long long n_ = __builtin_amdgcn_s_memtime(); cnt.c[slot] += n_ - t_last_; t_last_ = n_; } while (0)
constexpr bool kProfileTime = true;
#else
#define TIME_DECL do { } while (0)
#define TIME_ADD(slot) do { } while (0)
constexpr bool kProfileTime = false;
#endif
#ifdef RAYHS_PROFILE_LOOPS
#define LOOP_TICK(slot) do { if (COUNT) { cnt.c[slot]++; if ((int)__lane_id() == __ffsll((unsigned long long)__ballot(true)) - 1) cnt.c[(slot) + 1]++; } } while (0)
#else
#define LOOP_TICK(slot) do { } while (0)
#endif

// A culling limit in fp32: for v >= 0 a float >= v, at most three ulps above it (0 stays 0, inf stays inf).  A NEGATIVE v -- the
// closest hit so far lies in front of the point where the ray enters the mesh -- is moved further below zero, which is all
// met() asks of it: any limit < 0 means "nothing in this box can beat the hit already found" (max(tn, 0) <= limit fails).
// (Subnormal results are not rounded up; a limit that small only ever meets the t >= 1e-6 of a triangle hit.)
DEV float f32_up(double v) {
    return (float)v * 1.00000024f;
}

constexpr int kTileBits = 28, kTileMask = (1 << kTileBits) - 1;  // tile-order entry: tile index | issue-priority class << 28

// ---- per-lane reads of scene records: uniform base (an SGPR pair) + 32-bit byte offset, so that the load uses the
// scalar-base addressing form instead of 64-bit per-lane pointer arithmetic (abi.cpp keeps every array < 4 GiB)
template <typename T> DEV const T &rec(const T *base, int index) {
    return *(const T *)((const char *)base + (unsigned)index * (unsigned)sizeof(T));
}

// wave-uniform reads of a record through the scalar cache: base + UNSIGNED 32-bit byte offset, which the scalar load takes as
// its offset register (a signed index costs a sign extension, a 64-bit shift and a 64-bit add per fetch)
template <typename T> DEV const T __attribute__((address_space(4))) &crec(const T *base, int index) {
    return *(cptr<T>)((unsigned long long)base + (unsigned long long)((unsigned)index * (unsigned)sizeof(T)));
}

template <bool EXH, bool COUNT, typename StackT>
struct Tracer {
    typedef StackT __attribute__((address_space(3))) *stack_ptr;
    stack_ptr stack;  // this lane's LDS column: stack[level * TRACE_BLOCK]
    Counters &cnt;
#ifdef RAYHS_PROFILE_LOOPS
    volatile unsigned __attribute__((address_space(3))) *census_words = nullptr;  // one word per wave of the workgroup (per-lane-walk census, traverse())
#endif

    // closest-hit record
    // (the barycentrics of a winning triangle are not kept: the shading section re-runs the test on that one triangle,
    // which yields the same doubles; its place in the reference's order is not kept either: an exact tie between two
    // triangles of one mesh -- a ray through a shared edge or vertex -- reads it back from the triangle's record)
    double bt;
    int bobj, btri;
    int bprim;  // winning plane index i, or sphere index as -(i + 2); -1 for a triangle / miss
    bool bref;  // btri indexes the reference-order arrays (hit found by the reference tree), else the BVH-order ones
    bool occ;   // shadow query: an occluder was found

    unsigned long long m_armed;  // wave mask: begin() left the meshes of these lanes' queries to the fast BVH (traverse())

    // traversal state of the fast BVH: a local of traverse(), NOT part of the tracer (as members these floats sat in
    // a scratch slot between the calls, loaded at every entry and stored at every exit)
    struct Walk {
        int t_cur, t_sp;       // current inner node (FNODE_NONE = none), stack depth
        int t_pa, t_pb;        // postponed leaves
        float t_ax, t_ay, t_az, t_bx, t_by, t_bz;  // per-plane slab coefficients of the current mesh (see pbox)
        float t_enter;                             // shift of the traversal origin along the ray
        __device__ __forceinline__ bool traversing() const { return t_cur != FNODE_NONE || t_pa != FNODE_NONE || t_sp > 0; }
        __device__ __forceinline__ void stop_traversal() { t_cur = FNODE_NONE; t_pa = FNODE_NONE; t_pb = FNODE_NONE; t_sp = 0; }
    };

    __device__ Tracer(stack_ptr st, Counters &c) : stack(st), cnt(c) {
        bt = 0; bobj = -1; btri = -1; bprim = -1; bref = EXH; occ = false; m_armed = 0ull;
    }

    // ---- analytic primitives (src/Geometry.hs:68-96) ----
    // Plane (src/Geometry.hs:68-76): time = (n . (p - o)) / (d . n), a hit iff |d . n| > 0 and time > 0.  The two
    // dot products come first: a quotient of operands with opposite signs (or a zero / NaN operand) can never be
    // > 0, whatever the rounding, so when no lane of the wave can hit the division is skipped for the whole wave.
    // AXIS 0..2: the plane's normal has the single non-zero component AXIS (abi.cpp groups the planes so).  For a finite
    // ray the sums  d.x*n.x + d.y*n.y + d.z*n.z  and  n.(p - o)  then consist of that one product plus two exact zeros,
    // so the one product is the same double (up to the sign of a zero result, which no comparison below can see).
    template <int AXIS> __device__ __forceinline__ void plane_terms(double nk, double pk, V3 pn, V3 pp, const Ray &r, double &num, double &den) {
        if (AXIS == 0) { den = r.d.x * nk; num = nk * (pk - r.o.x); }
        else if (AXIS == 1) { den = r.d.y * nk; num = nk * (pk - r.o.y); }
        else if (AXIS == 2) { den = r.d.z * nk; num = nk * (pk - r.o.z); }
        else {
            den = dot(r.d, pn);
            num = dot(pn, sub(pp, r.o));
        }
    }
    // A SUPERSET of the lanes where num / den > 0 is all the caller needs (it only decides whether the wave divides at all;
    // the division's own result is then tested exactly): operands of equal sign.  One integer instruction on the high words
    // and one compare, instead of four fp64 compares and three mask operations; zeros and NaNs of equal sign bit pass and
    // are rejected by the exact test.
    static __device__ __forceinline__ unsigned long long quotient_may_be_positive(double num, double den) {  // wave mask
        int x;
        asm("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(__double2hiint(num)), "v"(__double2hiint(den)));  // (left to the compiler this became a 64-bit exclusive-or and a 64-bit compare)
        return __ballot(x >= 0);
    }
    // Sphere (src/Geometry.hs:78-85): the discriminant first; the roots (a square root and two divisions) only when
    // some lane of the wave has delta >= 0.
    template <class SP> __device__ __forceinline__ void sphere_terms(const SP &s, const Ray &r, double a, double &b, double &delta) {
        V3 oc = sub(r.o, ld3(s.c));
        b = 2.0 * dot(r.d, oc);
        double c = dot(oc, oc) - s.r2;
        delta = b * b - 4.0 * a * c;
    }
    static __device__ __forceinline__ unsigned long long sphere_roots(double a, double b, double delta, double &t) {  // wave mask of hits
        // guards of src/Geometry.hs:81-85 in order: delta < 0 -> miss; t0 > 0 -> t0; t1 > 0 -> t1; else miss.
        // Both roots are formed without branching (a NaN from sqrt of a negative delta is masked by the guard).
        double sq = sqrt(delta);
        double t0 = 0.5 * ((-b) - sq) / a;
        double t1 = 0.5 * ((-b) + sq) / a;
        const bool use0 = t0 > 0;
        t = use0 ? t0 : t1;
        return ~__ballot(delta < 0.0) & (__ballot(t0 > 0) | __ballot(t1 > 0));
    }

    // rayInterBox (src/KDTree.hs:39-56); returns pass and tmin
    template <class BP> __device__ __forceinline__ bool box_test(BP lo, BP hi, const Ray &r, V3 inv, double &tmin_out) {
        double t1 = inv.x * (lo[0] - r.o.x), t2 = inv.x * (hi[0] - r.o.x);
        double t3 = inv.y * (lo[1] - r.o.y), t4 = inv.y * (hi[1] - r.o.y);
        double t5 = inv.z * (lo[2] - r.o.z), t6 = inv.z * (hi[2] - r.o.z);
        double tmin = hs_max(hs_max(hs_min(t1, t2), hs_min(t3, t4)), hs_min(t5, t6));
        double tmax = hs_min(hs_min(hs_max(t1, t2), hs_max(t3, t4)), hs_max(t5, t6));
        tmin_out = tmin;
        return !(tmax < 0 || tmin > tmax);
    }

    // triangleIntersection (src/Mesh.hs:59-72)
    // SHORT (the fast BVH's walks only): 1 / det by rcp_in_range -- the same double wherever the lane is not rejected by |det| < eps
    template <bool SHORT = false, class TR> __device__ __forceinline__ bool tri_test(const TR &tr, const Ray &r, double &t, double &u, double &v) {
        V3 e1 = ld3(tr.e1), e2 = ld3(tr.e2);
        V3 p = cross(r.d, e2);
        double det = dot(e1, p);
        double idet = (SHORT && kShortDiv) ? rcp_in_range(det) : 1 / det;
        V3 t0 = sub(r.o, ld3(tr.p0));
        u = idet * dot(t0, p);
        V3 q = cross(t0, e1);
        v = idet * dot(r.d, q);
        t = idet * dot(e2, q);
        return !(fabs(det) < kEps || u < 0 || u > 1 || v < 0 || (u + v) > 1 || t < kEps);
    }

    // the same test on a triangle held as three vectors (registers): what the hit and exit sections re-run on the winning triangle
    static __device__ __forceinline__ bool tri_test_v(V3 p0, V3 e1, V3 e2, const Ray &r, double &t, double &u, double &v) {
        V3 p = cross(r.d, e2);
        double det = dot(e1, p);
        double idet = 1 / det;
        V3 t0 = sub(r.o, p0);
        u = idet * dot(t0, p);
        V3 q = cross(t0, e1);
        v = idet * dot(r.d, q);
        t = idet * dot(e2, q);
        return !(fabs(det) < kEps || u < 0 || u > 1 || v < 0 || (u + v) > 1 || t < kEps);
    }

    // the same test as a wave mask, straight-line (no early exit on a small determinant: its quotients are then inf or NaN,
    // and the determinant's own bit rejects the lane).  NaN operands make every comparison false, i.e. a hit, as in the
    // reference and in tri_test.
    template <class TR> __device__ __forceinline__ unsigned long long tri_test_mask(const TR &tr, const Ray &r, double &t, double &u, double &v) {
        V3 e1 = ld3(tr.e1), e2 = ld3(tr.e2);
        V3 p = cross(r.d, e2);
        double det = dot(e1, p);
        double idet = kShortDiv ? rcp_in_range(det) : 1 / det;   // (only the packet walk of the fast BVH uses the mask form)
        V3 t0 = sub(r.o, ld3(tr.p0));
        u = idet * dot(t0, p);
        V3 q = cross(t0, e1);
        v = idet * dot(r.d, q);
        t = idet * dot(e2, q);
        return ~(__ballot(fabs(det) < kEps) | __ballot(u < 0) | __ballot(u > 1) | __ballot(v < 0) | __ballot((u + v) > 1) | __ballot(t < kEps));
    }

    // for a point light a hit occludes iff sqrDist origin light > sqrDist origin hit (src/RayHs.hs:84-87)
    __device__ __forceinline__ bool in_front(const Ray &r, double t, bool point_light, double light_d2) {
        if (!point_light) return true;
        V3 pos = add(r.o, scl(t, r.d));
        V3 dd = sub(r.o, pos);
        return light_d2 > dot(dd, dd);
    }

    // tie rule: first triangle inside a leaf, right-most leaf inside a tree (src/Geometry.hs:54-57,
    // src/KDTree.hs:113-115), first object in the scene (src/RayHs.hs:67-71)
    __device__ __forceinline__ bool offer(double t, double u, double v, int obj, int tri, int leaf_first, int rank) {
        bool take = t < bt;
        if (t == bt) {
            if (obj != bobj) take = obj < bobj;
            else {  // the best so far is a triangle of the same mesh (an object is one primitive or one mesh)
                const DTri &best = rec(bref ? SC.tris : SC.ftris, btri);
                const int bleaf = best.leaf_first, brank = best.rank;
                take = leaf_first != bleaf ? leaf_first > bleaf : rank < brank;
            }
        }
        if (take) { bt = t; bobj = obj; btri = tri; bprim = -1; }
        return take;
    }

    // Does the ray's stretch [0, limit] meet the box it enters at tn and leaves at tf?  ONE comparison: max(tn, 0) <= min(tf, limit)
    // (a limit below zero means that nothing in the box can beat the hit already found).  The operands are never NaN: every
    // stored box is finite (an absent child is an EMPTY LEAF with its sibling's box, bvh_build.cpp), and a ray with a zero, huge
    // or non-finite component never gets here (begin() sends it through the reference tree).
    static __device__ __forceinline__ bool met(float tn, float tf, float limf) { return fmaxf(tn, 0.0f) <= fminf(tf, limf); }

    // slab test of one child box: t = plane * A + B (A = 1/d, B = -o/d), one fused multiply-add per plane.  (Two v_fma_f32, not
    // one v_pk_fma_f32: the packed form wants every coefficient twice -- 12 registers instead of 6 across the walks -- and buys no
    // throughput for it: it occupies the SIMD for 4.1 cycles, two plain ones for 2 x 2.2 (profiles/r05/micro/issue_rates.txt).
    // 62 -> 42 spilled registers, dragon frame -1.6 %, dragon_low -2.3 %: profiles/r04/experiments/scalar_fma_boxes.log.)
    // Culling only: the stored boxes are inflated far beyond the fp32 rounding of these operations (abi.cpp).
    typedef float f2 __attribute__((ext_vector_type(2)));
    template <class BP> __device__ __forceinline__ bool pbox(BP b, float ax, float ay, float az, float bx, float by, float bz, float limf, float &tnear) {
        f2 x, y, z;
        x.x = __builtin_fmaf(b[0], ax, bx); x.y = __builtin_fmaf(b[1], ax, bx);
        y.x = __builtin_fmaf(b[2], ay, by); y.y = __builtin_fmaf(b[3], ay, by);
        z.x = __builtin_fmaf(b[4], az, bz); z.y = __builtin_fmaf(b[5], az, bz);
        float tn = fmaxf(fmaxf(fminf(x.x, x.y), fminf(y.x, y.y)), fminf(z.x, z.y));
        float tf = fminf(fminf(fmaxf(x.x, x.y), fmaxf(y.x, y.y)), fmaxf(z.x, z.y));
        tnear = tn;
        return met(tn, tf, limf);
    }

    // the same test as a wave mask (each ballot is of one comparison, which then writes the mask itself)
    template <class BP> __device__ __forceinline__ unsigned long long pbox_mask(BP b, float ax, float ay, float az, float bx, float by, float bz, float limf, float &tnear) {
        f2 x, y, z;
        x.x = __builtin_fmaf(b[0], ax, bx); x.y = __builtin_fmaf(b[1], ax, bx);
        y.x = __builtin_fmaf(b[2], ay, by); y.y = __builtin_fmaf(b[3], ay, by);
        z.x = __builtin_fmaf(b[4], az, bz); z.y = __builtin_fmaf(b[5], az, bz);
        float tn = fmaxf(fmaxf(fminf(x.x, x.y), fminf(y.x, y.y)), fminf(z.x, z.y));
        float tf = fminf(fminf(fmaxf(x.x, x.y), fmaxf(y.x, y.y)), fmaxf(z.x, z.y));
        tnear = tn;
        return __ballot(met(tn, tf, limf));
    }

    // conservative fp32 slab test of the mesh bounds (root cull) as a wave mask; t is relative to the given origin
    __device__ __forceinline__ unsigned long long fbox_mask(const float *lo, const float *hi, float ox, float oy, float oz, float ix, float iy, float iz,
                                                            float limf, float &tnear) {
        float t1 = (lo[0] - ox) * ix, t2 = (hi[0] - ox) * ix;
        float t3 = (lo[1] - oy) * iy, t4 = (hi[1] - oy) * iy;
        float t5 = (lo[2] - oz) * iz, t6 = (hi[2] - oz) * iz;
        // (no NaN among these for the lanes that matter: begin() arms only finite rays with every |d| component above 1e-25)
        float tn = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6));
        float tf = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
        tnear = tn;
        return __ballot(met(tn, tf, limf));
    }

    // Start one scene query for this lane (kind == Q_NONE: lane does not start one).  Q_CLOSEST fills
    // the b* record (bobj = -1 on a miss); Q_SHADOW sets `occ` if any non-Emmit object occludes the
    // light (src/RayHs.hs:74-87).  Planes, spheres and (for the rare rays that need it) the reference
    // tree are finished here; the fast BVH traversal is only armed and runs in traverse().
    __device__ void begin(int kind, const Ray &r, bool point_light, double light_d2) {
        typedef unsigned long long mask_t;
        // (the lanes' kinds as wave masks first, each the ballot of ONE comparison of its own -- the compare then writes the mask itself --
        // and the lane predicates from the masks: a ballot of a predicate that already has other uses goes through a 0/1 vector register
        // and a second compare)
        int kind_ = kind;
        asm volatile("" : "+v"(kind_));
        const mask_t m_active = __ballot(kind_ != Q_NONE);
        const mask_t m_shadow = __ballot(kind_ >= Q_SHADOW);
        const bool active = __builtin_amdgcn_inverse_ballot_w64(m_active), shadow = __builtin_amdgcn_inverse_ballot_w64(m_shadow);
        m_armed = 0ull;
        // (a round in which no lane asks anything -- the last one of every chunk, whose lanes only store their pixels -- ends here: it ran the
        // whole preamble, eighty instructions for nobody, 0.52 M times on the dragon frame)
        if (m_active == 0ull) return;
        if (active) { bt = __builtin_inf(); bobj = -1; btri = -1; bprim = -1; bref = EXH; }

        // ---- planes and spheres: wave-uniform loops, scene constants via scalar loads.  Every per-lane yes/no of
        // these loops is a 64-bit WAVE MASK in scalar registers (lane i = bit i), combined with scalar logic and turned
        // back into a lane predicate only where a select needs one (inverse_ballot: the mask IS the select's condition).
        // As `bool` variables they were kept as 0/1 in vector registers across the loop and re-compared at every use:
        // a dozen vector instructions of bookkeeping per plane beside ~20 of arithmetic.  The only branches are
        // wave-level: nested per-lane ifs cost more scalar exec-mask bookkeeping than the arithmetic they guard.
        const mask_t m_point = __ballot(kind_ == Q_SHADOW_POINT);  // shadow queries towards a point light
        mask_t m_occ = 0;  // lanes whose shadow query has found an occluder
        const double a = dot(r.d, r.d);
        // The "beyond the light" shortcut of point-light shadow queries (rounds without such a query skip it).  A plane met at
        // |t| > K cannot occlude when K^2 a exceeds the light's squared distance by a margin: inFrontOfLight compares that
        // distance with |o - (o + t d)|^2 = t^2 a (1 +- 1e-15) +- |t| |o| 2^-50.  K = sqrt(light_d2) * 1.006 covers the
        // approximate root (v_sqrt_f64, ~1e-8) and any a >= 0.99 (directions towards point lights are unit vectors; the
        // shortcut is off otherwise); K >= 2e-9 |o| keeps the cancellation term below 1e-6 t^2 a.  Then |num| > |den| K
        // implies |num / den| > K (1 - 2^-52): one product and one comparison per plane, no reciprocal.  A NaN or infinite
        // K never passes (the exact test decides).
        double beyond_k = __builtin_inf();
        if (m_point != 0ull) {
            const double o_floor = fmax(fmax(fabs(r.o.x), fabs(r.o.y)), fabs(r.o.z)) * 2e-9;
            double k = __builtin_amdgcn_sqrt(light_d2) * 1.006;
            k = o_floor > k ? o_floor : k;  // (a NaN k stays NaN)
            beyond_k = (__builtin_amdgcn_inverse_ballot_w64(m_point) && a >= 0.99) ? k : __builtin_inf();  // (per lane: other lanes never pass)
        }
        // (the scene's head -- counts, plane groups, the plane and sphere tables -- in ONE sixteen-dword scalar load: field by field
        // these were eight loads, several of them alone in front of a branch that needs their value)
        typedef int i16v __attribute__((ext_vector_type(16)));
        const i16v head = *(cptr<i16v>)&kargs()->sc;
        const int n_planes = head[4], n_spheres = head[5], sphere_occluders = head[6], n_meshes = head[11];
        // |o| + |d| component by component: below 1e30 every component is finite and small enough for the fp32 boxes
        const double magnitude = ((((fabs(r.o.x) + fabs(r.o.y)) + fabs(r.o.z)) + fabs(r.d.x)) + fabs(r.d.y)) + fabs(r.d.z);
        const auto planes = (cptr<DPlane>)(((unsigned long long)(unsigned)head[1] << 32) | (unsigned long long)(unsigned)head[0]);
        // lanes that consider a plane: the closest-hit queries, and the shadow queries still open unless the plane is Emmit
        const mask_t m_closest = EXH ? m_active : (m_active & ~m_shadow);
        mask_t m_shadow_open = EXH ? 0ull : (m_active & m_shadow);  // (shrinks as occluders are found)
        auto plane_range = [&](auto axis_tag, const unsigned begin_off, const unsigned end_off) {
            constexpr int AXIS = decltype(axis_tag)::value;
            // (the loop runs on the record's byte offset -- the group bounds come as byte offsets from the launcher: one scalar add, one compare
            // and the load's own offset operand)
            for (unsigned off = begin_off; off < end_off; off += (unsigned)sizeof(DPlane)) {
            asm volatile("" : "+s"(off));  // (keeps the offset 32 bits wide: widened, it costs a 64-bit add per plane and a second counter)
            // everything the test reads of the plane, fetched as ONE batch of scalar loads at the top (left to itself the
            // compiler re-loaded `occluder` and `obj` inside each branch that mentions them: four dependent scalar-cache
            // round trips per plane)
            const auto &rec_ = *(cptr<DPlane>)((unsigned long long)planes + (unsigned long long)off);
            // (the 32-byte head as ONE eight-dword scalar load: field by field the compiler asks for 2 + 4 + 2 dwords)
            typedef int i8v __attribute__((ext_vector_type(8)));
            const i8v head = *(cptr<i8v>)&rec_;
            const int pl_obj = head[0], pl_occluder = head[1];
            const mask_t pl_shadow_skip = ((mask_t)(unsigned)head[3] << 32) | (mask_t)(unsigned)head[2];
            double nk = 0, pk = 0;
            V3 pn = mk(0, 0, 0), pp = mk(0, 0, 0);
            if (AXIS < 3) { nk = __hiloint2double(head[5], head[4]); pk = __hiloint2double(head[7], head[6]); asm volatile("" :: "s"(pl_obj), "s"(pl_occluder), "s"(nk), "s"(pk), "s"(pl_shadow_skip)); }
            else { pn = ld3(rec_.n); pp = ld3(rec_.p); asm volatile("" :: "s"(pl_obj), "s"(pl_occluder), "s"(pl_shadow_skip), "s"(pn.x), "s"(pn.y), "s"(pn.z), "s"(pp.x), "s"(pp.y), "s"(pp.z)); }
            // outside the exhaustive kernels a shadow query skips Emmit objects and stops at its first occluder (the record
            // carries "not an occluder" as a ready-made mask: one OR instead of a compare and three selects)
            const mask_t m_consider = m_closest | (m_shadow_open & ~pl_shadow_skip);
            if (COUNT) cnt.c[CNT_PRIMS] += __builtin_amdgcn_inverse_ballot_w64(m_consider) ? 1 : 0;
            if (__builtin_amdgcn_inverse_ballot_w64(m_consider)) { LOOP_TICK(18); }
            double num, den;
            plane_terms<AXIS>(nk, pk, pn, pp, r, num, den);
            // a shadow query only asks whether the hit lies in front of the light: a plane met beyond K (above) is settled
            // without the division
            // (unconditional: K is infinite in rounds without such a query, and two instructions cost less than the branches)
            const mask_t m_beyond = __ballot(fabs(num) > fabs(den) * beyond_k);
            // (every ballot is of ONE comparison: the compare then writes the mask itself; a ballot of `a && b` goes through
            // a 0/1 vector register)
            const mask_t m_wanted = m_consider & ~m_beyond & quotient_may_be_positive(num, den);
            // one wave-level branch per plane: everything above is straight-line (the loop ran six conditional branches
            // per plane before, and they cost more than the arithmetic they skipped)
            if (m_wanted != 0ull) {
                if (__builtin_amdgcn_inverse_ballot_w64(m_wanted)) { LOOP_TICK(20); }
                const double t = num / den;
                const mask_t m_hit = m_consider & __ballot(fabs(den) > 0) & __ballot(t > 0);
                if (pl_occluder != 0 && (m_hit & m_shadow) != 0ull) {
                    // (in_front is true for every directional light)
                    m_occ |= m_hit & m_shadow & (~m_point | __ballot(in_front(r, t, true, light_d2)));
                    if (!EXH) m_shadow_open &= ~m_occ;
                }
                const mask_t m_better = m_hit & ~m_shadow & (__ballot(t < bt) | (__ballot(t == bt) & __ballot(pl_obj < bobj)));
                const bool better = __builtin_amdgcn_inverse_ballot_w64(m_better);
                bt = better ? t : bt; bobj = better ? pl_obj : bobj; bprim = better ? (int)(off / (unsigned)sizeof(DPlane)) : bprim;
            }
            }
        };
        const mask_t m_big = m_active & __ballot(!(magnitude < 1e30));   // (also sends these lanes' meshes to the reference tree below)
        {
            // a ray with a non-finite component (0 * inf = NaN in the reference's sums) forces the general form on everyone
            unsigned ex = (unsigned)head[8], ey = (unsigned)head[9], ez = (unsigned)head[10];   // byte offsets where the axis groups end
            if (m_big != 0ull) ex = ey = ez = 0u;
            plane_range(std::integral_constant<int, 0>(), 0u, ex);
            plane_range(std::integral_constant<int, 1>(), ex, ey);
            plane_range(std::integral_constant<int, 2>(), ey, ez);
            plane_range(std::integral_constant<int, 3>(), ez, (unsigned)n_planes * (unsigned)sizeof(DPlane));
#ifdef RAYHS_PROBE_PLANES2  // development: what the plane loops cost (they are idempotent: run them twice and compare frame times)
            { unsigned ex2 = ex, ey2 = ey, ez2 = ez; asm volatile("" : "+s"(ex2), "+s"(ey2), "+s"(ez2));
              plane_range(std::integral_constant<int, 0>(), 0u, ex2); plane_range(std::integral_constant<int, 1>(), ex2, ey2);
              plane_range(std::integral_constant<int, 2>(), ey2, ez2); plane_range(std::integral_constant<int, 3>(), ez2, (unsigned)n_planes * (unsigned)sizeof(DPlane)); }
#endif
        }
        const auto spheres = (cptr<DSphere>)(((unsigned long long)(unsigned)head[3] << 32) | (unsigned long long)(unsigned)head[2]);
        // (a round of shadow queries skips the loop when no sphere can occlude: light bulbs are Emmit)
        const mask_t m_sphere_lanes = (EXH || sphere_occluders != 0) ? m_active : m_closest;
        unsigned sph_end;   // (the loop's extent as one compare and one select -- `i < n && lanes != 0` in its condition was eleven scalar instructions per round --
                            // and in BYTES: the loop counts in the records' offset like the plane loop, one add, one compare and the load's own operand)
        asm("s_cmp_lg_u64 %1, 0\n\ts_cselect_b32 %0, %2, 0" : "=s"(sph_end) : "s"(m_sphere_lanes), "s"((unsigned)n_spheres * (unsigned)sizeof(DSphere)) : "scc");
        for (unsigned soff = 0; soff < sph_end; soff += (unsigned)sizeof(DSphere)) {
            asm volatile("" : "+s"(soff));
            const int i = (int)(soff / (unsigned)sizeof(DSphere));   // (only a hit asks for it)
            // (one batch of scalar loads per sphere, as for the planes; "not an occluder" comes as a ready-made mask)
            struct { double c[3], r2; int obj, occluder; mask_t shadow_skip; } s;
            { const auto &q = *(cptr<DSphere>)((unsigned long long)spheres + (unsigned long long)soff);
              s.c[0] = q.c[0]; s.c[1] = q.c[1]; s.c[2] = q.c[2]; s.r2 = q.r2; s.obj = q.obj; s.occluder = q.occluder; s.shadow_skip = q.shadow_skip; }
            asm volatile("" :: "s"(s.c[0]), "s"(s.c[1]), "s"(s.c[2]), "s"(s.r2), "s"(s.obj), "s"(s.occluder), "s"(s.shadow_skip));
            const mask_t m_consider = m_closest | (m_shadow_open & ~s.shadow_skip);
            if (m_consider == 0ull) continue;
            if (COUNT) cnt.c[CNT_PRIMS] += __builtin_amdgcn_inverse_ballot_w64(m_consider) ? 1 : 0;
            if (__builtin_amdgcn_inverse_ballot_w64(m_consider)) { LOOP_TICK(22); }
            double b, delta, t = 0;
            sphere_terms(s, r, a, b, delta);
            if ((m_consider & ~__ballot(delta < 0.0)) == 0ull) continue;
            const mask_t m_hit = m_consider & sphere_roots(a, b, delta, t);
            if (s.occluder != 0 && (m_hit & m_shadow) != 0ull) {
                m_occ |= m_hit & m_shadow & (~m_point | __ballot(in_front(r, t, true, light_d2)));
                if (!EXH) m_shadow_open &= ~m_occ;
            }
            const mask_t m_better = m_hit & ~m_shadow & (__ballot(t < bt) | (__ballot(t == bt) & __ballot(s.obj < bobj)));
            const bool better = __builtin_amdgcn_inverse_ballot_w64(m_better);
            bt = better ? t : bt; bobj = better ? s.obj : bobj; bprim = better ? -(i + 2) : bprim;
        }
        bool occluded = __builtin_amdgcn_inverse_ballot_w64(m_occ);
        if (n_meshes > 0) {
            // The reference's slab test multiplies by 1/d: with a zero direction component and a box plane
            // through the origin it computes 0 * inf = NaN and may reject a box the ray does touch
            // (src/KDTree.hs:39-56).  Its boxes are therefore NOT a conservative cull for such rays, and only
            // its own tree reproduces its answer: those (rare) lanes walk the reference tree, here and now.
            // A ray with a huge or non-finite component goes the same way: the fp32 coordinates of the fast structure could
            // overflow or turn into NaN for it, and its box tests rely on having neither (met()); the reference's answer for a
            // NaN ray -- every comparison false: every box passed, every triangle "hit" -- is its own tree's anyway.
            // (|d| > 1e-25 in every component also keeps 1/d finite in fp32.  All of it on wave masks: as lane predicates the four tests, the
            // `armed` flag and its ballot in traverse() went through 0/1 vector registers and back)
            const mask_t m_ref = EXH ? m_active
                                     : (m_active & (__ballot(!(fabs(r.d.x) > 1e-25)) | __ballot(!(fabs(r.d.y) > 1e-25)) | __ballot(!(fabs(r.d.z) > 1e-25)))) | m_big;
            mask_t m_occ_now = m_occ;
            if (__builtin_expect(m_ref != 0ull, 0)) {   // (rare)
                occluded = meshes_reference(__builtin_amdgcn_inverse_ballot_w64(m_ref), shadow, r, point_light, light_d2, occluded);
                m_occ_now = __ballot(occluded);
                // (the hit record leaves this rare branch through an empty asm: its outputs are new values that share registers with the
                // bypass's, so the copies between the reference walk's registers and the round's stay in here -- they were ten moves on the bypass)
                int bref_ = bref;
                asm volatile("" : "+v"(bt), "+v"(bobj), "+v"(btri), "+v"(bprim), "+v"(bref_));
                bref = bref_ != 0;
            }
            // leave the meshes to the fast traversal
            m_armed = EXH ? 0ull : (m_active & ~m_ref & ~(m_shadow & m_occ_now));
        }
        if (active) occ = occluded;
    }

    // ---- the reference's own tree, both children always (src/KDTree.hs:96-115)
    __device__ __forceinline__ bool meshes_reference(bool active, bool shadow, const Ray &r, bool point_light, double light_d2, bool occluded) {
        const V3 inv = mk(1 / r.d.x, 1 / r.d.y, 1 / r.d.z);  // src/KDTree.hs:46-48
        const int n_meshes = SC.n_meshes;
        const auto meshes = as_const(SC.meshes);
        const DNode *const nodes = SC.nodes;
        const DTri *const tris = SC.tris;
        for (int m = 0; m < n_meshes; m++) {
            const auto &me = meshes[m];
            // outside the exhaustive kernels a shadow query may stop at its first occluder and skip Emmit objects
            int node = active && (EXH || !(shadow && (occluded || !me.occluder))) ? me.root : -1;
            int sp = 0;
            while (true) {
                if (node < 0) {
                    if (sp == 0) break;
                    sp--;
                    node = (int)stack[sp * TRACE_BLOCK];
                    continue;
                }
                const DNode nd = rec(nodes, node);
                if (COUNT) cnt.c[CNT_NODES]++;
                double tmin;
                int next = -1;
                if (box_test(nd.lo, nd.hi, r, inv, tmin)) {
                    if (nd.flags & DNODE_LEAF) {
                        const int first = nd.a, last = nd.a + nd.b;
                        for (int k = first; k < last; k++) {
                            const DTri tr = rec(tris, k);
                            if (COUNT) cnt.c[CNT_TRIS]++;
                            double t, u, v;
                            if (tri_test(tr, r, t, u, v)) {
                                if (shadow) { if (me.occluder) occluded = occluded || in_front(r, t, point_light, light_d2); }
                                else if (offer(t, u, v, me.obj, k, tr.leaf_first, tr.rank)) bref = true;
                            }
                        }
                    } else {
                        int l = nd.a, rr = nd.b;
                        if (l >= 0 && rr >= 0) { stack[sp * TRACE_BLOCK] = (StackT)rr; sp++; next = l; }
                        else next = l >= 0 ? l : rr;
                    }
                }
                node = next;
            }
        }
        return occluded;
    }

    // one inner-node visit: both child boxes tested, leaves postponed (nearer first), nearer inner child next, the
    // other pushed.  ND is an FNode in registers (per-lane gather) or behind a constant-address-space reference
    // (wave-uniform visit: the fields are scalar loads and the child references stay in SGPRs).
    template <class ND> __device__ __forceinline__ void visit(Walk &w, const ND &nd, float limf) {
        float tn0, tn1;
        const bool h0 = pbox(nd.b0, w.t_ax, w.t_ay, w.t_az, w.t_bx, w.t_by, w.t_bz, limf, tn0);  // (an absent child is an empty leaf)
        const bool h1 = pbox(nd.b1, w.t_ax, w.t_ay, w.t_az, w.t_bx, w.t_by, w.t_bz, limf, tn1);
        if (COUNT) cnt.c[CNT_NODES] += (nd.c0 != FNODE_EMPTY) + (nd.c1 != FNODE_EMPTY);
        const bool near0 = tn0 <= tn1;
        const bool l0 = h0 && nd.c0 < 0, l1 = h1 && nd.c1 < 0;    // hit leaves
        const bool i0 = h0 && nd.c0 >= 0, i1 = h1 && nd.c1 >= 0;  // hit inner children
        const int cnear = near0 ? nd.c0 : nd.c1, cfar = near0 ? nd.c1 : nd.c0;
        // leaves are postponed, nearer one first
        const bool both_leaves = l0 && l1, any_leaf = l0 || l1;
        w.t_pa = both_leaves ? cnear : (l0 ? nd.c0 : (l1 ? nd.c1 : FNODE_NONE));
        w.t_pb = both_leaves ? cfar : FNODE_NONE;
        // inner children: descend into the nearer, push the other.  The push is an unconditional LDS
        // write to the top-of-stack slot (kept only when sp advances) and the pop an unconditional read,
        // so the loop body has no branches at all.
        const bool both = i0 && i1, any_inner = i0 || i1;
        const int ia = both ? cnear : (i0 ? nd.c0 : nd.c1);
        stack[w.t_sp * TRACE_BLOCK] = (StackT)cfar;
        const int top = w.t_sp > 0 ? w.t_sp - 1 : 0;
        int popped = (int)stack[top * TRACE_BLOCK];
        asm volatile("" : "+v"(popped));  // keep the read unconditional: sunk into a branch it splits the loop body
        const bool pop = !any_inner && !any_leaf && w.t_sp > 0;
        w.t_cur = any_inner ? ia : (pop ? popped : FNODE_NONE);
        w.t_sp = w.t_sp + (both ? 1 : 0) - (pop ? 1 : 0);
    }

    // ---- fast BVH: fp32 conservative boxes, fp64 triangle tests.  Meshes are taken one at a time by the whole wave
    // (the mesh record comes through the scalar cache, its object index and root stay in SGPRs); inside a mesh the
    // lanes walk independently until the last one is done.
    __device__ void traverse(bool shadow, const Ray &r, bool point_light, double light_d2, double light_dist) {
        if (EXH) return;
        // (lane yes/no values of the set-up are wave masks in scalar registers, as in begin(): as `bool`s they went through 0/1
        // vector registers and back at every use)
        typedef unsigned long long mask_t;
        if (m_armed == 0ull) return;
        const mask_t m_shadow_q = __ballot(shadow);
        // reciprocal direction in fp32 (1 ulp): every use below is a conservative cull, never a result
        const float ix = __builtin_amdgcn_rcpf((float)r.d.x), iy = __builtin_amdgcn_rcpf((float)r.d.y), iz = __builtin_amdgcn_rcpf((float)r.d.z);
        TIME_DECL;
        Walk w;
        w.stop_traversal();
        w.t_ax = w.t_ay = w.t_az = w.t_bx = w.t_by = w.t_bz = w.t_enter = 0;
        // (read field by field, through ONE laundered copy of the kernarg pointer -- each SC.x makes its own, a scalar move apiece.  Block
        // loads lose here: eight dwords of bytes [40, 72) of the scene record were 2 % SLOWER (profiles/r04/experiments/scene_head_blocks.log),
        // the ALIGNED eight of bytes [32, 64) saved 8 M scalar instructions on the dragon frame and cost +2 % of its time, +6 % on dragon_low
        // (profiles/r06/experiments/round_fixed_costs_diet.log): the loop's first branch waits for n_meshes alone)
        const auto ka_ = kargs();
        const int n_meshes = ka_->sc.n_meshes;
        const auto meshes = as_const(ka_->sc.meshes);
        const FNode *const fnodes = ka_->sc.fnodes;
        const DTri *const ftris = ka_->sc.ftris;
        for (int m = 0; m < n_meshes; m++) {  // wave-uniform
            // (what the set-up reads of the mesh record: its first 64 bytes, ONE sixteen-dword scalar load -- field by field they were five)
            struct { int obj, occluder, froot; double center[3]; float flo[3], fhi[3]; } me;
            {
                typedef int i16v __attribute__((ext_vector_type(16)));
                const i16v raw = *(cptr<i16v>)&meshes[m];
                me.obj = raw[1]; me.occluder = raw[2]; me.froot = raw[3];
                for (int k = 0; k < 3; k++) {
                    me.center[k] = __hiloint2double(raw[5 + 2 * k], raw[4 + 2 * k]);
                    me.flo[k] = __builtin_bit_cast(float, (int)raw[10 + k]); me.fhi[k] = __builtin_bit_cast(float, (int)raw[13 + k]);
                }
            }
            if (me.froot == FNODE_NONE) continue;
            // a shadow query skips Emmit objects and stops at its first occluder
            unsigned long long m_occ = __ballot(occ);
            const mask_t m_want = m_armed & ~(m_shadow_q & (me.occluder ? m_occ : ~0ull));
            if (m_want == 0ull) continue;
            const double lim = shadow ? light_dist : bt;
            // root cull in mesh-centred fp32.  The origin may be far from the mesh, so its rounding error
            // (2^-24 of its distance) is added to the box as an explicit margin.
            const float cxo = (float)(r.o.x - me.center[0]), cyo = (float)(r.o.y - me.center[1]), czo = (float)(r.o.z - me.center[2]);
            const float mg = 4.76837158e-7f * fmaxf(fmaxf(fabsf(cxo), fabsf(cyo)), fabsf(czo));  // 2^-21 * max |o - c|
            float lo[3] = {me.flo[0] - mg, me.flo[1] - mg, me.flo[2] - mg};
            float hi[3] = {me.fhi[0] + mg, me.fhi[1] + mg, me.fhi[2] + mg};
            float tn_root;
            if (COUNT && __builtin_amdgcn_inverse_ballot_w64(m_want)) cnt.c[CNT_NODES]++;
#ifdef RAYHS_PROBE_CULL2  // development: what the root cull costs (run twice)
            { float cx2 = cxo, cy2 = cyo, cz2 = czo; asm volatile("" : "+v"(cx2), "+v"(cy2), "+v"(cz2));
              const float mg2 = 4.76837158e-7f * fmaxf(fmaxf(fabsf(cx2), fabsf(cy2)), fabsf(cz2));
              float lo2[3] = {me.flo[0] - mg2, me.flo[1] - mg2, me.flo[2] - mg2}, hi2[3] = {me.fhi[0] + mg2, me.fhi[1] + mg2, me.fhi[2] + mg2};
              float tn2; mask_t e2 = m_want & fbox_mask(lo2, hi2, cx2, cy2, cz2, ix, iy, iz, f32_up(lim), tn2);
              asm volatile("" :: "s"(e2), "v"(tn2)); }
#endif
            const mask_t em = m_want & fbox_mask(lo, hi, cxo, cyo, czo, ix, iy, iz, f32_up(lim), tn_root);  // lanes that enter this mesh
            if (em == 0ull) continue;
            const bool enter = __builtin_amdgcn_inverse_ballot_w64(em);
            // shift the origin to (just before) the entry point, so that its fp32 rounding error inside the
            // traversal (2^-24 of the mesh size) is far inside the 2^-18 inflation of the boxes
            w.t_enter = tn_root > 0 ? tn_root * 0.99999905f : 0.0f;
            const double te = (double)w.t_enter;
            const float sx = (float)((r.o.x + te * r.d.x) - me.center[0]);
            const float sy = (float)((r.o.y + te * r.d.y) - me.center[1]);
            const float sz = (float)((r.o.z + te * r.d.z) - me.center[2]);
            w.t_ax = ix; w.t_ay = iy; w.t_az = iz;
            w.t_bx = -sx * ix; w.t_by = -sy * iy; w.t_bz = -sz * iz;
            float limf = f32_up(lim - te);
            const int obj = me.obj;
            w.stop_traversal();
            if (enter) { LOOP_TICK(38); }
#ifndef RAYHS_NO_PACKET
            // Coherent wave -> packet traversal.  The lanes of a wave hold samples of the same few pixels, and their rays of
            // one round are of one kind (primary, or shadow rays towards the same light from neighbouring points): when
            // all entering rays point into the same octant the wave walks the BVH as ONE packet -- one current node and one
            // stack for the wave (scalar), every node and every leaf triangle fetched once through the scalar cache, both
            // child boxes tested by all lanes, a subtree entered if ANY lane's ray meets its box, the child that more lanes
            // prefer first.  No per-lane stack traffic, no per-lane child bookkeeping (30 instead of 55 vector
            // instructions per visit), and no tail of a few long rays walking alone while the other lanes idle: the packet
            // ends when its last lane is done, after about as many visits as its longest ray needed.  Boxes only cull, the
            // triangle tests and the tie rule are the per-lane path's, so the answer is the same.  Waves whose rays
            // neither share an octant nor form a narrow cone (reflections off the model) keep the per-lane walk below.
            // (Walking such a wave as a SEQUENCE of packets, one per bundle of directions, was measured and dropped: +10 % on
            // the dragon frame, +47 % on dragon_low -- profiles/r04/experiments/packet_groups_kbench.log.)
            bool as_packet;
#ifdef RAYHS_EXP_OCTANT_SPLIT
            mask_t m_packet = 0ull;   // (the lanes that walk as the packet)
#endif
            {
                // (the octant signs and the cone are only looked at by waves that enter: kept out of the code every round runs)
                float dx = (float)r.d.x, dy = (float)r.d.y, dz = (float)r.d.z;
                asm volatile("" : "+v"(dx), "+v"(dy), "+v"(dz));
                const mask_t nx = em & __ballot(dx < 0.0f), ny = em & __ballot(dy < 0.0f), nz = em & __ballot(dz < 0.0f);
                const bool same_octant = (nx == 0ull || nx == em) && (ny == 0ull || ny == em) && (nz == 0ull || nz == em);
                // ... or into a narrow cone (within ~25 degrees) around the first entering ray's direction: shadow rays
                // converging on a near light straddle an axis without being any less of a bundle
                const int lead_lane = __ffsll((unsigned long long)em) - 1;
                const float lx = __shfl(dx, lead_lane), ly = __shfl(dy, lead_lane), lz = __shfl(dz, lead_lane);
                const float cosine = (dx * lx + dy * ly + dz * lz) * __builtin_amdgcn_rsqf((dx * dx + dy * dy + dz * dz) * (lx * lx + ly * ly + lz * lz));
                as_packet = same_octant || (em & ~__ballot(cosine > 0.9f)) == 0ull;
#ifdef RAYHS_EXP_OCTANT_SPLIT   // experiment: a wave that is no bundle walks its LARGEST octant group as one packet and only the rest per lane
                if (!as_packet) {
                    mask_t best = 0ull; int best_n = 0;
                    for (int o = 0; o < 8; o++) {
                        const mask_t g = ((o & 1) ? nx : em & ~nx) & ((o & 2) ? ny : em & ~ny) & ((o & 4) ? nz : em & ~nz);
                        const int n = __popcll(g);
                        if (n > best_n) { best_n = n; best = g; }
                    }
                    if (best_n >= RAYHS_EXP_OCTANT_SPLIT) m_packet = best;
                }
#endif
            }
#ifdef RAYHS_PACKET_ALWAYS
            as_packet = true;
#endif
#ifdef RAYHS_EXP_OCTANT_SPLIT
            if (as_packet) m_packet = em;
            if (m_packet != 0ull) {
#else
            const mask_t m_packet = em;
            if (as_packet) {
#endif
                const stack_ptr wstack = stack - (int)(__lane_id());  // lane 0's column serves as the wave's stack
                // (lane yes/no values are wave masks in scalar registers, as in begin())
                unsigned long long m_live = m_packet;
                const unsigned long long m_shadow_lanes = m_shadow_q, m_point_lanes = __ballot(point_light);
                // (Requesting the next node's record before the leaves of the current visit are tested -- the scalar-cache round
                // trip beside the triangle tests instead of after them -- was tried: the second record's 16 scalar registers on
                // top of a triangle's 18 spilled 60 more SGPRs into vector lanes, 5.37 -> 5.90 ms.)
                int cur = me.froot, sp = 0;
                do {
                    LOOP_TICK(40);
#ifdef RAYHS_PROBE_SALU4   // development: four scalar instructions more per packet visit (+42 M per dragon frame): how does the frame's time answer? (profiles/r06/experiments/)
                    asm volatile("s_mov_b32 %0, %0\n\ts_mov_b32 %0, %0\n\ts_mov_b32 %0, %0\n\ts_mov_b32 %0, %0" : "+s"(cur));
#endif
#ifdef RAYHS_PROBE_VALU4   // ... and four vector instructions more
                    asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0\n\tv_mov_b32 %0, %0" : "+v"(limf));
#endif
#ifdef RAYHS_PROBE_NODE_FETCH2  // development: what a packet visit's node fetch costs -- a second, DEPENDENT scalar fetch (of the same line: a scalar-cache
                    // hit, ~115 cycles) in front of the real one; profiles/r06/experiments/packet_node_fetch_twice.log
                    cur += *(cptr<int>)((unsigned long long)fnodes + (unsigned long long)((unsigned)cur * (unsigned)sizeof(FNode)) + offsetof(FNode, pad_));  // (pad_ is 0)
#endif
                    // (the node as ONE sixteen-dword scalar load: left to the compiler -- which only knows the record's 16-byte alignment -- it is
                    // three, 8 + 4 + 4 dwords: two scalar-memory instructions more per visit, 21 M per dragon frame)
                    typedef int i16v __attribute__((ext_vector_type(16)));
                    const i16v raw = *(cptr<i16v>)((unsigned long long)fnodes + (unsigned long long)((unsigned)cur * (unsigned)sizeof(FNode)));
                    auto fl = [&](int k) { return __builtin_bit_cast(float, (int)raw[k]); };
                    const float nb0[6] = {fl(0), fl(1), fl(2), fl(3), fl(4), fl(5)}, nb1[6] = {fl(6), fl(7), fl(8), fl(9), fl(10), fl(11)};
                    const int c0 = raw[12], c1 = raw[13];
                    static_assert(offsetof(FNode, b1) == 24 && offsetof(FNode, c0) == 48 && offsetof(FNode, c1) == 52 && sizeof(FNode) == 64, "FNode as sixteen dwords");
                    float tn0, tn1;
                    const unsigned long long b0 = pbox_mask(nb0, w.t_ax, w.t_ay, w.t_az, w.t_bx, w.t_by, w.t_bz, limf, tn0);
                    const unsigned long long b1 = pbox_mask(nb1, w.t_ax, w.t_ay, w.t_az, w.t_bx, w.t_by, w.t_bz, limf, tn1);
                    const unsigned long long m0 = b0 & m_live, m1 = b1 & m_live;  // (an absent child is an empty leaf: no triangles to test)
                    if (COUNT) cnt.c[CNT_NODES] += __builtin_amdgcn_inverse_ballot_w64(m_live) ? (c0 != FNODE_EMPTY) + (c1 != FNODE_EMPTY) : 0;
                    // the child the FIRST lane that meets anything would take first: (ca, ma) before (cb, mb).  One find-first and one
                    // bit test on the scalar unit.  (The child MOST lanes prefer -- two population counts and a compare, five scalar
                    // instructions more per visit -- tested 0.13 % fewer boxes and was no faster on the dragon frame, 1 % slower on
                    // dragon_low: profiles/r04/experiments/child_order_first_lane.log.)
                    const unsigned long long pref0 = m0 & (~m1 | __ballot(tn0 <= tn1));
                    const unsigned long long m_any = m0 | m1;
                    int ca, cb, lane1;
                    unsigned long long ma, mb;
                    asm("s_ff1_i32_b64 %4, %9\n\ts_bitcmp1_b64 %10, %4\n\ts_cselect_b32 %0, %5, %6\n\ts_cselect_b32 %1, %6, %5\n\ts_cselect_b64 %2, %7, %8\n\ts_cselect_b64 %3, %8, %7"
                        : "=&s"(ca), "=&s"(cb), "=&s"(ma), "=&s"(mb), "=&s"(lane1) : "s"(c0), "s"(c1), "s"(m0), "s"(m1), "s"(m_any), "s"(pref0) : "scc");
                    // What the visit does with its two children, on the scalar unit.  ka / kb: the child if some live lane meets its box, else
                    // "none" (one compare and one select each).  Leaves are tested at once, nearer one first; then the inner children by three
                    // selects and compares: the nearer one is walked next, the other pushed when both are inner, a pop when neither is.  The
                    // walk ends when the stack is empty (cur = none); a walk whose last live lane has just been settled empties it (below).  (As a loop over
                    // the two sides with `continue`s, a `next == none` flag and two `break`s, the compiler's structured control flow spent ~20 scalar
                    // instructions per visit on flags it set, copied and tested again: profiles/r06/isa_mix_by_section_dragon.md.)
                    int ka = ma != 0ull ? ca : FNODE_NONE, kb = mb != 0ull ? cb : FNODE_NONE;
                    auto leaf = [&](const int c, const unsigned long long m_side) {
                        // leaf: its (<= 3) triangles, each fetched once for the wave; the reference's fp64 test on the ORIGINAL ray
                        // (every per-lane decision below is a wave mask too: as nested per-lane ifs -- hit, edge band, admitted,
                        // shadow or closest, the tie chain -- each triangle walked a ladder of six exec-mask levels)
                        const unsigned long long m_mine = m_side & m_live;
                        if (m_mine == 0ull) return;
                        unsigned long long m_leaf_occ = 0;
                        const int ref = ~c, first = ref >> 3, n = ref & 7;
                        for (int k = first; k < first + n; k++) {
                            const auto &tr = crec(ftris, k);
                            if (COUNT) cnt.c[CNT_TRIS] += __builtin_amdgcn_inverse_ballot_w64(m_mine) ? 1 : 0;
                            if (__builtin_amdgcn_inverse_ballot_w64(m_mine)) { LOOP_TICK(30); }
                            double t, u, v;
                            unsigned long long m_adm = m_mine & tri_test_mask(tr, r, t, u, v);
                            if (m_adm == 0ull) continue;
                            const unsigned long long m_edge = m_adm & (__ballot(u < kEdgeBand) | __ballot(v < kEdgeBand) | __ballot((u + v) > 1 - kEdgeBand));
                            if (m_edge != 0ull) {  // (one hit in a million)
                                bool admitted = true;
                                int k_rare = k;
                                asm volatile("" : "+s"(k_rare));   // (or the loop carries k * sizeof(DTriAttr) along for this one-in-a-million path)
                                if (__builtin_amdgcn_inverse_ballot_w64(m_edge)) admitted = reference_path_admits(SC.nodes, rec(SC.fattrs, k_rare).leaf_node, r.o, r.d);
                                m_adm &= __ballot(admitted);
                            }
                            const unsigned long long m_sh = m_adm & m_shadow_lanes;
                            if (m_sh != 0ull) m_leaf_occ |= m_sh & (~m_point_lanes | __ballot(in_front(r, t, true, light_d2)));
                            const unsigned long long m_cl = m_adm & ~m_shadow_lanes;
                            if (m_cl != 0ull) {
                                // offer(): smaller t; at an exact tie the first object, the right-most leaf, the first triangle of a leaf
                                unsigned long long m_take = __ballot(t < bt);
                                const unsigned long long m_eq = m_cl & __ballot(t == bt);
                                if (m_eq != 0ull) {  // (rare: a ray through an edge or a vertex two triangles share)
                                    int bleaf = -1, brank = -1;
                                    if (__builtin_amdgcn_inverse_ballot_w64(m_eq) && obj == bobj) {
                                        const DTri &best = rec(bref ? SC.tris : SC.ftris, btri);
                                        bleaf = best.leaf_first; brank = best.rank;
                                    }
                                    m_take |= m_eq & (__ballot(obj < bobj) | (__ballot(obj == bobj) & (__ballot(tr.leaf_first > bleaf) | (__ballot(tr.leaf_first == bleaf) & __ballot(tr.rank < brank)))));
                                }
                                const bool take = __builtin_amdgcn_inverse_ballot_w64(m_take & m_cl);
                                bt = take ? t : bt; bobj = take ? obj : bobj; btri = take ? k : btri;
                                bprim = take ? -1 : bprim; bref = take ? false : bref;
                                limf = take ? f32_up(bt - (double)w.t_enter) : limf;
                            }
                        }
                        m_occ |= m_leaf_occ;
                        m_live &= ~(m_occ & m_shadow_lanes);  // any occluder settles a shadow query
                    };
                    if (max((unsigned)ka, (unsigned)kb) > (unsigned)FNODE_NONE) {   // (a leaf reference is negative and not "none"; six visits in seven have none)
                        if ((unsigned)ka > (unsigned)FNODE_NONE) leaf(ka, ma);
                        if ((unsigned)kb > (unsigned)FNODE_NONE) leaf(kb, mb);
                        // (lanes only leave the walk here -- a shadow query that found its occluder: a walk none of whose lanes is live any more ends
                        // at once.  The test used to sit at the loop's foot, two scalar instructions on every visit for what one visit in seven can change)
                        asm("s_cmp_eq_u64 %3, 0\n\ts_cselect_b32 %0, 0x80000000, %0\n\ts_cselect_b32 %1, 0x80000000, %1\n\ts_cselect_b32 %2, 0, %2"
                            : "+s"(ka), "+s"(kb), "+s"(sp) : "s"(m_live) : "scc");   // (no children, empty stack: the selects below end the walk.  A `break` made the compiler keep a flag)
                    }
                    int next = ka >= 0 ? ka : kb;
                    if ((ka | kb) >= 0) { wstack[sp * TRACE_BLOCK] = (StackT)kb; sp++; }   // both inner
                    if (next < 0) {   // neither: the walk goes on with what the stack holds, if anything
                        next = FNODE_NONE;
                        if (sp != 0) { sp--; next = __builtin_amdgcn_readfirstlane((int)wstack[sp * TRACE_BLOCK]); }
                    }
                    cur = next;
                } while (cur >= 0);
                occ = __builtin_amdgcn_inverse_ballot_w64(m_occ);
                TIME_ADD(13);
#ifdef RAYHS_EXP_OCTANT_SPLIT
                if (m_packet == em) continue;
#else
                continue;
#endif
            }
#endif
#ifdef RAYHS_EXP_OCTANT_SPLIT
            if (enter && !__builtin_amdgcn_inverse_ballot_w64(m_packet)) w.t_cur = me.froot;
#else
            if (enter) w.t_cur = me.froot;
#endif
            TIME_ADD(12);
#ifdef RAYHS_PROFILE_LOOPS
            // Census of the per-lane walk (counting kernel of the -DRAYHS_PROFILE_LOOPS build; rows 44..59 of the counters): walks and their
            // node-loop iterations by the number of lanes that enter (<= 4, <= 16, <= 32, <= 64: rows 44..47 and 48..51), and -- what a
            // re-binning of rays between the four waves of a workgroup could draw on -- how many SIBLING waves are inside a per-lane walk of
            // their own at the moment this one starts (0..3: rows 52..55) and the lanes still walking in those siblings at that moment
            // (row 56), against this walk's own entering lanes (row 57).  Sixteen bytes of LDS behind the stack area hold one word per wave:
            // the lanes it has in a per-lane walk right now (0: not walking); lane 0 of a wave is the only writer of its word.
            int census_bucket = 0;
            if (COUNT) {
                const int n_enter = __popcll(em), my_wave = (int)(threadIdx.x >> 6);
                census_bucket = n_enter <= 4 ? 0 : n_enter <= 16 ? 1 : n_enter <= 32 ? 2 : 3;
                if ((int)__lane_id() == 0) {
                    int siblings = 0; unsigned their_lanes = 0;
                    for (int k = 0; k < TRACE_BLOCK / 64; k++)
                        if (k != my_wave) { const unsigned v = census_words[k]; siblings += v != 0u; their_lanes += v; }
                    cnt.c[44 + census_bucket]++;
                    cnt.c[52 + siblings]++;
                    cnt.c[56] += their_lanes;
                    cnt.c[57] += (unsigned)n_enter;
                    census_words[my_wave] = (unsigned)n_enter;
                }
            }
#endif
            while (true) {
                // phase 1: walk inner nodes until this lane holds a leaf (or runs out of nodes).  The body is
                // branch-free apart from the stack push/pop: the whole 64-byte node is requested at once (one
                // memory round trip per visit) and both child boxes are always tested.
#ifdef RAYHS_PROFILE_LOOPS
                if (COUNT) {  // per pass of the while-while loop: lanes still in this traversal (32), of which holding a leaf (34)
                    if (w.traversing()) { LOOP_TICK(32); }
                    if (w.t_pa != FNODE_NONE) { LOOP_TICK(34); }
                    const unsigned walking = (unsigned)__popcll(__ballot(w.traversing()));
                    if ((int)__lane_id() == 0) census_words[threadIdx.x >> 6] = walking;   // (what a sibling that starts a walk now would find here)
                }
#endif
                // (a WAVE-UNIFORM loop -- while some lane still has an inner node to visit and no leaf in hand -- around a per-lane body: as a
                // per-lane while it cost ~15 instructions of exec-mask loop control per pass, about half of that as a scalar loop)
#ifdef RAYHS_EXP_UNIFORM_NODE_LOOP   // (measured: +33 M scalar-unit instructions per dragon frame against the per-lane while: profiles/r06/experiments/)
                while (true) {
                  const bool go = w.t_cur != FNODE_NONE && w.t_pa == FNODE_NONE;
                  if (__ballot(go) == 0ull) break;
                  if (go) {
#else
                while (w.t_cur != FNODE_NONE && w.t_pa == FNODE_NONE) {
                  {
#endif
                    LOOP_TICK(10);
#ifdef RAYHS_PROFILE_LOOPS
                    if (COUNT && (int)__lane_id() == __ffsll((unsigned long long)__ballot(true)) - 1) cnt.c[48 + census_bucket]++;   // node-loop iterations of walks of this size
                    if (COUNT) cnt.c[36] += (unsigned)__popcll(__ballot(true)) < 16u;   // lane-visits made in passes with < 16 lanes active
                    if (COUNT) cnt.c[37] += (unsigned)__popcll(__ballot(true)) < 32u;   // ... < 32
#endif
#ifdef RAYHS_PROFILE_LOOPS
                    if (COUNT) {  // coherence probe: how many active lanes sit on the same node as the first active lane
                        const int leader = __builtin_amdgcn_readfirstlane(w.t_cur);
                        if (w.t_cur == leader) cnt.c[14]++;
                        if (__ballot(w.t_cur != leader) == 0ull && (int)__lane_id() == __ffsll((unsigned long long)__ballot(true)) - 1) cnt.c[15]++;  // wave-uniform visits
                    }
#endif
                    // Most visits are wave-uniform: the 64 lanes hold samples of the same few pixels and walk the upper tree
                    // together (55 % of the node-loop iterations on the dragon frame, 72 % at 4K / 64 spp).  Those fetch the
                    // node ONCE through the scalar cache into SGPRs; only diverged waves gather per lane.
                    const int lead = __builtin_amdgcn_readfirstlane(w.t_cur);
                    if (__ballot(w.t_cur != lead) == 0ull) {   // (the node as ONE sixteen-dword scalar load, as in the packet walk)
                        typedef int i16v __attribute__((ext_vector_type(16)));
                        const i16v raw = *(cptr<i16v>)((unsigned long long)fnodes + (unsigned long long)((unsigned)lead * (unsigned)sizeof(FNode)));
                        struct { float b0[6], b1[6]; int c0, c1; } nd;
                        for (int k = 0; k < 6; k++) { nd.b0[k] = __builtin_bit_cast(float, (int)raw[k]); nd.b1[k] = __builtin_bit_cast(float, (int)raw[6 + k]); }
                        nd.c0 = raw[12]; nd.c1 = raw[13];
                        visit(w, nd, limf);
                    }
                    else visit(w, rec(fnodes, w.t_cur), limf);
                  }
                }
                TIME_ADD(13);
                // phase 2: the postponed leaf, with the reference's fp64 triangle test on the ORIGINAL ray
                if (w.t_pa != FNODE_NONE) {
                    LOOP_TICK(12);
                    const int ref = ~w.t_pa;
                    const int first = ref >> 3, n = ref & 7;  // leaf references index the BVH-ordered arrays directly
                    for (int k = first; k < first + n; k++) {
                        const DTri tr = rec(ftris, k);
                        if (COUNT) cnt.c[CNT_TRIS]++;
                        LOOP_TICK(30);
                        double t, u, v;
                        if (tri_test<true>(tr, r, t, u, v)) {
                            bool admitted = true;
                            if (u < kEdgeBand || v < kEdgeBand || (u + v) > 1 - kEdgeBand)
                                admitted = reference_path_admits(SC.nodes, rec(SC.fattrs, k).leaf_node, r.o, r.d);
                            if (admitted) {
                                if (shadow) occ = occ || in_front(r, t, point_light, light_d2);
                                else if (offer(t, u, v, obj, k, tr.leaf_first, tr.rank)) { bref = false; limf = f32_up(bt - (double)w.t_enter); }
                            }
                        }
                    }
                    w.t_pa = w.t_pb;
                    w.t_pb = FNODE_NONE;
                    if (shadow && occ) w.stop_traversal();  // any occluder settles a shadow query
                }
                if (w.t_pa == FNODE_NONE && w.t_cur == FNODE_NONE && w.t_sp > 0) {
                    w.t_sp--;
                    w.t_cur = (int)stack[w.t_sp * TRACE_BLOCK];
                }
                TIME_ADD(14);
                if (__ballot(w.traversing()) == 0ull) break;
            }
#ifdef RAYHS_PROFILE_LOOPS
            if (COUNT && (int)__lane_id() == 0) census_words[threadIdx.x >> 6] = 0u;
#endif
        }
    }
};

// The two triangle tables (BVH order, reference order) as scalar values; which one a lane reads is then a select between two SGPR
// pairs.  (Written as `bref ? SC.tris : SC.ftris` the compiler selected between the two kernarg ADDRESSES per lane and loaded
// the pointer itself through the vector memory path: one more dependent memory round trip in front of every triangle hit.)
template <class T> DEV const T *scalar_value(const T *p) { asm volatile("" : "+s"(p)); return p; }

// ---- global per-lane workspace: LANE_SLOTS lane slots, then FRAME_SLOTS per continuation level ----
enum : int { LS_V = 0, LS_TERM = 3 };
enum : int { FS_KIND = 0, FS_F = 2, FS_S = 3, FS_K = 4, FS_A = 5, FS_RO = 8, FS_RD = 11 };
// slot-major: slot s of lane g sits at frames[s * stride + g]; addressed as uniform base + 32-bit byte offset
struct Workspace { double *base; unsigned stride, lane; };
DEV double *ws_at(const Workspace &ws, int slot) {
    unsigned lane = ws.lane;
    asm volatile("" : "+v"(lane));  // recompute the offset at every use: hoisted out of the main loop, one VGPR per slot would be spilled to scratch
    return (double *)((char *)ws.base + ((__umul24((unsigned)slot, ws.stride) + lane) << 3));
}
#ifdef RAYHS_PROBE_WS_TWICE  // development: every workspace store is made twice (a second copy behind the first): what do the HBM writes cost?
DEV void ws_store(const Workspace &ws, int slot, double v) { *ws_at(ws, slot) = v; *ws_at(ws, slot + LANE_SLOTS + 3 * FRAME_SLOTS) = v; }
#else
DEV void ws_store(const Workspace &ws, int slot, double v) { *ws_at(ws, slot) = v; }
#endif
DEV double ws_load(const Workspace &ws, int slot) { return *ws_at(ws, slot); }
DEV void ws_store3(const Workspace &ws, int slot, V3 v) { ws_store(ws, slot, v.x); ws_store(ws, slot + 1, v.y); ws_store(ws, slot + 2, v.z); }
DEV V3 ws_load3(const Workspace &ws, int slot) { return mk(ws_load(ws, slot), ws_load(ws, slot + 1), ws_load(ws, slot + 2)); }
DEV int fslot(int level, int s) { return LANE_SLOTS + level * FRAME_SLOTS + s; }

// ---- LDS context slots (12 doubles per lane) ----
enum : int { CX_P = 0, CX_N = 3, CX_CD = 6, CX_A = 9, CX_TERM = 12, CX_V = 15 };
constexpr bool kViewInLds = CTX_SLOTS >= 18;  // (development option; the light term moves to LDS at run time, DParams.ctx_slots)
DEV void cx_store3(lds_dptr cx, int slot, V3 v) { cx[slot * TRACE_BLOCK] = v.x; cx[(slot + 1) * TRACE_BLOCK] = v.y; cx[(slot + 2) * TRACE_BLOCK] = v.z; }
DEV V3 cx_load3(lds_dptr cx, int slot) { return mk(cx[slot * TRACE_BLOCK], cx[(slot + 1) * TRACE_BLOCK], cx[(slot + 2) * TRACE_BLOCK]); }

#define PARK_V(v) do { if (kViewInLds) cx_store3(cx, CX_V, v); else ws_store3(ws, LS_V, v); } while (0)
#define LOAD_V() (kViewInLds ? cx_load3(cx, CX_V) : ws_load3(ws, LS_V))
#define PARK_TERM(v) do { if (term_in_lds) cx_store3(cx, CX_TERM, v); else ws_store3(ws, LS_TERM, v); } while (0)
#define LOAD_TERM() (term_in_lds ? cx_load3(cx, CX_TERM) : ws_load3(ws, LS_TERM))

// always inlined into its kernel: the body reads the kernarg segment of the kernel it belongs to (kargs()), which an
// out-of-line copy shared by two kernels of one instantiation cannot
template <bool EXH, bool COUNT, typename StackT>
static __device__ __forceinline__ void trace_body() {
    extern __shared__ double lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int gtid = blockIdx.x * TRACE_BLOCK + tid;
    Counters cnt;
    for (int i = 0; i < kCounterSlots; i++) cnt.c[i] = 0;
    lds_dptr cx = (lds_dptr)lds + tid;                                                  // context column
    typedef StackT __attribute__((address_space(3))) *stack_ptr;
    const int ctx_slots = PRM.ctx_slots;
    stack_ptr stk = (stack_ptr)((lds_dptr)lds + ctx_slots * TRACE_BLOCK) + tid;          // stack column
    const bool term_in_lds = ctx_slots >= 15;  // wave-uniform: the launcher grants 3 more slots when LDS has room
    Tracer<EXH, COUNT, StackT> tr(stk, cnt);
#ifdef RAYHS_PROFILE_LOOPS
    // (the launcher of a -DRAYHS_PROFILE_LOOPS build adds 16 bytes behind the stack area)
    tr.census_words = (volatile unsigned __attribute__((address_space(3))) *)((stack_ptr)((lds_dptr)lds + ctx_slots * TRACE_BLOCK) + PRM.stack_cap * TRACE_BLOCK);
    if (tid < TRACE_BLOCK / 64) tr.census_words[tid] = 0u;
    __syncthreads();
#endif
    const Workspace ws = {kargs()->frames, (unsigned)PRM.frame_stride, (unsigned)gtid};

    // per-lane state that stays in registers across a scene query.  Which pixel and sample a lane works on is NOT
    // among it: a wave always holds one whole chunk of 64 consecutive work items, lane i the i-th, so pixel, row and
    // sample follow from the chunk's first item (SGPRs) and the lane number whenever they are needed (locate()).
    int state = ST_DONE;
    int fsp = 0;  // suspended irradiance calls = recursion depth of the ray in flight (every frame pushed deepens it by one)
    Ray ray; ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1);
    int qkind = Q_NONE;
    double q_d2 = 0, q_dist = 0;
    int c_obj = 0, c_li = 0;  // object being shaded (its material kind in bits 28..30: the light loop's end asks for it again) and light index

    // Work items are (pixel, sample) pairs: item / spp is a pixel slot in 8x8-tile order (slot >> 6 = tile,
    // slot & 63 = Morton position in the tile), item % spp the sample.  A wave draws a chunk of 64 consecutive items
    // (64-aligned: the samples of the same few pixels of ONE tile) from a ticket counter, works on them in lock-step
    // and draws the next chunk when all 64 are done.  (Per-pixel items left the frame waiting on single lanes that held
    // an expensive pixel, 16 samples in sequence; per-lane refill loses the lock-step: 16.2 ms against 11.3 when tried.)
    constexpr unsigned kChunk = 64;
    unsigned long long chunk_item0 = 0;  // first item of the chunk this wave holds
    bool exhausted = false;
    // XCD-aware work distribution: the item space is cut into 8 contiguous ranges (bands of tiles), each with
    // its own ticket counter on its own 128-byte L2 line (TICKET_STRIDE).  A wave starts on the range of the XCD it runs on
    // (HW_REG_XCC_ID), so one XCD's L2 mostly sees one part of the frame and of the BVH; when its range is
    // used up it moves on to the next one, so the split only steers locality, never the balance.
    constexpr int kRanges = 8;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
    int range = (int)(xcc & (kRanges - 1)), ranges_done = 0;
    // In-wave resolve (prm.wave_resolve: spp divides 64): the samples of a pixel sit in consecutive lanes of the chunk,
    // finished samples wait in the lane's LDS context slot, and when the whole chunk is done the lane holding sample 0
    // adds them up in sample order (the reference's foldl) and writes the pixel -- no sample buffer, no resolve pass.
    bool chunk_pending = false, item_valid = false;
    unsigned long long chunk_t0 = 0;  // when the current chunk was obtained, and the tile it belongs to (a chunk never straddles tiles)
    int chunk_tile = 0;
    unsigned long long ticket_ahead = 0;  // lane 0: ticket of `range` requested ahead of need
    // The FIRST chunk of a wave is not drawn but follows from where the wave sits in the grid: band = workgroup mod 8, place in
    // the band = (workgroup / 8) * 4 + wave.  The counters then only hand out what lies behind those places: a ticket c of band r
    // stands for item c + pre_items(r), pre_items(r) = 256 * (workgroups b < grid with b mod 8 == r).  At the start of a launch
    // every wave of the chip used to queue at eight counters for its first ticket (operations on one line are served ~10 ns
    // apart: 512 waves per line) before it could generate a single ray.
    auto pre_items = [&](int r) { return (unsigned long long)(((unsigned)gridDim.x + 7u - (unsigned)r) >> 3) * (unsigned long long)TRACE_BLOCK; };
    const int xcc_range = range;
    range = (int)(blockIdx.x & (kRanges - 1));
    ticket_ahead = (unsigned long long)(((unsigned)blockIdx.x >> 3) * (TRACE_BLOCK / 64) + ((unsigned)tid >> 6)) * kChunk - pre_items(range);
    bool first_chunk = true;

    // this lane's pixel (column, local row) and sample within the chunk the wave holds
    auto locate = [&](const auto &prm, int &px, int &lrow, int &sample) {
        const unsigned long long item = chunk_item0 + (unsigned)lane;
        // (this ran 518 k times per frame in every lane: shifts and a corrected float reciprocal where the operands allow it)
        const unsigned slot = prm.spp_shift >= 0 ? (unsigned)(item >> prm.spp_shift) : (unsigned)(item / (unsigned)prm.spp);
        sample = (int)(item - (unsigned long long)slot * (unsigned)prm.spp);
        const int tile = chunk_tile, within = (int)(slot & 63u);
        int ty = (int)((float)tile * prm.inv_tiles_x), tx = tile - ty * prm.tiles_x;  // tile < 2^24: off by one at most
        if (tx < 0) { ty--; tx += prm.tiles_x; } else if (tx >= prm.tiles_x) { ty++; tx -= prm.tiles_x; }
        // Morton order inside the tile: consecutive slots form 2x2, 4x4, ... blocks, so the few pixels a
        // wave holds at once are neighbours in both directions
        px = tx * 8 + ((within & 1) | ((within >> 1) & 2) | ((within >> 2) & 4));
        lrow = ty * 8 + (((within >> 1) & 1) | ((within >> 2) & 2) | ((within >> 3) & 4));
    };

    TIME_DECL;
#ifdef RAYHS_PROFILE_TIME
    const unsigned long long rt0_ = __builtin_amdgcn_s_memrealtime(), mt0_ = __builtin_amdgcn_s_memtime();
#elif defined(RAYHS_WAVE_ENDS)
    const unsigned long long wave_t0_ = __builtin_amdgcn_s_memrealtime();
#endif
    while (true) {
        bool refilled = false;  // (wave-uniform) this round's rays are the primary rays of a chunk just drawn
        qkind = Q_NONE;
        if (__ballot(state != ST_DONE) == 0ull) {  // every lane is done: finish the chunk the wave held, draw the next one
            const ParamsHot prm = params_hot();
            const PointersHot ptrs = pointers_hot();
            const auto &prm_rest = kargs()->prm;
            if (chunk_pending) {
                if (prm.wave_resolve && item_valid) {
                    int px, lrow, sample;
                    locate(prm, px, lrow, sample);
#ifndef RAYHS_NO_CHANNEL_RESOLVE
                    if (prm.spp >= 4) {
                        // the pixel's first three lanes take one colour channel each: a third of the additions and of the LDS reads of one
                        // lane summing all three (every instruction of the sum is issued for the whole wave whichever lanes want it)
                        if (sample < 3) {
                            const auto *const col = cx + (CX_A + sample) * TRACE_BLOCK - sample;   // sample k of this lane's pixel and channel: col[k]
                            double sum = 0;
#pragma unroll 4
                            for (int k = 0; k + 4 <= prm.spp; k += 4) {  // foldl (+) black, samples in order (src/RayHs.hs:169-171)
                                const double a = col[k], b = col[k + 1], c = col[k + 2], d = col[k + 3];   // four at a time: one LDS wait, same order of additions
                                sum = (((sum + a) + b) + c) + d;
                            }
                            store_channel(ptrs.out, prm.out_format, ((size_t)lrow * (size_t)prm.width + (size_t)px) * 3 + (size_t)sample, prm.inv_spp * sum);
                        }
                    } else
#endif
                    if (sample == 0) {
                        V3 sum = mk(0, 0, 0);
                        auto at = [&](int k) { return mk(cx[CX_A * TRACE_BLOCK + k], cx[(CX_A + 1) * TRACE_BLOCK + k], cx[(CX_A + 2) * TRACE_BLOCK + k]); };
                        int k = 0;
                        for (; k + 4 <= prm.spp; k += 4) {  // foldl (+) black, samples in order (src/RayHs.hs:169-171);
                            const V3 a = at(k), b = at(k + 1), c = at(k + 2), d = at(k + 3);  // four at a time: one LDS wait, same order of additions
                            sum = add(add(add(add(sum, a), b), c), d);
                        }
                        for (; k < prm.spp; k++) sum = add(sum, at(k));
                        store_pixel(ptrs.out, prm.out_format, (size_t)lrow * (size_t)prm.width + (size_t)px, scl(prm.inv_spp, sum));
                    }
                }
                // how long the chunk took, for the next frame's work order (costs are >= 0: their bit patterns order like the values)
                if (ptrs.tile_cost != nullptr && lane == 0)
                    atomicMax((unsigned *)(ptrs.tile_cost + chunk_tile), __float_as_uint((float)(__builtin_amdgcn_s_memrealtime() - chunk_t0)));
                chunk_pending = false;
                item_valid = false;
            }
            if (exhausted) break;
            {
                unsigned long long *const ticket = ptrs.ticket;
                const unsigned long long total_items = (unsigned long long)prm.n_tiles * 64ull * (unsigned long long)prm.spp;
                const unsigned long long per_range = ((total_items + (unsigned long long)kRanges * kChunk - 1) / ((unsigned long long)kRanges * kChunk)) * kChunk;
                // the ticket for this chunk was requested one refill ago; the next one is requested below, so that
                // the atomic's round trip (microseconds) runs beside the chunk's work instead of in front of it
                unsigned long long c = ticket_ahead;
                while (true) {  // wave-uniform; more than one pass only when a range runs dry (<= 8 times per wave)
                    c = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(c >> 32)) << 32) |
                        (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)c);
                    const unsigned long long start = (unsigned long long)range * per_range;
                    const unsigned long long size = start >= total_items ? 0ull : (total_items - start < per_range ? total_items - start : per_range);
                    c += pre_items(range);
                    if (c < size) { chunk_item0 = start + c; break; }  // sizes are multiples of 64: the chunk is whole
                    // This range is dry.  Before walking on to the next one, note the fact in a line of eight "this range is dry"
                    // flags behind the counters and look at the other seven.  The line is kept per XCD: written and read at
                    // WORKGROUP scope, i.e. through this XCD's L2 and no further.  A flag only ever goes from 0 to 1 and is set by a
                    // wave that has seen the range dry, so any view of the line -- however stale -- that shows all eight set means
                    // the frame is done, and the wave leaves without touching a counter; any other view only costs the walk.
                    // Per XCD the first waves to run dry walk the ranges (up to eight dependent atomic round trips on lines that
                    // every wave of the chip is drawing from: ~30 us) and set the flags; the ~500 that finish after them leave at
                    // once.  The walk behind EVERY wave's last chunk was 13 % of the 0.23 ms Cornell frame (-26 %; dragon_low -3.5 %,
                    // dragon -0.8 %: profiles/r04/experiments/dry_range_flags.log).  Measured and dropped on the way there: looking at
                    // the eight COUNTERS instead (thousands of device-scope loads on the very lines the tickets are drawn from: the
                    // waves still working wait 25-50 us for theirs at the end of a dragon frame, +2 %), flags at device scope (every
                    // access goes to the memory side, ~10 ns apart per line: +45 % on the Cornell frame), tickets good for four or
                    // eight chunks (a tile inside the model then runs as one chain: dragon_low +45 %).
                    {
                        unsigned long long *const dry_flags = ticket + kRanges * TICKET_STRIDE;  // (a 128-byte line of their own)
                        if (lane == 0) __hip_atomic_store(dry_flags + range, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        unsigned long long flag = 1ull;
                        if (lane < kRanges && lane != range) flag = __hip_atomic_load(dry_flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (__ballot(flag == 0ull) == 0ull) { exhausted = true; break; }
                    }
                    if (++ranges_done == kRanges) { exhausted = true; break; }
                    range = (range + 1) & (kRanges - 1);
                    if (lane == 0) c = atomicAdd(ticket + range * TICKET_STRIDE, (unsigned long long)kChunk);
                }
                if (exhausted) break;
                if (first_chunk) { first_chunk = false; if (ranges_done == 0) range = xcc_range; }  // from here on: the band of the XCD this wave runs on
                if (lane == 0) ticket_ahead = atomicAdd(ticket + range * TICKET_STRIDE, (unsigned long long)kChunk);
                // a chunk (64 consecutive items, 64-aligned) never straddles tiles: its tile is wave-uniform and comes
                // through the scalar cache, in the order the previous frame measured, slowest first (rayhs_order_tiles)
                const unsigned tslot = prm.spp_shift >= 0 ? (unsigned)(chunk_item0 >> (prm.spp_shift + 6)) : (unsigned)(chunk_item0 / ((unsigned long long)prm.spp * 64ull));
                const int entry = ptrs.tile_order ? as_const(ptrs.tile_order)[tslot] : (int)tslot;
                chunk_tile = entry & kTileMask;
                if (ptrs.tile_cost != nullptr) {
                    chunk_t0 = __builtin_amdgcn_s_memrealtime();
                    // the slower the tile was last time, the higher this wave's issue priority while it works on it: a
                    // chunk is a chain of dependent queries, and when a frame is small (one GPU's share of eight) the
                    // longest chain, not the total work, decides when the kernel ends (-7 % on a 1/8 share)
#ifndef RAYHS_NO_SETPRIO
                    switch ((unsigned)entry >> kTileBits) {
                    case 0: __builtin_amdgcn_s_setprio(0); break;
                    case 1: __builtin_amdgcn_s_setprio(1); break;
                    case 2: __builtin_amdgcn_s_setprio(2); break;
                    default: __builtin_amdgcn_s_setprio(3); break;
                    }
#endif
                }
#ifdef RAYHS_PROFILE_TIME
                cnt.c[7]++;  // chunks this wave obtained (lane-0 value is reduced)
                if (cnt.c[6] == 0) cnt.c[6] = 1;  // waves that obtained any work
#endif
                int px, lrow, sample;
                locate(prm, px, lrow, sample);
                item_valid = px < prm.width && lrow < prm.n_rows;
                chunk_pending = true;
                // Every lane of a fresh chunk starts its sample here and now (a lane only ever starts one right after a refill):
                // the item was located a moment ago, and the rounds in between never ask "is this lane starting a sample?"
                if (item_valid) {
                    // pixelCoord (src/Image.hs:31-32) + optional sample offset (src/RayHs.hs:178-188)
                    int yrow;
                    // local row -> image row (stripe interleave, include/rayhs.h)
                    if (prm.stripe_count == 1 && ptrs.stripe_map == nullptr) yrow = prm.row0 + lrow;
                    else {
                        const int s_local = prm.stripe_shift >= 0 ? lrow >> prm.stripe_shift : lrow / prm_rest.stripe_rows;
                        const int in_stripe = lrow - s_local * prm_rest.stripe_rows;
                        // round-robin ownership, or an explicit list of this device's stripes (rayhs_multi: cost-balanced)
                        const int s_global = ptrs.stripe_map ? ptrs.stripe_map[s_local] : s_local * prm.stripe_count + prm_rest.stripe_rank;
                        yrow = prm.row0 + s_global * prm_rest.stripe_rows + in_stripe;
                    }
                    double fx = (double)px, fy = (double)yrow;
                    if (prm.spp > 1) {
                        const uint64_t seed_mixed = prm.seed;  // (mix64(seed + golden ratio), done by the launcher)
                        uint64_t pixel = (uint64_t)px + (uint64_t)prm.width * (uint64_t)yrow;
                        fx = fx + (sample01(seed_mixed, pixel, (uint32_t)sample, 0, (uint32_t)prm.spp) - 0.5);
                        fy = fy + (sample01(seed_mixed, pixel, (uint32_t)sample, 1, (uint32_t)prm.spp) - 0.5);
                    }
                    ray = ray_from_pixel(kargs()->cam, fx, fy);
                    fsp = 0;
                    if (COUNT) cnt.c[CNT_PRIMARY]++;
                    qkind = Q_CLOSEST; state = ST_HIT;
                }
                refilled = true;
            }
        }

        // ---- advance: consume the last query's answer, run shading up to the next query.
        // Straight-line sections (hit / shadow / exit -> lights -> specular -> return).  The sections talk to
        // each other through the lane's LDS context slots (P, N, CD, A -- A doubles as the radiance being returned), not
        // through registers: every V3 that crossed a section boundary in registers had to be copied into place on each
        // path that reaches the join, and those copies were a quarter of the vector instructions of a round.  The
        // compiler barriers between the sections keep it from turning the LDS hand-over back into register copies.
        TIME_ADD(0);  // refill
        if (!refilled && state != ST_DONE) {
            LOOP_TICK(16);
            enum : int { A_NONE, A_LIGHT_NEXT, A_SPEC, A_RETURN, A_QUERY_READY };
            int step = A_NONE;

            if (state == ST_HIT) {
                LOOP_TICK(24);
                // traceRay (src/RayHs.hs:149-154)
                if (tr.bobj < 0) { cx_store3(cx, CX_A, mk(0, 0, 0)); step = A_RETURN; }
                else {
                    const DObject *const objects = SC.objects;
                    // (what every material reads of the object; ior of a Transparent one is fetched where it is used)
                    struct { int mat_kind, cmap_kind; double c1[3]; } ob;
                    { const DObject &q = rec(objects, tr.bobj); ob.mat_kind = q.mat_kind; ob.cmap_kind = q.cmap_kind; ob.c1[0] = q.c1[0]; ob.c1[1] = q.c1[1]; ob.c1[2] = q.c1[2]; }
                    // Everything a triangle hit reads -- the object record above, the triangle, its normals -- is requested in ONE
                    // batch, in front of the first store of the section (a load cannot be moved above a store that might alias it).
                    // Left to itself the section made four DEPENDENT vector-memory round trips: the two array bases chosen per
                    // lane (reference order or BVH order: a per-lane pointer, loaded per lane), the triangle, then the normals.  Hits
                    // found by the reference tree (rays with a zero direction component) are rare: the wave asks once whether it holds any.
                    const bool is_tri = tr.btri >= 0;
                    const bool some_ref = EXH || __ballot(is_tri && tr.bref) != 0ull;
                    const DTri *t_arr = scalar_value(SC.ftris);
                    const DTriAttr *a_arr = scalar_value(SC.fattrs);
                    if (some_ref) {  // (wave-uniform, rare)
                        const DTri *const t_ref = scalar_value(SC.tris);
                        const DTriAttr *const a_ref = scalar_value(SC.attrs);
                        t_arr = tr.bref ? t_ref : t_arr; a_arr = tr.bref ? a_ref : a_arr;
                    }
                    // (three vectors, not a struct of arrays: that one lived in scratch, every 16-byte piece of the triangle stored
                    // there behind its own wait, the loads one after the other)
                    V3 h_p0 = mk(0, 0, 0), h_e1 = h_p0, h_e2 = h_p0;
                    V3 h_n0 = mk(0, 0, 0), h_n1 = h_n0, h_n2 = h_n0;
                    if (is_tri) {
                        const DTri &q = rec(t_arr, tr.btri);
                        h_p0 = ld3(q.p0); h_e1 = ld3(q.e1); h_e2 = ld3(q.e2);
                        const DTriAttr &at = rec(a_arr, tr.btri);
                        h_n0 = ld3(at.n0); h_n1 = ld3(at.n1); h_n2 = ld3(at.n2);
                    } else if (tr.bprim >= 0) h_n0 = ld3(rec(SC.planes, tr.bprim).n);   // a plane's normal,
                    else h_n0 = ld3(rec(SC.spheres, -tr.bprim - 2).c);                  // a sphere's centre: the same round trip
                    asm volatile("" ::: "memory");
                    if (COUNT) cnt.c[CNT_SHADED]++;
                    c_obj = tr.bobj | (ob.mat_kind << 28);  // (abi.cpp: fewer than 2^25 objects)
                    const V3 c_p = add(ray.o, scl(tr.bt, ray.d));  // rayAt (src/Geometry.hs:27-29)
                    V3 c_n;
                    double hu = 0, hv = 0;
                    const int mat_kind = ob.mat_kind;
                    // (a flat colour goes to its context slot at once: six registers less across the triangle test below)
                    if ((mat_kind == 1 || mat_kind == 2) && ob.cmap_kind == 0) cx_store3(cx, CX_CD, ld3(ob.c1));
                    if (mat_kind == 0 || mat_kind == 2 || mat_kind == 4) PARK_V(ray.d);  // Mirror, Plastic, Transparent: the view direction is read again after the light loop
                    const bool need_uv = ((mat_kind == 1 || mat_kind == 2) && ob.cmap_kind != 0) || mat_kind == 6;
                    if (is_tri) {
                        double u, v, t_again;  // the same test on the same ray and triangle: the same doubles as during the walk
                        tr.tri_test_v(h_p0, h_e1, h_e2, ray, t_again, u, v);
                        const double w = 1 - u - v;
                        // barycentricInterp u n1 v n2 (1-u-v) n0 (src/Mesh.hs:53-57,73-74)
                        c_n = add(add(scl(u, h_n1), scl(v, h_n2)), scl(w, h_n0));
                        if (need_uv) {
                            const DTriAttr &at = rec(a_arr, tr.btri);
                            hu = u * at.uv1[0] + v * at.uv2[0] + w * at.uv0[0];
                            hv = u * at.uv1[1] + v * at.uv2[1] + w * at.uv0[1];
                        }
                    } else if (tr.bprim >= 0) {  // re-derive the analytic hit from t
                        c_n = h_n0;
                        if (need_uv) { const auto &pl = rec(SC.planes, tr.bprim); V3 rel = sub(c_p, ld3(pl.p)); hu = dot(ld3(pl.t), rel); hv = dot(ld3(pl.b), rel); }
                    } else {
                        c_n = unit(sub(c_p, h_n0));
                        if (need_uv) { const UV2 uv = polar_uv(c_n); hu = uv.u; hv = uv.v; }  // src/Geometry.hs:96
                    }
                    // the context of the hit being shaded waits in LDS while its rays are traced
                    cx_store3(cx, CX_P, c_p); cx_store3(cx, CX_N, c_n);
                    // irradiance (src/RayHs.hs:107-147)
                    if (mat_kind == 1 || mat_kind == 2) {  // Diffuse, Plastic
                        if (ob.cmap_kind != 0) cx_store3(cx, CX_CD, color_at_slow(as_const(SC.objects), tr.bobj, SC.texels, hu, hv));
                        cx_store3(cx, CX_A, mk(0, 0, 0));  // accumDiffuse folds from black
                        c_li = 0;
                        step = A_LIGHT_NEXT;
                    } else if (mat_kind == 0) {  // Mirror
                        step = A_SPEC;
                    } else if (mat_kind == 3) {  // Emmit
                        cx_store3(cx, CX_A, ld3(rec(objects, tr.bobj).ce)); step = A_RETURN;
                    } else if (mat_kind == 4) {  // Transparent: try to transmit (src/RayHs.hs:136-143)
                        V3 refdir;
                        if (fsp != PRM.max_depth && refract(ray.d, c_n, 1.0, rec(objects, tr.bobj).ior, refdir)) {
                            cx_store3(cx, CX_CD, refdir);  // keep refDir across the exit query
                            ray.o = add(c_p, scl(kEps, refdir)); ray.d = refdir;  // rayEps
                            if (COUNT) cnt.c[CNT_REFRACT_EXIT]++;
                            qkind = Q_CLOSEST; state = ST_EXIT; step = A_QUERY_READY;
                        } else step = A_SPEC;
                    } else if (mat_kind == 5) { cx_store3(cx, CX_A, c_n); step = A_RETURN; }   // ShowNormal
                    else { cx_store3(cx, CX_A, mk(hu, hv, 0)); step = A_RETURN; }              // ShowUV
                }
            } else if (state == ST_SHADOW) {
                // accumDiffuse fold step (src/RayHs.hs:90-96)
                cx_store3(cx, CX_A, add(cx_load3(cx, CX_A), tr.occ ? mk(0, 0, 0) : LOAD_TERM()));
                c_li++;
                step = A_LIGHT_NEXT;
            } else if (state == ST_EXIT) {
                // the transmitted ray's exit point (src/RayHs.hs:140-143)
                const auto &ob = rec(SC.objects, c_obj & 0x0fffffff);
                bool have = false;
                if (tr.bobj >= 0) {
                    V3 outp = add(ray.o, scl(tr.bt, ray.d));
                    V3 outn;
                    if (tr.btri >= 0) {
                        // (one batch of loads, as in the hit section)
                        const bool some_ref = EXH || __ballot(tr.bref) != 0ull;
                        const DTri *t_arr = scalar_value(SC.ftris);
                        const DTriAttr *a_arr = scalar_value(SC.fattrs);
                        if (some_ref) {
                            const DTri *const t_ref = scalar_value(SC.tris);
                            const DTriAttr *const a_ref = scalar_value(SC.attrs);
                            t_arr = tr.bref ? t_ref : t_arr; a_arr = tr.bref ? a_ref : a_arr;
                        }
                        const DTri &h_q = rec(t_arr, tr.btri);
                        const V3 h_p0 = ld3(h_q.p0), h_e1 = ld3(h_q.e1), h_e2 = ld3(h_q.e2);
                        const DTriAttr &at = rec(a_arr, tr.btri);
                        const V3 h_n0 = ld3(at.n0), h_n1 = ld3(at.n1), h_n2 = ld3(at.n2);
                        asm volatile("" ::: "memory");
                        double u, v, t_again;
                        tr.tri_test_v(h_p0, h_e1, h_e2, ray, t_again, u, v);
                        const double w = 1 - u - v;
                        outn = add(add(scl(u, h_n1), scl(v, h_n2)), scl(w, h_n0));
                    } else if (tr.bprim >= 0) outn = ld3(rec(SC.planes, tr.bprim).n);
                    else outn = unit(sub(outp, ld3(rec(SC.spheres, -tr.bprim - 2).c)));
                    V3 outdir;
                    if (refract(cx_load3(cx, CX_CD), neg(outn), ob.ior, 1.0, outdir)) {
                        have = true;
                        // suspend this irradiance call: radiance child first, specular child later
                        const V3 c_p = cx_load3(cx, CX_P), c_n = cx_load3(cx, CX_N);
                        V3 c_v = LOAD_V();
                        V3 rf = reflect(c_v, c_n);
                        ws_store(ws, fslot(fsp, FS_KIND), (double)FR_TRANS_WAIT_RADIANCE);
                        ws_store(ws, fslot(fsp, FS_F), fresnel(ob.fr, dot(c_n, neg(c_v)), (PRM.flags_dev & 2) != 0));
                        ws_store(ws, fslot(fsp, FS_S), dot(rf, c_n));
                        ws_store(ws, fslot(fsp, FS_K), ob.one_m_r0);
                        ws_store3(ws, fslot(fsp, FS_RO), add(c_p, scl(kEps, rf)));
                        ws_store3(ws, fslot(fsp, FS_RD), rf);
                        fsp++;
                        ray.o = add(outp, scl(kEps, outdir)); ray.d = outdir;
                        if (COUNT) cnt.c[CNT_REFRACT]++;
                        qkind = Q_CLOSEST; state = ST_HIT; step = A_QUERY_READY;
                    }
                }
                if (!have) step = A_SPEC;
            }

            asm volatile("" ::: "memory");
            TIME_ADD(1);  // hit / shadow / exit sections
            // The loop is WAVE-UNIFORM (it runs while some lane still looks for a light whose term is not zero: more than one pass only then)
            // around a per-lane body.  As a per-lane do-while with three ways out it cost ~37 scalar instructions of exec-mask bookkeeping
            // per pass (the structurizer's flags for every exit), 95 M per dragon frame -- more than the loop's arithmetic issues on the
            // scalar unit (profiles/r06/isa_mix_by_section_dragon.md, segments 210 / 211 / 216 of the basic-block profile).
#ifdef RAYHS_EXP_UNIFORM_LIGHT_LOOP
            if (__ballot(step == A_LIGHT_NEXT) != 0ull) {
                // (every lane reads its slots: the values only matter in the lanes that use them, and a conditional read needs them zeroed first)
                const V3 c_p = cx_load3(cx, CX_P), c_n = cx_load3(cx, CX_N), c_cd = cx_load3(cx, CX_CD);
                const int n_lights = SC.n_lights;
                do {
                  if (step == A_LIGHT_NEXT) {
#else
            if (step == A_LIGHT_NEXT) {
                const V3 c_p = cx_load3(cx, CX_P), c_n = cx_load3(cx, CX_N), c_cd = cx_load3(cx, CX_CD);
                const int n_lights = SC.n_lights;
                do {
                  {
#endif
                    LOOP_TICK(26);
                    if (c_li < n_lights) {
                        // lightAt (src/Light.hs:12-17) + diffuse (src/Material.hs:31-33)
                        const DLight L = rec(SC.lights, c_li);  // by value: the whole 64-byte record in one batch of loads, not field by field per branch
                        V3 ld, lc;
                        const bool l_point = L.kind != 0;
                        if (!l_point) { ld = ld3(L.v); lc = ld3(L.c); q_d2 = 0; q_dist = __builtin_inf(); }
                        else {
                            V3 dv = sub(ld3(L.v), c_p);
                            const double d2 = dot(dv, dv);
                            // The root and the three divisions without their range fix-ups (rcp_in_range above: the same doubles) when every
                            // lane here has 2^-399 <= d2 < 2^399 (one subtraction and one comparison on the high word: d in [2^-200, 2^200))
                            // and a light whose radius the launcher found in [1e-60, 1e60] (DLight.short_ok): then den in [1, 2^400],
                            // den^2 < 2^801, and no operand or quotient comes near the ranges the fix-ups exist for.
                            const bool in_range = kShortDiv && ((unsigned)__double2hiint(d2) - 0x27000000u) < (0x58e00000u - 0x27000000u) && L.short_ok != 0;
                            double d, falloff, inv_d;
                            if (kShortDiv && __ballot(!in_range) == 0ull) {
                                d = sqrt_in_range(d2);
                                const double den = 1.0 + div_in_range(d, L.radius);
                                falloff = rcp_in_range(den * den);
                                inv_d = rcp_in_range(d);
                            } else {
                                d = sqrt(d2);
                                const double den = 1.0 + d / L.radius;
                                falloff = 1.0 / (den * den);
                                inv_d = 1 / d;
                            }
                            ld = scl(inv_d, dv);
                            lc = scl(falloff, ld3(L.c));
                            q_dist = d;
                        }
                        // the light's diffuse term, added after the shadow query if unoccluded
                        const V3 term = scl(hs_max(dot(ld, c_n), 0) * kPiInv, mulv(c_cd, lc));
                        // A term of +-0 in every channel (the surface faces away from the light) leaves the running sum as it
                        // is whether the light is occluded (c + black) or not (c + term): the sum starts at +0 and IEEE addition
                        // of +-0 to anything but -0 is the identity.  The shadow query cannot matter and is not traced.  (The
                        // exhaustive kernels trace it: their work counters are the reference's.)
                        if (!EXH && term.x == 0 && term.y == 0 && term.z == 0) {
                            if (COUNT) { cnt.c[CNT_SHADOW]++; cnt.c[CNT_SHADOW_SKIPPED]++; }
                            c_li++;   // (step stays A_LIGHT_NEXT: this lane takes the next pass)
                        } else {
                            PARK_TERM(term);
                            ray.o = add(c_p, scl(kEps, ld)); ray.d = ld;  // rayEps p ld
                            if (l_point) { V3 dd = sub(ray.o, ld3(L.v)); q_d2 = dot(dd, dd); }  // sqrDist origin p (src/RayHs.hs:87)
                            if (COUNT) cnt.c[CNT_SHADOW]++;
                            qkind = l_point ? Q_SHADOW_POINT : Q_SHADOW; state = ST_SHADOW; step = A_QUERY_READY;  // p, n, cd and the running sum A stay parked in LDS
                        }
                    } else if ((c_obj >> 28) == 1) {
                        cx_store3(cx, CX_A, add(scl(0.2, c_cd), cx_load3(cx, CX_A)));  // Diffuse (src/RayHs.hs:111-114)
                        step = A_RETURN;
                    } else step = A_SPEC;  // Plastic (src/RayHs.hs:117-120)
                  }
#ifdef RAYHS_EXP_UNIFORM_LIGHT_LOOP
                } while (__ballot(step == A_LIGHT_NEXT) != 0ull);
#else
                } while (step == A_LIGHT_NEXT);
#endif
            }

            asm volatile("" ::: "memory");
            TIME_ADD(2);  // light section
            if (step == A_SPEC) {
                // fresnel * specular (src/RayHs.hs:99-104,119,124,132)
                const auto &ob = rec(SC.objects, c_obj & 0x0fffffff);
                const bool spec_with_A = (c_obj >> 28) == 2;  // Plastic: the light sum in the A slot goes with the reflection (the other ways here -- Mirror, Transparent -- carry none)
                const V3 c_n = cx_load3(cx, CX_N);
                V3 c_v = LOAD_V();
                double F = fresnel(ob.fr, dot(c_n, neg(c_v)), (PRM.flags_dev & 2) != 0);
                if (fsp < PRM.max_depth) {
                    V3 rf = reflect(c_v, c_n);
                    ws_store(ws, fslot(fsp, FS_KIND), (double)(spec_with_A ? FR_PLASTIC : FR_MIRROR));
                    ws_store(ws, fslot(fsp, FS_F), F);
                    ws_store(ws, fslot(fsp, FS_S), dot(rf, c_n));
                    if (spec_with_A) ws_store3(ws, fslot(fsp, FS_A), cx_load3(cx, CX_A));
                    fsp++;
                    ray.o = add(cx_load3(cx, CX_P), scl(kEps, rf)); ray.d = rf;
                    if (COUNT) cnt.c[CNT_REFLECT]++;
                    qkind = Q_CLOSEST; state = ST_HIT; step = A_QUERY_READY;
                } else {
                    V3 spec = scl(F, mk(0, 0, 0));
                    if (spec_with_A) spec = add(cx_load3(cx, CX_A), spec);
                    cx_store3(cx, CX_A, spec);
                    step = A_RETURN;
                }
            }

            asm volatile("" ::: "memory");
            TIME_ADD(3);  // specular section
            // hand the radiance in the A slot to the suspended callers until one needs another ray, or the sample is done
            // (a wave-uniform loop around a per-lane body, like the light section's)
#ifdef RAYHS_EXP_UNIFORM_LIGHT_LOOP
            if (__ballot(step == A_RETURN) != 0ull) {
                V3 result = cx_load3(cx, CX_A);
                do {
                  if (step == A_RETURN) {
#else
            if (step == A_RETURN) {
                V3 result = cx_load3(cx, CX_A);
                do {
                  {
#endif
                    LOOP_TICK(28);
                    if (fsp == 0) {
                        const auto &prm = kargs()->prm;
                        if (prm.wave_resolve) {
                            cx_store3(cx, CX_A, result);  // waits in the A slot for the in-wave resolve (stored again when nothing was returned to: a flag saying so costs more)
                        } else {
                            int px, lrow, sample;
                            locate(prm, px, lrow, sample);
                            const size_t pix = (size_t)lrow * (size_t)prm.width + (size_t)px;
                            if (prm.spp > 1) {
                                // one sample of distributedTracePixel (src/RayHs.hs:173-183); rayhs_resolve folds them in order
                                double *sb = prm.sbuf + (pix * (size_t)prm.spp + (size_t)sample) * 3;
                                sb[0] = result.x; sb[1] = result.y; sb[2] = result.z;
                            } else {
                                store_pixel(kargs()->out, prm.out_format, pix, result);
                            }
                        }
                        if (COUNT) cnt.c[CNT_PIXELS]++;
                        state = ST_DONE;
                        step = A_NONE;
                    } else {
                        int lv = fsp - 1;
                        // one batch of loads per level (a single memory round trip): what every kind but MIRROR reads
                        const int kind = (int)ws_load(ws, fslot(lv, FS_KIND));
                        const double F = ws_load(ws, fslot(lv, FS_F)), S = ws_load(ws, fslot(lv, FS_S));
                        const V3 fA = ws_load3(ws, fslot(lv, FS_A));
                        if (kind == FR_PLASTIC) {
                            result = add(fA, scl(F, scl(S, result)));
                            fsp = lv;
                        } else if (kind == FR_MIRROR) {
                            result = scl(F, scl(S, result));
                            fsp = lv;
                        } else if (kind == FR_TRANS_WAIT_RADIANCE) {
                            ws_store3(ws, fslot(lv, FS_A), result);
                            ws_store(ws, fslot(lv, FS_KIND), (double)FR_TRANS_WAIT_SPEC);
                            ray.o = ws_load3(ws, fslot(lv, FS_RO));
                            ray.d = ws_load3(ws, fslot(lv, FS_RD));
                            // (the frame stays suspended: the reflected ray runs at this frame's depth + 1 = fsp)
                            if (COUNT) cnt.c[CNT_REFLECT]++;
                            qkind = Q_CLOSEST; state = ST_HIT; step = A_QUERY_READY;
                        } else {
                            double K = ws_load(ws, fslot(lv, FS_K));
                            result = add(scl(K, fA), scl(F, scl(S, result)));
                            fsp = lv;
                        }
                    }
                  }
#ifdef RAYHS_EXP_UNIFORM_LIGHT_LOOP
                } while (__ballot(step == A_RETURN) != 0ull);
#else
                } while (step == A_RETURN);
#endif
            }

        }

        // ---- the single scene-query site shared by every lane of the wave.  The compiler barriers keep
        // parked values in LDS / the workspace instead of forwarding them through registers.
        asm volatile("" ::: "memory");
        TIME_ADD(10);
        const bool q_point = qkind == Q_SHADOW_POINT;
        tr.begin(qkind, ray, q_point, q_d2);
        TIME_ADD(11);
        tr.traverse(qkind >= Q_SHADOW, ray, q_point, q_d2, q_dist);
        TIME_ADD(15);
        asm volatile("" ::: "memory");
    }

#ifdef RAYHS_PROFILE_TIME
    cnt.c[8] = __builtin_amdgcn_s_memrealtime() - rt0_;  // 100 MHz ticks over the wave's life
    unsigned long long *const counters = kargs()->counters;
    if (lane == 0 && counters != nullptr) {  // earliest start / latest end over all waves (slots 4, 5 are unused by the profile build)
        atomicMax(&counters[4], ~rt0_);
        atomicMax(&counters[5], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
    cnt.c[9] = __builtin_amdgcn_s_memrealtime();           // summed over waves: (latest end - mean end) is the idle tail
    if (lane == 0) {  // per wave, into the (now idle) workspace: start, end, when its last chunk was drawn, chunks (RAYHS_DUMP_WAVE_ENDS)
        unsigned long long *const w_ = (unsigned long long *)kargs()->frames + (size_t)(gtid >> 6) * 4;
        w_[0] = rt0_; w_[1] = cnt.c[9]; w_[2] = chunk_t0; w_[3] = cnt.c[7];
    }
    (void)mt0_;
#elif defined(RAYHS_WAVE_ENDS)  // development: the same record from the production kernel (two clock reads and one store per wave)
    if (lane == 0 && !COUNT) {
        unsigned long long *const w_ = (unsigned long long *)kargs()->frames + (size_t)(gtid >> 6) * 4;
        w_[0] = wave_t0_; w_[1] = __builtin_amdgcn_s_memrealtime(); w_[2] = chunk_t0; w_[3] = 1;
    }
#endif
#ifndef RAYHS_PROFILE_TIME
    unsigned long long *const counters = kargs()->counters;
#endif
    if ((COUNT || kProfileTime) && counters != nullptr) {
        for (int i = 0; i < kCounterSlots; i++) {
            unsigned long long v = cnt.c[i];
            if (kProfileTime && (i == 4 || i == 5)) continue;
            if (kProfileTime && (i >= 6 || i < 4) && lane != 0) v = 0;  // per-wave values: lane 0 only
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
            if (lane == 0 && v) atomicAdd(&counters[i], v);
        }
    }
}

}  // namespace rayhs

using namespace rayhs;

#ifndef RAYHS_WAVES_PER_EU
#define RAYHS_WAVES_PER_EU 4
#endif
#define RAYHS_DEFINE_KERNEL(NAME, EXH, COUNT, STACKT, WPE)                                                        \
    extern "C" __global__ void __launch_bounds__(TRACE_BLOCK, WPE)                                                \
    RAYHS_CAT(NAME, RAYHS_KERNEL_SUFFIX)(KArgs never_read_through_this_parameter) {                                \
        trace_body<EXH, COUNT, STACKT>();                                                                         \
    }

// average (src/RayHs.hs:169-171): foldl (+) black over the samples in order, then mul (1 / n)
extern "C" __global__ void __launch_bounds__(256)
RAYHS_CAT(rayhs_resolve, RAYHS_KERNEL_SUFFIX)(const double *sbuf, void *out, int n_pixels, int spp, int out_format) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pixels) return;
    const double *s = sbuf + (size_t)i * (size_t)spp * 3;
    V3 sum = mk(0, 0, 0);
    for (int k = 0; k < spp; k++) sum = add(sum, mk(s[3 * k], s[3 * k + 1], s[3 * k + 2]));
    store_pixel(out, out_format, (size_t)i, scl(1.0 / (double)spp, sum));
}

// Work order of the tiles for the NEXT frame from the chunk times the trace kernel just measured (and clears them).
// A persistent kernel ends when its slowest wave does.  A chunk of mesh or reflection pixels takes ~1 ms against
// 0.07 ms for an average one; in scan order such chunks are still being started when the frame is nearly done, and
// the other waves idled 12 % of the dragon frame waiting for them.  Handing the tiles out by decreasing cost --
// longest processing time first, with the cost MEASURED on the previous frame (same view, or a neighbouring one in an
// animation) -- leaves only cheap chunks for the end.  A counting sort into 1024 cost classes (the order inside a
// class does not matter), then the sorted list is dealt in groups of 8 to the 8 bands of the item space (one per XCD,
// see trace_body), so that every band starts with its share of the slow tiles.
// The classes are LOGARITHMIC, 64 per binade over [2^5, 2^21) (ticks of 10 ns: 0.3 us .. 21 ms per chunk; the static
// estimate is scaled into the same range): the class of a cost is a bit field of the float, so no pass over the costs is
// needed to find their maximum first, and the sort is two small many-workgroup kernels -- histogram, then scan + scatter
// -- instead of one workgroup doing three passes (0.040 ms per frame before, a tenth of the GPU time of the 0.25 ms
// Cornell frame; the scatter alone was 65 k scattered stores issued by one compute unit).
// Only the order in which pixels are worked on changes, never a result.
namespace rayhs {
constexpr int kOrderClasses = 1024, kOrderBlock = 1024;
DEV int cost_class_raw(float cost) {  // 0 (cheapest / never measured) .. 1023; monotone in cost
    const int e = (int)(__float_as_uint(cost) >> 17) - ((127 + 5) << 6);
    return min(max(e, 0), kOrderClasses - 1);
}
}  // namespace rayhs
// Block b counts the classes of ITS tiles (tile i belongs to block (i / 1024) % blocks in both kernels) into row b of
// `hist`: every row is written whole, nothing needs clearing, and no global atomic is involved.  (A first version
// added into one shared histogram and drew places from shared per-class cursors: the tiles of a frame crowd into a few
// classes, tens of thousands of atomics landed on a handful of addresses, and the scatter took 47 us against the 29 us
// of the one-workgroup kernel it replaced.)
extern "C" __global__ void __launch_bounds__(1024)
RAYHS_CAT(rayhs_order_hist, RAYHS_KERNEL_SUFFIX)(const float *cost, int n, unsigned *hist) {
    __shared__ unsigned local[kOrderClasses];
    const int t = threadIdx.x;
    local[t] = 0;
    __syncthreads();
    for (int i = blockIdx.x * kOrderBlock + t; i < n; i += gridDim.x * kOrderBlock) atomicAdd(&local[kOrderClasses - 1 - cost_class_raw(cost[i])], 1u);  // class 0 = slowest
    __syncthreads();
    hist[blockIdx.x * kOrderClasses + t] = local[t];
}
// Every block sums the rows (class totals, and what the blocks before it hold of each class), scans the totals, and
// puts its own tiles in place; clears the costs it has read and (block 0) the ticket counters of the next launch.
extern "C" __global__ void __launch_bounds__(1024)
RAYHS_CAT(rayhs_order_tiles, RAYHS_KERNEL_SUFFIX)(float *cost, int n, int *order, const unsigned *hist, unsigned long long *ticket_ring, int first, int count, int rings, int ticket_slots) {
    __shared__ unsigned cls_start[kOrderClasses], mine[kOrderClasses];
    __shared__ int top;
    const int t = threadIdx.x;
    unsigned total = 0, before = 0;
    for (int b = 0; b < (int)gridDim.x; b++) {
        const unsigned v = hist[b * kOrderClasses + t];
        total += v;
        before += b < (int)blockIdx.x ? v : 0u;
    }
    cls_start[t] = total;
    if (t == 0) top = kOrderClasses - 1;
    if (blockIdx.x == 0 && ticket_ring != nullptr && t < ticket_slots)  // the counters of the next `count` launches of the handle (ring slots first, first + 1, ...)
        for (int k = 0; k < count; k++) ticket_ring[(size_t)((first + k) % rings) * (size_t)ticket_slots + t] = 0ull;
    __syncthreads();
    if (total) atomicMin(&top, t);  // the slowest class that holds a tile
    for (int off = 1; off < kOrderClasses; off <<= 1) {  // inclusive prefix sum (Hillis-Steele)
        const unsigned add_ = t >= off ? cls_start[t - off] : 0u;
        __syncthreads();
        cls_start[t] += add_;
        __syncthreads();
    }
    mine[t] = cls_start[t] - total + before;  // where this block's tiles of class t begin
    __syncthreads();
    const int cycles = n / 64, rest = n % 64;  // sorted position p goes to band (p / 8) % 8
    const int top_raw = kOrderClasses - 1 - top;
    for (int i = blockIdx.x * kOrderBlock + t; i < n; i += gridDim.x * kOrderBlock) {
        const int raw = cost_class_raw(cost[i]), c = kOrderClasses - 1 - raw;
        const int p = (int)atomicAdd(&mine[c], 1u);
        const int g = p >> 3, band = g & 7;
        int band_begin = 0;
        for (int b = 0; b < band; b++) band_begin += cycles * 8 + min(max(rest - 8 * b, 0), 8);
        // issue priority of the wave that will work on it: 0..3 by quarters of the slowest tile's time (64 classes per
        // binade: 3/4 of the top is 27 classes below it, 1/2 is 64, 1/4 is 128)
        const int below = top_raw - raw;
        const int prio = below <= 26 ? 3 : below <= 63 ? 2 : below <= 127 ? 1 : 0;
        order[band_begin + (g >> 3) * 8 + (p & 7)] = i | (prio << kTileBits);
        cost[i] = 0.0f;
    }
}

// Work order of the FIRST frame of a geometry, which has no measured chunk times yet (and is the only frame a one-shot
// `rayhs -oX scene.json` renders, src/RayHs.hs:215-234): a static estimate per 8x8 tile.  One thread per tile looks
// through the tile's centre and its four corners.  Rays that meet a mesh's root box (the reference root's box,
// inflated) make the tile expensive in proportion to their number -- those chunks walk the BVH, and their shadow and
// reflection rays start inside it.  The surface the centre ray sees first adds the rest: a reflective or refractive
// material spawns secondary rays (more when the mirrored ray heads for a mesh), and every light whose shadow segment
// crosses a mesh box costs a traversal.  rayhs_order_tiles turns the estimate into the same slowest-first order and
// issue priorities as a measured frame.  Scheduling only: nothing here enters a pixel.
namespace rayhs {
DEV bool hits_mesh_box(const DMesh &me, V3 o, V3 d, double t_limit) {
    double tmin = -__builtin_inf(), tmax = __builtin_inf();
    const double oo[3] = {o.x, o.y, o.z}, iv[3] = {1 / d.x, 1 / d.y, 1 / d.z};
    for (int a = 0; a < 3; a++) {
        const double pad = 0.02 * (me.hi[a] - me.lo[a]) + 1e-9;
        const double t1 = (me.lo[a] - pad - oo[a]) * iv[a], t2 = (me.hi[a] + pad - oo[a]) * iv[a];
        tmin = fmax(tmin, fmin(t1, t2)); tmax = fmin(tmax, fmax(t1, t2));  // fmin / fmax drop a NaN (0 * inf): that axis then does not cull
    }
    return tmax >= fmax(tmin, 0.0) && tmin <= t_limit;
}
// How many boxes of the upper kStaticDepth levels of a mesh's BVH the ray meets: a cheap stand-in for "how much of the
// mesh is along this ray" (the root box alone is mostly empty space around the model).
constexpr int kStaticDepth = 9;
DEV int upper_boxes_met(const DScene &sc, const DMesh &me, V3 o, V3 d) {
    const float ox = (float)(o.x - me.center[0]), oy = (float)(o.y - me.center[1]), oz = (float)(o.z - me.center[2]);
    const float ix = 1.0f / (float)d.x, iy = 1.0f / (float)d.y, iz = 1.0f / (float)d.z;
    int stack[kStaticDepth + 1], depth[kStaticDepth + 1], sp = 0, met = 0;
    stack[sp] = me.froot; depth[sp] = 0; sp++;
    while (sp > 0) {
        sp--;
        const int node = stack[sp], dep = depth[sp];
        const FNode &nd = sc.fnodes[node];
        for (int ch = 0; ch < 2; ch++) {
            const float *b = ch ? nd.b1 : nd.b0;
            const int c = ch ? nd.c1 : nd.c0;
            if (c == FNODE_EMPTY) continue;
            const float t1 = (b[0] - ox) * ix, t2 = (b[1] - ox) * ix, t3 = (b[2] - oy) * iy, t4 = (b[3] - oy) * iy, t5 = (b[4] - oz) * iz, t6 = (b[5] - oz) * iz;
            const float tn = fmaxf(fmaxf(fminf(t1, t2), fminf(t3, t4)), fminf(t5, t6)), tf = fminf(fminf(fmaxf(t1, t2), fmaxf(t3, t4)), fmaxf(t5, t6));
            if (!(tf >= fmaxf(tn, 0.0f))) continue;
            met++;
            if (c >= 0 && dep + 1 < kStaticDepth && sp <= kStaticDepth) { stack[sp] = c; depth[sp] = dep + 1; sp++; }
        }
    }
    return met;
}
}  // namespace rayhs
extern "C" __global__ void __launch_bounds__(256)
RAYHS_CAT(rayhs_static_costs, RAYHS_KERNEL_SUFFIX)(DScene sc, DCamera cam, DParams prm, float *cost) {
    // Eight lanes per tile: lane k < 5 looks through one of the tile's five rays (k = 0: the centre ray, which also classes
    // the surface it sees), the counts are added up across the eight lanes.  With one THREAD per tile the kernel was a
    // long serial chain on 512 waves, half the chip idle: 0.14 ms in front of the first frame, more than half of the 0.25 ms
    // Cornell frame it was there to speed up.
    const int tile = (blockIdx.x * 256 + threadIdx.x) >> 3, k_lane = threadIdx.x & 7;
    const bool live = tile < prm.n_tiles;  // the same for the eight lanes of a tile (they meet again in the shuffles below)
    if (__ballot(live) == 0ull) return;    // the whole wave is past the last tile (the ballot is evaluated by every lane)
    const int ty = (live ? tile : 0) / prm.tiles_x, tx = (live ? tile : 0) - ty * prm.tiles_x;
    auto image_row = [&](int lrow) {  // local row -> image row (stripe interleave, as in the trace kernel)
        if (prm.stripe_count == 1 && prm.stripe_map == nullptr) return prm.row0 + lrow;
        const int s_local = lrow / prm.stripe_rows, in_stripe = lrow - s_local * prm.stripe_rows;
        const int s_global = prm.stripe_map ? prm.stripe_map[s_local] : s_local * prm.stripe_count + prm.stripe_rank;
        return prm.row0 + s_global * prm.stripe_rows + in_stripe;
    };
    const int x0 = tx * 8, x1 = min(x0 + 8, prm.width), l0 = ty * 8, l1 = min(l0 + 8, prm.n_rows);
    const double ytop = (double)image_row(l0), ybot = (double)image_row(l1 - 1) + 1.0;
    const double inf = __builtin_inf();
    float c = 1.0f;
    int box_rays = 0, boxes_met = 0;
    for (int k = k_lane; k < 5 && live; k += 8) {  // (one pass: this lane's ray)
        // sample offsets reach half a pixel beyond the pixel corners (src/RayHs.hs:178-188): look half a pixel wider
        const double fx = k == 0 ? 0.5 * (double)(x0 + x1) : ((k & 1) ? (double)x0 - 0.5 : (double)x1 + 0.5);
        const double fy = k == 0 ? 0.5 * (ytop + ybot) : ((k & 2) ? ytop - 0.5 : ybot + 0.5);
        const Ray r = ray_from_pixel(cam, fx, fy);
        bool in_box = false;
        for (int m = 0; m < sc.n_meshes; m++)
            if (sc.meshes[m].froot != FNODE_NONE && hits_mesh_box(sc.meshes[m], r.o, r.d, inf)) { in_box = true; boxes_met += upper_boxes_met(sc, sc.meshes[m], r.o, r.d); }
        box_rays += in_box ? 1 : 0;
        if (k != 0) continue;
        // the first analytic surface the centre ray meets (approximate is enough: it only picks a class)
        double best = inf;
        int obj = -1;
        V3 nrm = mk(0, 0, 0);
        for (int i = 0; i < sc.n_planes; i++) {
            const DPlane &pl = sc.planes[i];
            const double den = dot(r.d, ld3(pl.n)), t = dot(ld3(pl.n), sub(ld3(pl.p), r.o)) / den;
            if (fabs(den) > 0 && t > 0 && t < best) { best = t; obj = pl.obj; nrm = ld3(pl.n); }
        }
        for (int i = 0; i < sc.n_spheres; i++) {
            const DSphere &sp = sc.spheres[i];
            const V3 oc = sub(r.o, ld3(sp.c));
            const double a = dot(r.d, r.d), b = 2.0 * dot(r.d, oc), cc = dot(oc, oc) - sp.r2, delta = b * b - 4.0 * a * cc;
            if (delta < 0) continue;
            const double sq = sqrt(delta), t0 = 0.5 * (-b - sq) / a, t1 = 0.5 * (-b + sq) / a, t = t0 > 0 ? t0 : t1;
            if (t > 0 && t < best) { best = t; obj = sp.obj; nrm = sub(add(r.o, scl(t, r.d)), ld3(sp.c)); }
        }
        if (obj < 0) continue;
        const V3 p = add(r.o, scl(best, r.d));
        const int mk_ = sc.objects[obj].mat_kind;  // Mirror 0, Diffuse 1, Plastic 2, Transparent 4 (src/RayHs.hs:107-143)
        if (mk_ == 1 || mk_ == 2)  // one shadow query per light; the ones that cross a mesh box walk its BVH
            for (int li = 0; li < sc.n_lights; li++) {
                const DLight &L = sc.lights[li];
                const V3 d = L.kind == 0 ? ld3(L.v) : sub(ld3(L.v), p);
                for (int m = 0; m < sc.n_meshes; m++)
                    if (sc.meshes[m].froot != FNODE_NONE && hits_mesh_box(sc.meshes[m], p, d, L.kind == 0 ? inf : 1.0)) { c += 0.25f; break; }
            }
        if (mk_ == 0 || mk_ == 2 || mk_ == 4) {
            c += mk_ == 4 ? 4.0f : 2.5f;
            const V3 n = unit(nrm), rf = reflect(r.d, n);
            for (int m = 0; m < sc.n_meshes; m++)
                if (sc.meshes[m].froot != FNODE_NONE && hits_mesh_box(sc.meshes[m], add(p, scl(1e-6, rf)), rf, inf)) { c += 3.0f; break; }
        }
    }
    for (int off = 1; off < 8; off <<= 1) {  // the eight lanes of a tile are neighbours in the wave
        box_rays += __shfl_xor(box_rays, off);
        boxes_met += __shfl_xor(boxes_met, off);
        c += __shfl_xor(c, off) - 1.0f;  // (every lane started from the base cost 1; only lane 0 added to it)
    }
    if (!live || k_lane != 0) return;
    // weights fitted to the chunk times a measured frame records (dragon scenes, profiles/r02/static_costs.md): in units of a
    // diffuse wall tile, a tile is worth ~3 when its rays only graze the upper boxes and up to ~17 deep inside the model
    if (box_rays > 0) c += boxes_met < 40 ? 0.045f * (float)boxes_met : fminf(16.0f, 1.8f + 0.26f * (float)(boxes_met - 40));
    if (prm.flags_dev & 1) { cost[tile] = (float)boxes_met; return; }  // development: dump the raw feature (RAYHS_DUMP_TILE_COSTS)
    cost[tile] = c * 4096.0f;  // into the range the ordering kernels resolve (they class costs on a fixed logarithmic scale made for ticks)
}

RAYHS_DEFINE_KERNEL(rayhs_trace, false, false, unsigned short, RAYHS_WAVES_PER_EU)       // < 65536 BVH nodes
RAYHS_DEFINE_KERNEL(rayhs_trace_wide, false, false, unsigned int, RAYHS_WAVES_PER_EU)   // any scene
#ifdef RAYHS_PROFILE_TIME
RAYHS_DEFINE_KERNEL(rayhs_trace_count, false, false, unsigned short, RAYHS_WAVES_PER_EU)
#else
RAYHS_DEFINE_KERNEL(rayhs_trace_count, false, true, unsigned int, 2)
#endif
RAYHS_DEFINE_KERNEL(rayhs_trace_exh_count, true, true, unsigned int, 1)

// rayhs_multi: a device's finished stripes (compact, in the order of its stripe list) go to their rows of the whole
// frame, which lives on the first device -- a peer WRITE over xGMI when the two are different GPUs.  HBM / link bound:
// every byte is read once and written once, in units of `unit` bytes (16 when the row size allows it).
extern "C" __global__ void __launch_bounds__(256)
RAYHS_CAT(rayhs_scatter_stripes, RAYHS_KERNEL_SUFFIX)(const char *src, char *dst, const int *stripe_map, int n_rows, int stripe_rows,
                                                      long long row_bytes, int unit) {
    const long long units_per_row = row_bytes / unit, total = units_per_row * (long long)n_rows;
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long long)gridDim.x * 256) {
        const int lrow = (int)(k / units_per_row);
        const long long in_row = (k - (long long)lrow * units_per_row) * unit;
        const int s_local = lrow / stripe_rows, in_stripe = lrow - s_local * stripe_rows;
        const long long drow = (long long)stripe_map[s_local] * stripe_rows + in_stripe;
        const char *a = src + (long long)lrow * row_bytes + in_row;
        char *b = dst + drow * row_bytes + in_row;
        if (unit == 16) *(uint4 *)b = *(const uint4 *)a;
        else if (unit == 8) *(uint2 *)b = *(const uint2 *)a;
        else if (unit == 4) *(unsigned *)b = *(const unsigned *)a;
        else *(unsigned short *)b = *(const unsigned short *)a;
    }
}

// rayhs_selftest_arith (include/rayhs.h): rcp_in_range / div_in_range / sqrt_in_range of short_div.h against the compiler's own correctly
// rounded sequences, bit for bit, on pseudo-random operands with exponents spread over the ranges the trace kernel uses them in.
// bad[c]: operands of case c on which the two differ (expected: none).  Cases: 0  1 / x, |x| in [2^-767, 2^1021];  1  1 / x, |x| in
// [2^-20, 2^366] (the determinant of an accepted triangle);  2  n / d, both in [2^-200, 2^200] (a light's d / radius), every 128th +0 / |d| (a primary ray through the image's centre column);  3  sqrt x, x in
// [2^-399, 2^399] (a light's squared distance behind its range test);  4  sqrt x, x in [2^-767, 2^1023].
namespace rayhs {
DEV double selftest_draw(uint64_t i, uint64_t salt, int elo, int ehi, bool positive) {
    const uint64_t h = mix64(i * 0x9e3779b97f4a7c15ULL + salt);
    uint64_t mant = h & ((1ULL << 52) - 1);
    if ((i & 63) == 0) mant = ((h >> 60) & 1) ? (1ULL << 52) - 1 : 0;   // every 64th operand an edge: a power of two, or all ones below it
    const int e = elo + (int)((h >> 52) % (uint64_t)(ehi - elo + 1));
    const uint64_t bits = ((positive ? 0ULL : (mix64(h) & 1)) << 63) | ((uint64_t)(e + 1023) << 52) | mant;
    return __longlong_as_double((long long)bits);
}
}  // namespace rayhs
extern "C" __global__ void __launch_bounds__(256)
RAYHS_CAT(rayhs_selftest_arith, RAYHS_KERNEL_SUFFIX)(unsigned long long *bad, unsigned long long n, unsigned long long seed) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    auto same = [](double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); };
    const double x0 = selftest_draw(i, seed + 1, -767, 1021, false), x1 = selftest_draw(i, seed + 2, -20, 366, false);
    double d2 = selftest_draw(i, seed + 3, -200, 200, false), n2 = selftest_draw(i, seed + 4, -200, 200, false);
    if ((i & 127) == 1) { n2 = 0.0; d2 = fabs(d2); }   // (+0 by a positive divisor: the centre column of ray_from_pixel)
    const double x3 = selftest_draw(i, seed + 5, -399, 399, true), x4 = selftest_draw(i, seed + 6, -767, 1023, true);
    if (!same(rcp_in_range(x0), 1.0 / x0)) atomicAdd(&bad[0], 1ULL);
    if (!same(rcp_in_range(x1), 1.0 / x1)) atomicAdd(&bad[1], 1ULL);
    if (!same(div_in_range(n2, d2), n2 / d2)) atomicAdd(&bad[2], 1ULL);
    if (!same(sqrt_in_range(x3), sqrt(x3))) atomicAdd(&bad[3], 1ULL);
    if (!same(sqrt_in_range(x4), sqrt(x4))) atomicAdd(&bad[4], 1ULL);
}

// rayhs_device_prepare launches this once: the first launch of a process pays for the compute queue and for putting this
// translation unit's code object on the device (~20 ms), which a one-shot `rayhs -oX scene.json` can spend while it still parses
extern "C" __global__ void RAYHS_CAT(rayhs_warmup, RAYHS_KERNEL_SUFFIX)() {}

// Host-side launcher of this translation unit's kernels (called from abi.cpp).
namespace rayhs {
hipError_t RAYHS_CAT(launch_selftest_arith, RAYHS_KERNEL_SUFFIX)(hipStream_t stream, unsigned long long *bad, unsigned long long n, unsigned long long seed) {
    hipLaunchKernelGGL(RAYHS_CAT(rayhs_selftest_arith, RAYHS_KERNEL_SUFFIX), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, bad, n, seed);
    return hipGetLastError();
}
hipError_t RAYHS_CAT(launch_warmup, RAYHS_KERNEL_SUFFIX)(hipStream_t stream) {
    hipLaunchKernelGGL(RAYHS_CAT(rayhs_warmup, RAYHS_KERNEL_SUFFIX), dim3(1), dim3(64), 0, stream);
    return hipGetLastError();
}
static const void *RAYHS_CAT(kernel_ptr, RAYHS_KERNEL_SUFFIX)(int variant) {
    switch (variant) {
    case 0: return (const void *)RAYHS_CAT(rayhs_trace, RAYHS_KERNEL_SUFFIX);
    case 1: return (const void *)RAYHS_CAT(rayhs_trace_count, RAYHS_KERNEL_SUFFIX);
    case 2: return (const void *)RAYHS_CAT(rayhs_trace_exh_count, RAYHS_KERNEL_SUFFIX);
    default: return (const void *)RAYHS_CAT(rayhs_trace_wide, RAYHS_KERNEL_SUFFIX);
    }
}
hipError_t RAYHS_CAT(launch_trace, RAYHS_KERNEL_SUFFIX)(int variant, int grid, size_t lds_bytes, hipStream_t stream,
                                                        const DScene &sc, const DCamera &cam, const DParams &prm, void *out,
                                                        unsigned long long *counters, unsigned long long *ticket, double *frames) {
    KArgs ka;
    ka.sc = sc; ka.cam = cam; ka.prm = prm; ka.out = out; ka.counters = counters; ka.ticket = ticket; ka.frames = frames;
    void *args[] = {(void *)&ka};
    return hipLaunchKernel(RAYHS_CAT(kernel_ptr, RAYHS_KERNEL_SUFFIX)(variant), dim3((unsigned)grid), dim3(TRACE_BLOCK), args, lds_bytes, stream);
}
hipError_t RAYHS_CAT(launch_resolve, RAYHS_KERNEL_SUFFIX)(hipStream_t stream, const double *sbuf, void *out, int n_pixels, int spp, int out_format) {
    hipLaunchKernelGGL(RAYHS_CAT(rayhs_resolve, RAYHS_KERNEL_SUFFIX), dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, stream, sbuf, out, n_pixels, spp, out_format);
    return hipGetLastError();
}
hipError_t RAYHS_CAT(launch_static_costs, RAYHS_KERNEL_SUFFIX)(hipStream_t stream, const DScene &sc, const DCamera &cam, const DParams &prm, float *cost) {
    hipLaunchKernelGGL(RAYHS_CAT(rayhs_static_costs, RAYHS_KERNEL_SUFFIX), dim3((unsigned)((prm.n_tiles + 31) / 32)), dim3(256), 0, stream, sc, cam, prm, cost);  // 8 lanes per tile
    return hipGetLastError();
}
hipError_t RAYHS_CAT(launch_scatter_stripes, RAYHS_KERNEL_SUFFIX)(hipStream_t stream, const void *src, void *dst, const int *stripe_map, int n_rows,
                                                               int stripe_rows, long long row_bytes) {
    if (n_rows <= 0 || row_bytes <= 0) return hipSuccess;
    int unit = 2;
    for (int u : {16, 8, 4})
        if (row_bytes % u == 0 && (unsigned long long)src % (unsigned)u == 0 && (unsigned long long)dst % (unsigned)u == 0) { unit = u; break; }
    const long long total = row_bytes / unit * (long long)n_rows;
    const unsigned grid = (unsigned)std::min<long long>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(RAYHS_CAT(rayhs_scatter_stripes, RAYHS_KERNEL_SUFFIX), dim3(grid), dim3(256), 0, stream, (const char *)src, (char *)dst, stripe_map,
                       n_rows, stripe_rows, row_bytes, unit);
    return hipGetLastError();
}
// aux: kOrderMaxBlocks * 1024 unsigned (one histogram row per block).  ticket_ring (may be NULL): `ticket_slots` counters per launch, the
// scatter kernel zeroes for the next `count` launches of the handle (ring slots first, first + 1, ... modulo `rings`).
hipError_t RAYHS_CAT(launch_order_tiles, RAYHS_KERNEL_SUFFIX)(hipStream_t stream, float *cost, int n, int *order, unsigned *aux, unsigned long long *ticket_ring, int first, int count, int rings, int ticket_slots) {
    const unsigned grid = (unsigned)std::max(1, std::min(ORDER_MAX_BLOCKS, (n + kOrderBlock - 1) / kOrderBlock));
    hipLaunchKernelGGL(RAYHS_CAT(rayhs_order_hist, RAYHS_KERNEL_SUFFIX), dim3(grid), dim3(kOrderBlock), 0, stream, (const float *)cost, n, aux);
    hipLaunchKernelGGL(RAYHS_CAT(rayhs_order_tiles, RAYHS_KERNEL_SUFFIX), dim3(grid), dim3(kOrderBlock), 0, stream, cost, n, order, (const unsigned *)aux, ticket_ring, first, count, rings, ticket_slots);
    return hipGetLastError();
}
int RAYHS_CAT(occupancy_trace, RAYHS_KERNEL_SUFFIX)(int variant, size_t lds_bytes) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, RAYHS_CAT(kernel_ptr, RAYHS_KERNEL_SUFFIX)(variant), TRACE_BLOCK, lds_bytes) != hipSuccess) return 0;
    return n;
}
#ifdef RAYHS_PROFILE_TIME
int RAYHS_CAT(stack_entry_bytes, RAYHS_KERNEL_SUFFIX)(int variant) { return variant <= 1 ? 2 : 4; }
#else
int RAYHS_CAT(stack_entry_bytes, RAYHS_KERNEL_SUFFIX)(int variant) { return variant == 0 ? 2 : 4; }
#endif
}  // namespace rayhs
