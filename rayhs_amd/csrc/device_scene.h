// device_scene.h -- HBM-resident scene layout shared by the launcher (abi.cpp)
// and the trace kernels (trace_kernel.hip).
//
// Reals stay fp64 (src/Vec.hs:29-31) wherever a value enters the result.
// Values the reference recomputes per ray from scene constants (triangle edges,
// plane bitangent, r*r, Fresnel r0) are computed ONCE on the host with the same
// fp64 operations, so the stored value is bit-identical to what the reference
// would derive.  Only the culling boxes of the fast traversal structure are
// fp32 (conservative, see bvh_build.cpp): they never enter a result.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace rayhs {

// ---- reference tree, used by the exhaustive (validation / counting) kernels ----
// 64 B per node (src/KDTree.hs:59-61).
struct alignas(16) DNode {
    double lo[3], hi[3];  // Box (src/KDTree.hs:13-14)
    int32_t a, b;         // inner: left/right child (-1 = Empty); leaf: first triangle, count
    int32_t flags;        // bit 31 = leaf, bits 0..1 = split axis of an inner node
    int32_t parent;       // parent node, -1 for a root
};
static constexpr int32_t DNODE_LEAF = (int32_t)0x80000000;

// ---- fast traversal structure (bvh_build.cpp) ----
// 64 B per node: both children's boxes as mesh-centred fp32 planes {lo.x, hi.x, lo.y, hi.y, lo.z, hi.z}, inflated
// by 2^-18 of the mesh extent and rounded outward (abi.cpp), so that the slab test is one packed fused
// multiply-add per axis,  t = plane * (1/d) + (-o/d),  with no decode step.  (32-byte nodes with 16-bit grid
// coordinates were 2 % slower once the waves were coherent: 12 conversions per visit cost more than the bytes.)
// Child reference: >= 0 inner node, < 0 leaf ~((first << 3) | count) into the BVH-ordered triangle
// array.  A node always has two children: the missing sibling of a mesh that is one leaf is the EMPTY LEAF -1
// (~0: no triangles) with that leaf's box, so that every stored box is finite and the walks test no reference.
struct alignas(16) FNode {
    float b0[6], b1[6];
    int32_t c0, c1;
    int32_t pad_[2];
};
static constexpr int32_t FNODE_NONE = (int32_t)0x80000000;  // "no node": an empty mesh's root, the walks' idle marker
static constexpr int32_t FNODE_EMPTY = -1;                  // the empty leaf

// 80 B: what triangleIntersection reads (src/Mesh.hs:59-72): p0, e1 = p1-p0, e2 = p2-p0,
// plus the triangle's place in the reference's order for exact-t ties.
struct alignas(16) DTri {
    double p0[3], e1[3], e2[3];
    int32_t rank;        // index in the reference's left-to-right leaf order
    int32_t leaf_first;  // rank of the first triangle of its reference leaf
};
// 128 B: read once per shaded hit (src/Mesh.hs:73-74).
struct alignas(16) DTriAttr {
    double n0[3], n1[3], n2[3];
    double uv0[2], uv1[2], uv2[2];
    int32_t leaf_node;    // reference-tree leaf holding the triangle (BVH-ordered copy only)
    int32_t pad;
};

struct alignas(32) DPlane {   // src/Geometry.hs:62, 70-79
    // the first 32 bytes are what the query loop reads of an axis-aligned plane: ONE scalar load
    int32_t obj, occluder;          // scene object index; 0 for Emmit (never an occluder, src/RayHs.hs:81)
    uint64_t shadow_skip;           // occluder ? 0 : ~0 -- the lanes' "shadow query" mask is OR-ed with it (an Emmit plane is
                                    // skipped by every shadow query, an occluder only by those already occluded)
    double nk, pk;                  // planes whose normal has ONE non-zero component k (DScene.plane_axis_end): n[k], p[k]
    double p[3], n[3];
    double t[3], b[3];              // b = cross t n (only the UV of a shaded hit reads these)
};
static_assert(sizeof(DPlane) == 128, "plane record: 32-byte head + four vectors");
struct alignas(16) DSphere {  // src/Geometry.hs:63, 81-96.  48 bytes: one eight-dword and one four-dword scalar load per test
    double c[3], r2;                // r2 = r ^ 2
    int32_t obj, occluder;
    uint64_t shadow_skip;           // occluder ? 0 : ~0, as for the planes: the lanes' "shadow query" mask is and-ed with its complement
};
static_assert(sizeof(DSphere) == 48, "sphere record");
struct alignas(64) DMesh {
    // the first 64 bytes are what the fast traversal's per-mesh set-up reads: ONE sixteen-dword scalar load (trace_kernel.hip, traverse())
    int32_t root;                   // reference tree root, -1 = Empty
    int32_t obj, occluder;
    int32_t froot;                  // fast BVH root node, FNODE_NONE = no triangles
    double center[3];               // origin of the fp32 box coordinates
    float flo[3], fhi[3];           // bounds of the mesh in those coordinates, rounded outward and inflated
    // ---- the rest
    double lo[3], hi[3];            // the reference root's box (src/KDTree.hs:84)
    int32_t tri_base;               // first triangle of this mesh in the BVH-ordered arrays
    int32_t pad;
};
static_assert(offsetof(DMesh, center) == 16 && offsetof(DMesh, flo) == 40 && offsetof(DMesh, fhi) == 52 && offsetof(DMesh, lo) == 64 && sizeof(DMesh) == 128,
              "the head of DMesh is one 64-byte fetch");
struct DObject {  // Material (src/Material.hs:13-19) + ColorMap (src/ColorMap.hs:11-16) of one object
    // the first 64 bytes are what shading a hit reads (DObjectHot): fetched per lane as one batch of four 16-byte loads
    int32_t mat_kind, cmap_kind, tex_w, tex_h;
    double c1[3];       // flat colour / checker color1
    double ior;
    double fr;          // r0 1.0 ior  (src/Material.hs:28)
    double one_m_r0;    // 1 - r0 ior 1.0 (src/RayHs.hs:133)
    double ce[3], c2[3], size;
    int64_t tex_off;
};
struct DObjectHot { int32_t mat_kind, cmap_kind, tex_w, tex_h; double c1[3], ior, fr, one_m_r0; };
static_assert(sizeof(DObjectHot) == 64, "the hot head of DObject is one 64-byte fetch");
struct DLight {   // src/Light.hs:8-10
    int32_t kind;
    int32_t short_ok;   // point light whose radius lies in [1e-60, 1e60]: its falloff may be computed without the divisions' range fix-ups (trace_kernel.hip, rcp_in_range)
    double v[3], c[3], radius;
};

struct DScene {
    // ---- the first 64 bytes are what every scene query reads: the trace kernel fetches them at the top of begin() with one
    // scalar load (by dword index: keep this order)
    const DPlane *planes;
    const DSphere *spheres;
    int32_t n_planes, n_spheres;
    int32_t sphere_occluders;  // spheres that can occlude (not Emmit): with none, a round of shadow queries skips the sphere loop
    int32_t n_lights;
    // planes are stored grouped: the records in [0, plane_axis_end[0]) have a normal along x, [.., [1]) along y, [.., [2]) along z, the
    // rest are in general position (the query loop runs a body specialised per group; ties go by object index).  BYTE offsets into
    // `planes` (multiples of sizeof(DPlane)): the loop counts in them
    int32_t plane_axis_end[3];
    int32_t n_meshes;
    const DMesh *meshes;
    const FNode *fnodes;       // fast BVH
    const DTri *ftris;         // BVH leaf order
    // ---- the rest
    const DTriAttr *fattrs;    // BVH leaf order
    const DNode *nodes;        // reference tree
    const DTri *tris;          // reference order
    const DTriAttr *attrs;     // reference order
    const DObject *objects;
    const DLight *lights;
    const double *texels;
    int32_t n_objects, n_nodes, n_tris, n_fnodes;
};
static_assert(offsetof(DScene, n_planes) == 16 && offsetof(DScene, plane_axis_end) == 32 && offsetof(DScene, n_meshes) == 44 && offsetof(DScene, meshes) == 48 &&
              offsetof(DScene, ftris) == 64, "the head of DScene is read as two fixed blocks");

// Camera constants of rayFromPixel / unproject / lookAt (src/Projection.hs:22-46,
// src/Mat.hs:89-93), evaluated on the host once per call (the reference
// re-evaluates them per pixel to the same values).
struct DCamera {
    double pos[3];
    double m[9];          // lookAt matrix, row-major
    double w, h;          // fromIntegral width / height
    double half_w, half_h;  // w/2, h/2
    double apw, aph;      // aspectSize
    double f;             // 0.5 * h / (tan 0.5 * fovy)
    int32_t proj_kind;
    int32_t short_ok;     // the constants are in the range the short division / square root of short_div.h need (make_camera, abi.cpp)
};

struct DParams {
    // ---- the first 64 bytes are what a wave reads when it draws a chunk of work: the trace kernel fetches them with ONE scalar
    // load (ParamsHot in trace_kernel.hip mirrors this block field by field)
    int32_t spp;
    int32_t spp_shift;                // log2(spp) when spp is a power of two, else -1 (the refill then divides)
    int32_t width;
    int32_t n_rows;                   // rows rendered by this call
    int32_t tiles_x;                  // 8x8-pixel tiles per row of tiles
    float inv_tiles_x;                // 1 / tiles_x: tile -> (row, column) of the tile grid without an integer division
    int32_t n_tiles;                  // tiles over (width x n_rows)
    int32_t wave_resolve;             // 1: spp divides 64, samples are folded inside the wave (no sample buffer)
    int32_t out_format;
    int32_t stripe_count;
    int32_t row0;                     // first candidate row
    int32_t stripe_shift;             // log2(stripe_rows) when that is a power of two, else -1
    uint64_t seed;                    // mix64(seed + 0x9e3779b97f4a7c15): the seed term of rayhs_sample01, mixed on the host
    double inv_spp;                   // 1.0 / spp (the same IEEE division the reference's average performs, done once on the host)
    // ---- the rest
    int32_t height, max_depth;
    int32_t stripe_rows, stripe_rank;
    int32_t stack_cap;                // per-lane LDS stack entries
    int32_t ctx_slots;                // per-lane LDS context doubles: CTX_SLOTS, or CTX_SLOTS + 3 (light term too) when it fits
    int32_t frame_stride;             // threads in the grid (workspace stride)
    int32_t flags_dev;                // development switches of the helper kernels (bit 0: rayhs_static_costs emits its raw feature); bit 1: RAYHS_FLAG_POW5_SQMUL
    // ---- the last 32 bytes are pointers a chunk refill reads; the trace kernel's argument block continues with four more
    // (KArgs in trace_kernel.hip: out, counters, ticket, frames), and the eight come with one scalar load
    double *sbuf;                     // spp > 1: per-sample radiance, 3 doubles per (pixel, sample)
    const int32_t *tile_order;        // work order of the 8x8 tiles, slowest first (NULL: scan order); from the previous frame's tile_cost
    const int32_t *stripe_map;        // NULL: round-robin stripes (stripe_rank / stripe_count); else the global stripe index of every
                                      // local stripe of this launch, ascending (rayhs_multi: cost-balanced ownership)
    float *tile_cost;                 // per tile: longest time (100 MHz ticks) one of its chunks took in THIS frame (NULL: not recorded)
};
static_assert(sizeof(DParams) % 8 == 0 && offsetof(DParams, sbuf) + 32 == sizeof(DParams), "DParams ends with its four pointers");

// counters[] slots
enum { CNT_PRIMARY = 0, CNT_REFLECT, CNT_REFRACT, CNT_REFRACT_EXIT, CNT_SHADOW, CNT_NODES, CNT_TRIS,
       CNT_PRIMS, CNT_SHADED, CNT_PIXELS,
       CNT_SHADOW_SKIPPED,  // shadow queries counted in CNT_SHADOW but not traced (zero light term); slot 10 doubles as a
                            // development counter in -DRAYHS_PROFILE_LOOPS / -DRAYHS_PROFILE_TIME builds
       CNT_SLOTS = 16,
       CNT_SLOTS_DEV = 64 };  // development builds (-DRAYHS_PROFILE_LOOPS) also write slots 16..63 (trace_kernel.hip, LOOP_TICK and the
                              // per-lane-walk census): the library's own buffer holds that many

static constexpr int ORDER_MAX_BLOCKS = 64;  // workgroups of the tile-ordering kernels (one 1024-entry histogram row each)
static constexpr int TRACE_BLOCK = 256;     // threads per workgroup (4 waves)
static constexpr int TICKET_STRIDE = 16;    // uint64 between the ticket counters of two bands: each alone on a 128-byte L2 line; the line
                                            // behind the eighth holds the eight "band is dry" flags
static constexpr int FRAME_SLOTS = 14;      // doubles per continuation frame
static constexpr int LANE_SLOTS = 6;        // per-lane doubles kept in the global workspace besides frames (view dir, light term)
#ifndef RAYHS_CTX_SLOTS
#define RAYHS_CTX_SLOTS 12
#endif
// per-lane doubles parked in LDS across a shadow / exit query: p, n, cd, A; with 18 also the light term
// and the view direction (otherwise those two go through the global workspace)
static constexpr int CTX_SLOTS = RAYHS_CTX_SLOTS;

}  // namespace rayhs
