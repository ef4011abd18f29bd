/*
 * rayhs.h -- C ABI of librayhs_hip.so, the MI355X (gfx950) replacement for the
 * per-pixel trace loop of oilandrust/rayhs.
 *
 * The reference has no FFI; the seam this ABI replaces is the pure function
 *     rayTrace :: Rendering -> Image                      (src/RayHs.hs:161-166)
 * i.e. the closure handed to generatePixels (src/Image.hs:34-36).  A host (the
 * reference's Haskell Descriptors/JSON/Scene code, or the C++ host shipped here)
 * flattens its immutable Scene (src/Scene.hs:8-10) into the SoA arrays of
 * rayhs_scene_desc, hands them over once with rayhs_scene_create, then calls
 * rayhs_render* per frame.  INTEGRATION.md shows the `foreign import ccall`
 * stub a maintainer of the reference would add.
 *
 * Conventions: plain pointers and sizes only; 0 on success, negative
 * rayhs_status otherwise; no exceptions cross the boundary; the library owns
 * opaque handles, the caller owns every buffer it passes in (inputs are copied
 * by rayhs_scene_create); a handle may be used from one thread at a time;
 * rayhs_last_error() is thread-local and valid until the next call on the
 * thread.  There is no CPU fallback: without a HIP device every compute entry
 * point fails with RAYHS_ERR_NO_DEVICE.
 */
#ifndef RAYHS_H
#define RAYHS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  Version 2 GREW rayhs_stats by 8 bytes (shadow_skipped behind kernel_ms) and added the rayhs_multi_*
 * entry points.  The library writes sizeof(rayhs_stats) bytes of ITS version through every `stats` pointer, so a
 * caller compiled against version 1 that asks for stats must be rebuilt against this header; it cannot be served
 * by a version-2 library (callers that pass stats = NULL are unaffected).  A host that mirrors the structs by hand
 * (a Haskell Storable instance, ctypes) checks its layout with rayhs_sizeof() at start-up and the version with
 * rayhs_abi_version(); rayhs_amd/lib.py refuses a library that answers differently.
 * Version 3 changed no struct: it ADDED rayhs_sizeof(), rayhs_device_prepare() and RAYHS_FLAG_POW5_SQMUL, and render
 * calls now fail with RAYHS_ERR_INVALID on flag bits the library does not know (RAYHS_FLAG_ALL) instead of ignoring
 * them -- a version-2 library would have rendered the flat-product reading of x ^ 5 for a caller that asked for the other. */
/* Version 4 changed no struct: it ADDED rayhs_config_reload(), rayhs_scene_capture_tile_order() and rayhs_scene_tile_order(). */
/* Version 5 changed no struct: it ADDED rayhs_selftest_arith(); rayhs_config_reload() became safe beside running entry points. */
#define RAYHS_ABI_VERSION 5

typedef enum {
    RAYHS_OK = 0,
    RAYHS_ERR_INVALID = -1,    /* bad argument / inconsistent scene description */
    RAYHS_ERR_NO_DEVICE = -2,  /* no usable HIP device */
    RAYHS_ERR_HIP = -3,        /* a HIP runtime call failed */
    RAYHS_ERR_IO = -4,         /* file could not be read / written */
    RAYHS_ERR_PARSE = -5,      /* "Failed to read scene" (src/RayHs.hs:226) or OBJ parse error */
    RAYHS_ERR_UNSUPPORTED = -6 /* e.g. tree deeper than the per-lane LDS stack */
} rayhs_status;

/* Geometry constructors in scene order (src/Geometry.hs:62-63, src/KDTree.hs:59-61) */
enum { RAYHS_GEOM_PLANE = 0, RAYHS_GEOM_SPHERE = 1, RAYHS_GEOM_MESH = 2 };
/* Material constructors (src/Material.hs:13-19) */
enum { RAYHS_MAT_MIRROR = 0, RAYHS_MAT_DIFFUSE = 1, RAYHS_MAT_PLASTIC = 2, RAYHS_MAT_EMMIT = 3,
       RAYHS_MAT_TRANSPARENT = 4, RAYHS_MAT_SHOWNORMAL = 5, RAYHS_MAT_SHOWUV = 6 };
/* ColorMap constructors (src/ColorMap.hs:11-16) */
enum { RAYHS_CMAP_FLAT = 0, RAYHS_CMAP_CHECKER = 1, RAYHS_CMAP_TEXTURE = 2 };
/* Light constructors (src/Light.hs:8-10) */
enum { RAYHS_LIGHT_DIRECTIONAL = 0, RAYHS_LIGHT_POINT = 1 };
/* Projection constructors (src/Projection.hs:10-15) */
enum { RAYHS_PROJ_ORTHOGRAPHIC = 0, RAYHS_PROJ_PERSPECTIVE = 1 };

/*
 * Flattened Scene (src/Scene.hs:8-10).  All reals are fp64, as in the
 * reference (src/Vec.hs:29-31).  Vectors are xyz triples, colours rgb triples.
 *
 * Objects keep SCENE ORDER: closest-hit ties are broken by it
 * (minimumBy, src/RayHs.hs:67-71).  obj_geom_index[i] indexes the plane,
 * sphere or mesh arrays according to obj_geom_kind[i].  Material arrays are
 * per object.
 *
 * A mesh is the bounding-box tree of src/KDTree.hs:59-61 in any node order:
 * node k has box node_lo/node_hi[3k..], and either
 *   inner: node_tri_count[k] < 0, children node_left[k] / node_right[k]
 *          (-1 = Empty, src/KDTree.hs:61), node_axis[k] = split axis 0..2, or
 *   leaf : node_tri_count[k] >= 0 triangles starting at node_first_tri[k].
 * Triangles must be stored so that leaves appear in left-to-right (DFS) order
 * with the reference's in-leaf order preserved: tie-breaking between equal t
 * depends on it (src/Geometry.hs:54-57, src/KDTree.hs:113-115).
 * tri_pos: 9 doubles p0,p1,p2; tri_nrm: 9 doubles n0,n1,n2; tri_uv: 6 doubles.
 */
typedef struct rayhs_scene_desc {
    int32_t n_objects;
    const int32_t *obj_geom_kind;   /* RAYHS_GEOM_* */
    const int32_t *obj_geom_index;
    const int32_t *mat_kind;        /* RAYHS_MAT_* */
    const double *mat_ior;          /* mirror / plastic / transparent */
    const double *mat_ce;           /* 3 per object (emmit) */
    const int32_t *cmap_kind;       /* RAYHS_CMAP_* (diffuse / plastic) */
    const double *cmap_c1;          /* 3 per object: flat colour or checker color1 */
    const double *cmap_c2;          /* 3 per object: checker color2 */
    const double *cmap_size;        /* checker size */
    const int32_t *tex_w, *tex_h;   /* texture extent per object (0 if none) */
    const int64_t *tex_offset;      /* first texel (in texels) inside `texels` */
    int64_t n_texels;
    const double *texels;           /* 3 doubles per texel, row-major (src/Bitmap.hs:17-18) */

    int32_t n_planes;
    const double *plane_point, *plane_normal, *plane_tangent; /* 3 each */
    int32_t n_spheres;
    const double *sphere_center;    /* 3 each */
    const double *sphere_radius;

    int32_t n_meshes;
    const int32_t *mesh_root;       /* root node index, -1 = Empty tree */
    int32_t n_nodes;
    const double *node_lo, *node_hi;
    const int32_t *node_left, *node_right, *node_first_tri, *node_tri_count, *node_axis;
    int32_t n_tris;
    const double *tri_pos, *tri_nrm, *tri_uv;

    int32_t n_lights;
    const int32_t *light_kind;      /* RAYHS_LIGHT_* */
    const double *light_vec;        /* direction or position, 3 each */
    const double *light_color;      /* 3 each */
    const double *light_radius;     /* point lights */
} rayhs_scene_desc;

/* Camera (src/Projection.hs:17-20) + Projection (src/Projection.hs:10-15).
 * proj_width/proj_height/near are carried but unused, as in the reference
 * (src/Projection.hs:34). */
typedef struct rayhs_camera {
    double position[3], target[3], up[3];
    int32_t proj_kind;  /* RAYHS_PROJ_* */
    double fovy, proj_width, proj_height, near_;
} rayhs_camera;

/* Output element types. F64 carries the reference's Double colours unchanged;
 * F32 is the 12 B/pixel framebuffer of SURVEY.md 8(d); I16 is toIntC
 * (src/Image.hs:54-55) evaluated on the fp64 value, saturated to int16. */
enum { RAYHS_OUT_F32 = 0, RAYHS_OUT_F64 = 1, RAYHS_OUT_I16 = 2 };

/* flags */
#define RAYHS_FLAG_FMA        1u /* use the kernels built with fused multiply-add (~10 % faster).  Default is the
                                  -ffp-contract=off build, whose fp64 operations are the reference's one for one;
                                  with FMA last bits differ and exact-tie cases (a ray parallel to a plane) can flip */
#define RAYHS_FLAG_EXHAUSTIVE 2u /* visit exactly the reference's node set (no ordering, no pruning, no any-hit exit) */
#define RAYHS_FLAG_COUNT      4u /* instrumented run: fill the work counters of rayhs_stats */
#define RAYHS_FLAG_POW5_SQMUL 8u /* fresnel's x ^ (5 :: Int) (src/Material.hs:26-29) as GHC.Real defines (^): ((x*x)*(x*x))*x.  The
                                  default is the flat product x*x*x*x*x that GHC's RULES "^5/Int" produce under -O (the
                                  reference builds with -O2, build.sh:1).  No GHC was available to confirm which one the
                                  reference binary runs; they differ by an ulp or two of the Fresnel factor */
#define RAYHS_FLAG_ALL (RAYHS_FLAG_FMA | RAYHS_FLAG_EXHAUSTIVE | RAYHS_FLAG_COUNT | RAYHS_FLAG_POW5_SQMUL) /* any other bit: RAYHS_ERR_INVALID */

/* Rendering (src/RayHs.hs:41-45) plus the tiling needed to shard a frame.
 * Rows row0 <= y < row1 are considered; of those this call renders the rows
 * whose stripe index ((y - row0) / stripe_rows) % stripe_count == stripe_rank.
 * Rendered rows are written compactly in increasing y.  stripe_count = 1
 * renders the whole range.  spp = 1 is the reference's live path (rayTrace:
 * pixel-corner rays, no RNG); spp > 1 follows distributedTracePixel
 * (src/RayHs.hs:173-188) with the build-defined counter sampler and `seed`. */
typedef struct rayhs_render_params {
    int32_t width, height, max_depth;
    int32_t spp;
    uint64_t seed;
    int32_t row0, row1;
    int32_t stripe_rows, stripe_rank, stripe_count;
    int32_t out_format; /* RAYHS_OUT_* */
    uint32_t flags;     /* RAYHS_FLAG_* */
} rayhs_render_params;

/* One ray = one scene query of the reference: closestIntersection
 * (src/RayHs.hs:67) or shadowIntersection (src/RayHs.hs:74).  rays_* count the
 * reference's queries; rays actually traced = their sum - shadow_skipped.
 * Work counters are filled only with RAYHS_FLAG_COUNT. */
typedef struct rayhs_stats {
    uint64_t rays_primary, rays_reflect, rays_refract, rays_refract_exit, rays_shadow;
    uint64_t nodes_tested;  /* rayInterBox evaluations (src/KDTree.hs:39) */
    uint64_t tris_tested;   /* triangleIntersection evaluations (src/Mesh.hs:59) */
    uint64_t prims_tested;  /* plane / sphere tests (src/Geometry.hs:68) */
    uint64_t shaded_hits;   /* irradiance evaluations (src/RayHs.hs:107) */
    uint64_t pixels;        /* pixels written */
    double kernel_ms;       /* HIP-event time of the trace kernel of this call (host-buffer entry point only) */
    uint64_t shadow_skipped; /* of rays_shadow: queries of the reference that were NOT traced because the light's diffuse
                              * term was exactly zero, so occluded or not the pixel is the same (ABI version 2) */
} rayhs_stats;

typedef struct rayhs_scene rayhs_scene;

const char *rayhs_last_error(void);
int rayhs_abi_version(void);
/* sizeof, as this library was compiled, of the structs that cross the boundary by pointer; 0 for an unknown id */
enum { RAYHS_SIZEOF_SCENE_DESC = 0, RAYHS_SIZEOF_CAMERA = 1, RAYHS_SIZEOF_RENDER_PARAMS = 2, RAYHS_SIZEOF_STATS = 3 };
uint32_t rayhs_sizeof(int which);
/* number of visible HIP devices (0 if none or the runtime is unusable) */
int rayhs_device_count(void);

/* The library reads its RAYHS_* environment switches ONCE, when it is loaded (never on the launch path): scheduling policy
 * that cannot change a pixel -- RAYHS_SCAN_ORDER, RAYHS_NO_STATIC_ORDER, RAYHS_REORDER_EVERY=n, RAYHS_NO_WAVE_RESOLVE,
 * RAYHS_ROUND_ROBIN_STRIPES, RAYHS_LPT_STRIPES, RAYHS_MULTI_PARALLEL_CREATE (one thread per device at rayhs_multi_create; opt-in),
 * RAYHS_MULTI_SERIAL_CREATE (overrides it: devices brought up one after the other, the default), RAYHS_MULTI_SERIAL -- and the stderr
 * diagnostics RAYHS_DEBUG_COUNTERS, RAYHS_DEBUG_TIMING.  A host that changes its environment afterwards calls this to have them read
 * again; safe while other threads are inside the library (they keep the snapshot of the switches they are reading: the new one is
 * swapped in, never written over the old).  (Development knobs -- stack cap, workgroups per CU, dumps, BVH leaf size --
 * exist only in builds made with -DRAYHS_DEV_KNOBS.)  ABI version 4; safe beside running entry points since version 5. */
void rayhs_config_reload(void);

/* Diagnostic (ABI version 5): the trace kernels take some fp64 divisions and square roots WITHOUT the range fix-ups of the compiler's
 * correctly rounded sequences, where the operands' range makes those fix-ups identities (rayhs_amd/csrc/short_div.h).  This runs both
 * forms on `n` pseudo-random operands per case on `device` and reports, per case, on how many the two differ bit for bit -- zero, or the
 * shortcut is not exact on this hardware / compiler.  Cases: 0  1/x, |x| in [2^-767, 2^1021];  1  1/x, |x| in [2^-20, 2^366];  2  n/d,
 * both in [2^-200, 2^200], and +0 / |d|;  3  sqrt x, x in [2^-399, 2^399];  4  sqrt x, x in [2^-767, 2^1023].  RAYHS_ERR_NO_DEVICE without a GPU. */
#define RAYHS_SELFTEST_ARITH_CASES 5
int rayhs_selftest_arith(int device, uint64_t n, uint64_t seed, uint64_t mismatches[RAYHS_SELFTEST_ARITH_CASES]);

/* Optional: bring up the HIP runtime, the context of `device` and the kernels' code object now, e.g. from a second host
 * thread while the first one is still decoding the scene -- the first rayhs_scene_create of a process otherwise spends
 * 0.1-0.2 s doing exactly that ("end to end").  Thread-safe; calling it more than once, or never,
 * changes nothing but timing.  RAYHS_ERR_NO_DEVICE / RAYHS_ERR_INVALID as for rayhs_scene_create. */
int rayhs_device_prepare(int device);

/* Copies the description into device `device` (HBM-resident until destroy).  Every entry point that takes a
 * scene makes that device current on the calling thread (hipSetDevice) and leaves it current. */
int rayhs_scene_create(const rayhs_scene_desc *desc, int device, rayhs_scene **out);
void rayhs_scene_destroy(rayhs_scene *scene);

/* rows this (params) selects, i.e. the row count of the output buffer */
int rayhs_render_rows(const rayhs_render_params *params);

/* Renders into a HOST buffer of rayhs_render_rows()*width*3 elements of
 * params->out_format; synchronous.  stats may be NULL. */
int rayhs_render(const rayhs_scene *scene, const rayhs_camera *camera,
                 const rayhs_render_params *params, void *out_host, rayhs_stats *stats);

/* Renders into a DEVICE buffer on the scene's device, asynchronously on
 * `hip_stream` (a hipStream_t, NULL = default stream).  Apart from the first
 * call with a new frame geometry (size, rows, stripes, spp, depth), which
 * allocates the launch workspace, there is no host synchronisation and no
 * allocation: safe to enqueue back to back.  Frames of the same geometry
 * schedule their tiles from the chunk times the previous one measured
 * (DESIGN.md section 4); results do not depend on it.  If
 * d_counters is non-NULL it must point to 16 zeroed uint64 on the device and
 * RAYHS_FLAG_COUNT must be set.
 *
 * Streams: a scene handle owns ONE set of launch state (continuation workspace,
 * sample buffer, tile costs and order), so launches of one handle never run
 * side by side.  On one stream they are ordered anyway; when a call names a
 * different stream than the previous call on the same handle (including a
 * switch between rayhs_render, which uses the NULL stream, and a caller
 * stream), the library makes the new stream wait for the previous launch with
 * an event (device-side, no host synchronisation).  To overlap frames, use one
 * handle per stream (bench.py does).  The caller still orders its own use of
 * d_out and d_counters across streams. */
int rayhs_render_device(const rayhs_scene *scene, const rayhs_camera *camera,
                        const rayhs_render_params *params, void *d_out, void *hip_stream,
                        uint64_t *d_counters);

/* Same as rayhs_render_device, and records two caller-owned HIP events (hipEvent_t, may be NULL) on the
 * stream immediately before and after the TRACE kernel alone (not the ticket reset, not the resolve
 * pass), so that a caller can time exactly the kernel a profiler reports.  When the params select no rows
 * nothing is launched and both events are recorded back to back. */
int rayhs_render_device_events(const rayhs_scene *scene, const rayhs_camera *camera,
                               const rayhs_render_params *params, void *d_out, void *hip_stream,
                               uint64_t *d_counters, void *ev_before_trace, void *ev_after_trace);

/* Diagnostics / tests (ABI version 4): the order in which a frame's 8x8 tiles are handed to the waves -- slowest first, from
 * the chunk times earlier frames measured or, for the first frame of a geometry, from a static estimate.
 * Scheduling only: no pixel depends on it.  After rayhs_scene_capture_tile_order(scene, 1) every launch of the handle keeps a
 * copy of the order it was given (a device-to-device copy in stream order); rayhs_scene_tile_order waits for the last launch,
 * copies that order into order[0..cap) (tile indices, row-major over the tile grid) and returns the number of tiles, 0 when
 * the launch ran in scan order (small frames, counting kernels) or nothing was captured; *from_static (may be NULL) = 1 when
 * the order came from the static estimate.  order = NULL only asks for the count. */
int rayhs_scene_capture_tile_order(rayhs_scene *scene, int enable);
int rayhs_scene_tile_order(const rayhs_scene *scene, int32_t *order, int cap, int *from_static);

/* ---- several GPUs of one node, one process -------------------------------
 * What replaces the reference's parallel pixel map (generatePixels over GHC
 * sparks, src/Image.hs:34-36): the rows [row0, row1) are cut into stripes of
 * params->stripe_rows rows (8 if < 1), every listed device traces its share of
 * them from its own copy of the scene, and pushes the finished stripes into
 * the whole frame on devices[0] with peer writes (xGMI).  No collective on the
 * data path: stripes are independent.  Ownership of the stripes is balanced by
 * a static cost estimate of the frame (the model sits in the middle rows), not
 * dealt round-robin; params->stripe_rank / stripe_count are ignored.  The
 * frame is bit-identical to the single-device one.  `devices` may name a
 * device more than once (a one-GPU rehearsal of the N-GPU path).  The same
 * one-thread-at-a-time rule as for scene handles applies. */
typedef struct rayhs_multi rayhs_multi;
int rayhs_multi_create(const rayhs_scene_desc *desc, const int *devices, int n_devices, rayhs_multi **out);
void rayhs_multi_destroy(rayhs_multi *multi);
int rayhs_multi_device_count(const rayhs_multi *multi);
/* Whole frame (rows row0..row1) into a HOST buffer of (row1-row0)*width*3 elements; synchronous.
 * stats (may be NULL): counters summed over the devices, kernel_ms = the slowest device's trace kernel. */
int rayhs_multi_render(rayhs_multi *multi, const rayhs_camera *camera, const rayhs_render_params *params,
                       void *out_host, rayhs_stats *stats);
/* Same into a DEVICE buffer that lives on devices[0]; synchronous. */
int rayhs_multi_render_device(rayhs_multi *multi, const rayhs_camera *camera, const rayhs_render_params *params,
                              void *d_out, rayhs_stats *stats);
/* After a render: per device the estimated share of the frame's cost (0..1), the rows it was given and (if that render
 * asked for stats) its trace-kernel time.  Each array holds n >= device count entries or is NULL. */
int rayhs_multi_shares(const rayhs_multi *multi, double *shares, int *rows, double *kernel_ms, int n);

/* ---- host preparation (C++ restatement of the reference's L3/L4 host code;
 * the Haskell host would do this itself and call rayhs_scene_create) -------- */

/* RenderDesc (src/Descriptors.hs:19-23) decoded from JSON (src/JSON.hs:22-141),
 * with the Scene built (src/Descriptors.hs:39-55: readOBJ, Mesh.transform,
 * buildKDTree) and flattened.  Mesh/texture paths are opened relative to the
 * current working directory, as in the reference (src/Mesh.hs:218-221). */
typedef struct rayhs_render_desc rayhs_render_desc;
int rayhs_render_desc_load(const char *json_path, rayhs_render_desc **out);
/* same, from a JSON string */
int rayhs_render_desc_parse(const char *json_text, rayhs_render_desc **out);
void rayhs_render_desc_free(rayhs_render_desc *rd);
const rayhs_scene_desc *rayhs_render_desc_scene(const rayhs_render_desc *rd);
const rayhs_camera *rayhs_render_desc_camera(const rayhs_render_desc *rd);
int rayhs_render_desc_width(const rayhs_render_desc *rd);
int rayhs_render_desc_height(const rayhs_render_desc *rd);
int rayhs_render_desc_max_depth(const rayhs_render_desc *rd);
/* deepest tree level over all meshes (root = 0), -1 if there are none */
int rayhs_render_desc_tree_depth(const rayhs_render_desc *rd);

/* writePPM (src/Image.hs:60-75): ASCII P3, "r g b" + two spaces per pixel, rows
 * joined by '\n', no trailing newline.  `q` holds toIntC values (RAYHS_OUT_I16). */
int rayhs_write_ppm_i16(const char *path, int width, int height, const int16_t *q);
/* same from fp64 colours, quantised with toIntC on the host */
int rayhs_write_ppm_f64(const char *path, int width, int height, const double *rgb);

/* the build-defined sampler: u(seed, pixel, sample, dim) in [0,1) */
double rayhs_sample01(uint64_t seed, uint64_t pixel, uint32_t sample, uint32_t dim, uint32_t spp);

#ifdef __cplusplus
}
#endif
#endif /* RAYHS_H */
