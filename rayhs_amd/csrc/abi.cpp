// abi.cpp -- implementation of the C ABI declared in include/rayhs.h:
// scene upload into HBM, kernel launch, host-prep wrappers.  No CPU fallback:
// every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rayhs.h"
#include "bvh_build.hpp"
#include "device_scene.h"
#include "host_scene.hpp"

namespace rayhs {
hipError_t launch_trace_fast(int variant, int grid, size_t lds_bytes, hipStream_t stream, const DScene &sc, const DCamera &cam,
                             const DParams &prm, void *out, unsigned long long *counters, unsigned long long *ticket, double *frames);
hipError_t launch_resolve_fast(hipStream_t stream, const double *sbuf, void *out, int n_pixels, int spp, int out_format);
hipError_t launch_resolve_exact(hipStream_t stream, const double *sbuf, void *out, int n_pixels, int spp, int out_format);
hipError_t launch_order_tiles_fast(hipStream_t stream, float *cost, int n, int *order, unsigned *aux, unsigned long long *ticket_ring, int first, int count, int rings, int ticket_slots);
hipError_t launch_scatter_stripes_fast(hipStream_t stream, const void *src, void *dst, const int *stripe_map, int n_rows, int stripe_rows, long long row_bytes);
hipError_t launch_scatter_stripes_exact(hipStream_t stream, const void *src, void *dst, const int *stripe_map, int n_rows, int stripe_rows, long long row_bytes);
hipError_t launch_static_costs_fast(hipStream_t stream, const DScene &sc, const DCamera &cam, const DParams &prm, float *cost);
hipError_t launch_static_costs_exact(hipStream_t stream, const DScene &sc, const DCamera &cam, const DParams &prm, float *cost);
hipError_t launch_order_tiles_exact(hipStream_t stream, float *cost, int n, int *order, unsigned *aux, unsigned long long *ticket_ring, int first, int count, int rings, int ticket_slots);
hipError_t launch_trace_exact(int variant, int grid, size_t lds_bytes, hipStream_t stream, const DScene &sc, const DCamera &cam,
                              const DParams &prm, void *out, unsigned long long *counters, unsigned long long *ticket, double *frames);
hipError_t launch_warmup_exact(hipStream_t stream);
hipError_t launch_selftest_arith_exact(hipStream_t stream, unsigned long long *bad, unsigned long long n, unsigned long long seed);
int occupancy_trace_fast(int variant, size_t lds_bytes);
int occupancy_trace_exact(int variant, size_t lds_bytes);
int stack_entry_bytes_fast(int variant);
int stack_entry_bytes_exact(int variant);
}  // namespace rayhs

using namespace rayhs;

static thread_local std::string g_error;
static int set_error(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t e_ = (expr);                                                                               \
        if (e_ != hipSuccess) return set_error(RAYHS_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// Every RAYHS_* environment switch of the library, read ONCE -- when the library is loaded, and again only when the host asks
// (rayhs_config_reload, include/rayhs.h).  Nothing on the launch path calls getenv: a 0.17 ms Cornell step used to make about
// fifteen such calls, and getenv races with any other thread's setenv.  Three groups:
//   * policy switches a host may legitimately set (scheduling only, never a pixel): scan order, no static first-frame order,
//     the sample-buffer resolve, stripe ownership, one thread per device at rayhs_multi_create;
//   * diagnostics that only print (RAYHS_DEBUG_COUNTERS, RAYHS_DEBUG_TIMING);
//   * development knobs -- some of them UNSAFE (a stack cap below the tree depth overruns the per-lane LDS stack) -- which exist
//     only in builds made with -DRAYHS_DEV_KNOBS (the experiment builds under build_exp/); the shipped library has none of them.
struct Config {
    bool no_wave_resolve = false, scan_order = false, no_static_order = false;
    bool multi_parallel_create = false, multi_serial_create = false, round_robin_stripes = false, lpt_stripes = false, multi_serial = false;
    bool debug_counters = false, debug_timing = false;
    long long reorder_every = 4;
#ifdef RAYHS_DEV_KNOBS
    bool term_in_workspace = false, warm_copy = false, warm_launch = false, debug_sync_first = false, dump_raw_feature = false;
    int stack_cap = 0, blocks_per_cu = 0;
    long long freeze_order_after = -1;
    std::string dump_tile_costs, dump_wave_ends;
#endif
};
static Config read_config() {
    Config c;
    auto on = [](const char *name) { return std::getenv(name) != nullptr; };
    c.no_wave_resolve = on("RAYHS_NO_WAVE_RESOLVE"); c.scan_order = on("RAYHS_SCAN_ORDER"); c.no_static_order = on("RAYHS_NO_STATIC_ORDER");
    c.multi_parallel_create = on("RAYHS_MULTI_PARALLEL_CREATE"); c.multi_serial_create = on("RAYHS_MULTI_SERIAL_CREATE");
    c.round_robin_stripes = on("RAYHS_ROUND_ROBIN_STRIPES"); c.lpt_stripes = on("RAYHS_LPT_STRIPES"); c.multi_serial = on("RAYHS_MULTI_SERIAL");
    c.debug_counters = on("RAYHS_DEBUG_COUNTERS"); c.debug_timing = on("RAYHS_DEBUG_TIMING");
    if (const char *e = std::getenv("RAYHS_REORDER_EVERY")) c.reorder_every = std::max(1LL, std::atoll(e));
#ifdef RAYHS_DEV_KNOBS
    c.term_in_workspace = on("RAYHS_TERM_IN_WORKSPACE"); c.warm_copy = on("RAYHS_WARM_COPY"); c.warm_launch = on("RAYHS_WARM_LAUNCH");
    c.debug_sync_first = on("RAYHS_DEBUG_SYNC_FIRST"); c.dump_raw_feature = on("RAYHS_DUMP_RAW_FEATURE");
    if (const char *e = std::getenv("RAYHS_STACK_CAP")) c.stack_cap = std::atoi(e);
    if (const char *e = std::getenv("RAYHS_BLOCKS_PER_CU")) c.blocks_per_cu = std::atoi(e);
    if (const char *e = std::getenv("RAYHS_FREEZE_ORDER_AFTER")) c.freeze_order_after = std::atoll(e);
    if (const char *e = std::getenv("RAYHS_DUMP_TILE_COSTS")) c.dump_tile_costs = e;
    if (const char *e = std::getenv("RAYHS_DUMP_WAVE_ENDS")) c.dump_wave_ends = e;
#endif
    return c;
}
// The switches live in an immutable snapshot behind a shared pointer that rayhs_config_reload() SWAPS (atomically): an entry point that is
// running on another thread -- a rayhs_multi worker, say -- keeps reading the snapshot it took, whole and alive, while the next mention
// sees the new one.  (It was one plain global struct that reload overwrote in place under every reader.)  Every `g_cfg.x` below is
// a read through a snapshot held for the length of that expression; strings are copied out before they are used.
static std::shared_ptr<const Config> g_cfg_snapshot = std::make_shared<const Config>(read_config());
static std::shared_ptr<const Config> config_snapshot() { return std::atomic_load(&g_cfg_snapshot); }
#define g_cfg (*config_snapshot())
extern "C" void rayhs_config_reload(void) { std::atomic_store(&g_cfg_snapshot, std::shared_ptr<const Config>(std::make_shared<const Config>(read_config()))); }

// uint64 per launch: 8 range counters, each alone on its own 128-byte L2 line (stride 16), then -- on a 128-byte line of their own --
// the 8 "range is dry" flags.  (At a 64-byte stride two counters shared an L2 line, and so did the flags, which are read at
// workgroup scope, with a counter that takes device-scope atomics in the same kernel.)  144 * 8 B = 1152 B = 9 lines per launch.
static constexpr int kTicketSlots = 8 * TICKET_STRIDE + 16;  // (TICKET_STRIDE: device_scene.h)
static constexpr int kTicketRings = 8;    // launches in flight may overlap on different streams
static constexpr int kTicketsAhead = 4;   // launches whose counters one ordering kernel zeroes ahead (< kTicketRings - frames in flight)
static_assert(kTicketsAhead < kTicketRings, "an ordering kernel must never zero the counters of the launch it runs behind");

struct rayhs_scene {
    int device = 0;
    int cu_count = 0;
    DScene d{};
    std::vector<void *> allocs;
    int tree_depth = -1;  // deepest level of the reference trees over all meshes
    bool huge_coords = false;  // a vertex coordinate beyond 1e90: every frame of this scene goes through the exhaustive kernel (prepare_launch)
    int fast_depth = 0;   // deepest inner node of the fast BVHs
    // launch workspace (grow-only; steady-state renders allocate nothing)
    mutable double *frames = nullptr;
    mutable size_t frames_bytes = 0;
    mutable unsigned long long *tickets = nullptr;
    mutable double *sbuf = nullptr;  // per-sample radiance of the frame in flight (spp > 1)
    mutable size_t sbuf_bytes = 0;
    mutable unsigned next_ticket = 0;
    mutable unsigned long long *counters = nullptr;
    mutable int grid_cache[2][4][2] = {{{0}}};  // [exact][variant] -> {lds_bytes, blocks per CU}
    // feedback scheduling (plan_tile_order): chunk times measured by a frame order the tiles of the next one
    mutable int32_t *order_dev = nullptr;
    mutable float *cost_dev = nullptr;
    mutable unsigned *order_aux = nullptr;       // the ordering kernels' histogram rows (ORDER_MAX_BLOCKS x 1024)
    mutable long long ticket_clean_lo = 0, ticket_clean_hi = -1;  // launch numbers whose ticket counters an ordering kernel behind an earlier launch has already zeroed
    mutable long long frames_on_key = 0;         // launches since the frame geometry (order_key) last changed
    mutable size_t order_cap = 0;
    mutable bool order_valid = false;            // order_dev holds an order for order_key
    mutable int order_key[10] = {0};             // frame geometry the order and the costs belong to
    // One set of launch state (continuation workspace, sample buffer, tile costs / order) per handle: launches of one
    // handle must not overlap.  On one stream they cannot; when the caller switches streams the new stream is made to
    // wait for the previous launch (include/rayhs.h, "streams").
    mutable hipStream_t last_stream = nullptr;
    mutable hipEvent_t last_done = nullptr;      // recorded behind the last kernel of every launch
    mutable bool have_last = false;
    // rayhs_render (host-buffer entry point): device frame and timing events kept between calls
    mutable void *host_out_dev = nullptr;
    mutable size_t host_out_bytes = 0;
    mutable hipEvent_t host_e0 = nullptr, host_e1 = nullptr;
    // rayhs_scene_capture_tile_order / rayhs_scene_tile_order (tests, diagnostics): the work order the last launch was given
    bool capture_order = false;
    mutable int32_t *captured_dev = nullptr;
    mutable size_t captured_cap = 0;
    mutable int captured_tiles = 0, captured_static = 0;
};

extern "C" const char *rayhs_last_error(void) { return g_error.c_str(); }
extern "C" int rayhs_abi_version(void) { return RAYHS_ABI_VERSION; }
extern "C" uint32_t rayhs_sizeof(int which) {
    switch (which) {
    case RAYHS_SIZEOF_SCENE_DESC: return (uint32_t)sizeof(rayhs_scene_desc);
    case RAYHS_SIZEOF_CAMERA: return (uint32_t)sizeof(rayhs_camera);
    case RAYHS_SIZEOF_RENDER_PARAMS: return (uint32_t)sizeof(rayhs_render_params);
    case RAYHS_SIZEOF_STATS: return (uint32_t)sizeof(rayhs_stats);
    default: return 0;
    }
}
extern "C" int rayhs_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int rayhs_selftest_arith(int device, uint64_t n, uint64_t seed, uint64_t mismatches[RAYHS_SELFTEST_ARITH_CASES]) {
    if (!mismatches || n == 0 || n > (1ULL << 32)) return set_error(RAYHS_ERR_INVALID, "rayhs_selftest_arith: n must be in [1, 2^32] and mismatches not NULL");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 1) return set_error(RAYHS_ERR_NO_DEVICE, "no HIP device");
    if (device < 0 || device >= count) return set_error(RAYHS_ERR_INVALID, "device %d out of range (%d visible)", device, count);
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d = nullptr;
    HIP_TRY(hipMalloc((void **)&d, sizeof(uint64_t) * RAYHS_SELFTEST_ARITH_CASES));
    hipError_t e = hipMemset(d, 0, sizeof(uint64_t) * RAYHS_SELFTEST_ARITH_CASES);
    if (e == hipSuccess) e = launch_selftest_arith_exact(nullptr, d, n, seed);
    if (e == hipSuccess) e = hipMemcpy(mismatches, d, sizeof(uint64_t) * RAYHS_SELFTEST_ARITH_CASES, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) return set_error(RAYHS_ERR_HIP, "rayhs_selftest_arith failed: %s", hipGetErrorString(e));
    return RAYHS_OK;
}

extern "C" int rayhs_device_prepare(int device) {
    using clk = std::chrono::steady_clock;
    const bool verbose = g_cfg.debug_timing;  // diagnostics: where a process's HIP start-up goes
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const clk::time_point t0 = clk::now();
    int ndev = rayhs_device_count();                 // the first HIP call of the process: hipInit
    const clk::time_point t1 = clk::now();
    if (ndev <= 0) return set_error(RAYHS_ERR_NO_DEVICE, "no HIP device available (librayhs_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return set_error(RAYHS_ERR_INVALID, "device %d out of range (have %d)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr));                       // creates the context
    // What a process pays once besides hipInit (measured, profiles/r04/cli_startup.md): ~20 ms at its first host -> device copy
    // (the runtime's copy path), ~8 ms at the first copy back, ~20 ms at its first launch.  The occupancy query below resolves
    // this library's kernels in < 1 ms: the 600 KB code object is NOT what start-up waits for.  Paying the copy / launch costs
    // here instead (RAYHS_WARM_COPY / RAYHS_WARM_LAUNCH, development builds only) moved them without shortening a one-shot run.
    const clk::time_point t2 = clk::now();
    double ms_copy1 = 0, ms_copy2 = 0;
#ifdef RAYHS_DEV_KNOBS
    if (g_cfg.warm_copy) {   // (development: measured, it moves the cost without shortening the run -- profiles/r04/cli_startup.md)
        void *q = nullptr; char buf[64] = {0};
        if (hipMalloc(&q, 64) == hipSuccess) {
            (void)hipMemcpy(q, buf, 64, hipMemcpyHostToDevice);
            const clk::time_point c1 = clk::now();
            (void)hipMemcpy(buf, q, 64, hipMemcpyDeviceToHost);
            (void)hipFree(q);
            ms_copy1 = ms(t2, c1); ms_copy2 = ms(c1, clk::now());
        }
    }
#endif
    const clk::time_point t3 = clk::now();
    (void)occupancy_trace_exact(0, 40 * 1024);
#ifdef RAYHS_DEV_KNOBS
    if (g_cfg.warm_launch) { (void)launch_warmup_exact(nullptr); (void)hipStreamSynchronize(nullptr); }  // (development, as above)
#endif
    const clk::time_point t4 = clk::now();
    if (verbose)
        std::fprintf(stderr, "rayhs_device_prepare(%d): hipInit (first HIP call) %.1f ms, context %.1f ms, first 64-byte copy to the device %.1f ms and back %.1f ms, "
                             "first launch (queue + code object of the exact kernels) %.1f ms\n", device, ms(t0, t1), ms(t1, t2), ms_copy1, ms_copy2, ms(t3, t4));
    return RAYHS_OK;
}

// ---------------------------------------------------------------- validation
static int check_desc(const rayhs_scene_desc *d, int &tree_depth) {
    if (!d) return set_error(RAYHS_ERR_INVALID, "scene description is NULL");
    if (d->n_objects < 0 || d->n_planes < 0 || d->n_spheres < 0 || d->n_meshes < 0 || d->n_nodes < 0 || d->n_tris < 0 || d->n_lights < 0)
        return set_error(RAYHS_ERR_INVALID, "negative count in scene description");
    // the kernels address records as base + 32-bit byte offset (largest record: 128 B per triangle, 64 B per node)
    if (d->n_tris > (1 << 25) || d->n_nodes > (1 << 25))
        return set_error(RAYHS_ERR_INVALID, "scene too large: at most 2^25 triangles and tree nodes are supported");
    // (the shading state keeps a hit's material kind in bits 28..30 of its object index)
    if (d->n_objects > (1 << 25)) return set_error(RAYHS_ERR_INVALID, "scene too large: at most 2^25 objects are supported");
    // an array may be NULL only when its count is zero
    struct { const void *p; int64_t n; const char *name; } arrays[] = {
        {d->obj_geom_kind, d->n_objects, "obj_geom_kind"}, {d->obj_geom_index, d->n_objects, "obj_geom_index"},
        {d->mat_kind, d->n_objects, "mat_kind"}, {d->mat_ior, d->n_objects, "mat_ior"}, {d->mat_ce, d->n_objects, "mat_ce"},
        {d->cmap_kind, d->n_objects, "cmap_kind"}, {d->cmap_c1, d->n_objects, "cmap_c1"}, {d->cmap_c2, d->n_objects, "cmap_c2"},
        {d->cmap_size, d->n_objects, "cmap_size"}, {d->texels, d->n_texels, "texels"},
        {d->plane_point, d->n_planes, "plane_point"}, {d->plane_normal, d->n_planes, "plane_normal"}, {d->plane_tangent, d->n_planes, "plane_tangent"},
        {d->sphere_center, d->n_spheres, "sphere_center"}, {d->sphere_radius, d->n_spheres, "sphere_radius"},
        {d->mesh_root, d->n_meshes, "mesh_root"}, {d->node_lo, d->n_nodes, "node_lo"}, {d->node_hi, d->n_nodes, "node_hi"},
        {d->node_left, d->n_nodes, "node_left"}, {d->node_right, d->n_nodes, "node_right"}, {d->node_first_tri, d->n_nodes, "node_first_tri"},
        {d->node_tri_count, d->n_nodes, "node_tri_count"}, {d->node_axis, d->n_nodes, "node_axis"},
        {d->tri_pos, d->n_tris, "tri_pos"}, {d->tri_nrm, d->n_tris, "tri_nrm"}, {d->tri_uv, d->n_tris, "tri_uv"},
        {d->light_kind, d->n_lights, "light_kind"}, {d->light_vec, d->n_lights, "light_vec"}, {d->light_color, d->n_lights, "light_color"},
        {d->light_radius, d->n_lights, "light_radius"}};
    for (const auto &a : arrays)
        if (a.n > 0 && a.p == nullptr) return set_error(RAYHS_ERR_INVALID, "%s is NULL but its count is %lld", a.name, (long long)a.n);
    if (d->n_texels < 0) return set_error(RAYHS_ERR_INVALID, "negative count in scene description");
    for (int i = 0; i < d->n_objects; i++) {
        int k = d->obj_geom_kind[i], g = d->obj_geom_index[i];
        int lim = k == RAYHS_GEOM_PLANE ? d->n_planes : k == RAYHS_GEOM_SPHERE ? d->n_spheres : k == RAYHS_GEOM_MESH ? d->n_meshes : -1;
        if (lim < 0 || g < 0 || g >= lim) return set_error(RAYHS_ERR_INVALID, "object %d: bad geometry kind/index (%d, %d)", i, k, g);
        if (d->mat_kind[i] < 0 || d->mat_kind[i] > RAYHS_MAT_SHOWUV) return set_error(RAYHS_ERR_INVALID, "object %d: bad material kind", i);
        if (d->cmap_kind[i] < 0 || d->cmap_kind[i] > RAYHS_CMAP_TEXTURE) return set_error(RAYHS_ERR_INVALID, "object %d: bad colour map kind", i);
        if (d->cmap_kind[i] == RAYHS_CMAP_TEXTURE && (d->mat_kind[i] == RAYHS_MAT_DIFFUSE || d->mat_kind[i] == RAYHS_MAT_PLASTIC)) {
            if (!d->tex_w || !d->tex_h || !d->tex_offset) return set_error(RAYHS_ERR_INVALID, "object %d: texture colour map but tex_w / tex_h / tex_offset is NULL", i);
            int64_t n = (int64_t)d->tex_w[i] * d->tex_h[i];
            if (d->tex_w[i] <= 0 || d->tex_h[i] <= 0 || d->tex_offset[i] < 0 || d->tex_offset[i] + n > d->n_texels)
                return set_error(RAYHS_ERR_INVALID, "object %d: texture outside the texel array", i);
        }
    }
    // trees: indices in range, every node reachable at most once (no cycles), depth measured
    std::vector<int> depth((size_t)d->n_nodes, -1);
    tree_depth = -1;
    for (int m = 0; m < d->n_meshes; m++) {
        int root = d->mesh_root[m];
        if (root == -1) continue;
        if (root < 0 || root >= d->n_nodes) return set_error(RAYHS_ERR_INVALID, "mesh %d: root out of range", m);
        std::vector<int> todo{root};
        if (depth[(size_t)root] >= 0) return set_error(RAYHS_ERR_INVALID, "mesh %d: root shared with another tree", m);
        depth[(size_t)root] = 0;
        while (!todo.empty()) {
            int k = todo.back();
            todo.pop_back();
            tree_depth = std::max(tree_depth, depth[(size_t)k]);
            if (d->node_tri_count[k] >= 0) {
                if (d->node_first_tri[k] < 0 || (int64_t)d->node_first_tri[k] + d->node_tri_count[k] > d->n_tris)
                    return set_error(RAYHS_ERR_INVALID, "node %d: triangle range out of bounds", k);
                continue;
            }
            if (d->node_axis[k] < 0 || d->node_axis[k] > 2) return set_error(RAYHS_ERR_INVALID, "node %d: bad split axis", k);
            for (int c : {d->node_left[k], d->node_right[k]}) {
                if (c == -1) continue;
                if (c < 0 || c >= d->n_nodes) return set_error(RAYHS_ERR_INVALID, "node %d: child out of range", k);
                if (depth[(size_t)c] >= 0) return set_error(RAYHS_ERR_INVALID, "node %d: child %d reached twice", k, c);
                depth[(size_t)c] = depth[(size_t)k] + 1;
                todo.push_back(c);
            }
        }
    }
    for (int i = 0; i < d->n_lights; i++)
        if (d->light_kind[i] != RAYHS_LIGHT_DIRECTIONAL && d->light_kind[i] != RAYHS_LIGHT_POINT)
            return set_error(RAYHS_ERR_INVALID, "light %d: bad kind", i);
    return RAYHS_OK;
}

static thread_local double g_ms_malloc = 0, g_ms_copy = 0;  // development (RAYHS_DEBUG_TIMING)
template <class T>
static int upload(rayhs_scene *s, const std::vector<T> &v, const T **out) {
    using clk = std::chrono::steady_clock;
    *out = nullptr;
    size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    void *p = nullptr;
    const clk::time_point t0 = clk::now();
    HIP_TRY(hipMalloc(&p, bytes));
    const clk::time_point t1 = clk::now();
    s->allocs.push_back(p);
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    g_ms_malloc += std::chrono::duration<double, std::milli>(t1 - t0).count();
    g_ms_copy += std::chrono::duration<double, std::milli>(clk::now() - t1).count();
    *out = (const T *)p;
    return RAYHS_OK;
}

static double r0(double n1, double n2) { double q = (n1 - n2) / (n1 + n2); return q * q; }  // src/Material.hs:22-24

// Everything rayhs_scene_create derives from a description before any device is touched: validation, the device
// records of device_scene.h, triangle ranks, the fast BVH.  Packed ONCE; rayhs_multi_create uploads the same pack to
// every device it lists (the host preparation used to be repeated per device, serially).
struct PackedScene {
    std::vector<DNode> nodes;
    std::vector<DTri> tris, ftris;
    std::vector<DTriAttr> attrs, fattrs;
    std::vector<FNode> fnodes;
    std::vector<DPlane> planes;
    std::vector<DSphere> spheres;
    std::vector<DMesh> meshes;
    std::vector<DObject> objects;
    std::vector<DLight> lights;
    std::vector<double> texels;
    int plane_axis_end[3] = {0, 0, 0};
    int n_objects = 0, n_lights = 0, n_nodes = 0, n_tris = 0;
    int tree_depth = -1, fast_depth = 0;
    bool huge_coords = false;   // some vertex coordinate is finite and beyond 1e90 in magnitude (see prepare_launch)
};

static int pack_scene(const rayhs_scene_desc *desc, PackedScene &pk) {
    int tree_depth = -1;
    int rc = check_desc(desc, tree_depth);
    if (rc) return rc;
    // pack the SoA description into the device records of device_scene.h
    const rayhs_scene_desc &d = *desc;
    std::vector<DNode> &nodes = pk.nodes; nodes.assign((size_t)d.n_nodes, DNode());
    for (int k = 0; k < d.n_nodes; k++) {
        DNode &n = nodes[(size_t)k];
        for (int c = 0; c < 3; c++) { n.lo[c] = d.node_lo[3 * k + c]; n.hi[c] = d.node_hi[3 * k + c]; }
        if (d.node_tri_count[k] >= 0) { n.a = d.node_first_tri[k]; n.b = d.node_tri_count[k]; n.flags = DNODE_LEAF; }
        else { n.a = d.node_left[k]; n.b = d.node_right[k]; n.flags = d.node_axis[k] & 3; }
        n.parent = -1;
    }
    for (int k = 0; k < d.n_nodes; k++)
        if (d.node_tri_count[k] < 0)
            for (int c : {d.node_left[k], d.node_right[k]})
                if (c >= 0) nodes[(size_t)c].parent = k;
    auto pack_tri = [&](int k, DTri &t, DTriAttr &a) {
        const double *p = d.tri_pos + 9 * (size_t)k;
        for (int c = 0; c < 3; c++) {
            t.p0[c] = p[c];
            t.e1[c] = p[3 + c] - p[c];  // e1 = p1 - p0, e2 = p2 - p0 (src/Mesh.hs:65-66)
            t.e2[c] = p[6 + c] - p[c];
        }
        const double *n = d.tri_nrm + 9 * (size_t)k, *uv = d.tri_uv + 6 * (size_t)k;
        for (int c = 0; c < 3; c++) { a.n0[c] = n[c]; a.n1[c] = n[3 + c]; a.n2[c] = n[6 + c]; }
        for (int c = 0; c < 2; c++) { a.uv0[c] = uv[c]; a.uv1[c] = uv[2 + c]; a.uv2[c] = uv[4 + c]; }
        a.leaf_node = -1; a.pad = 0;
    };
    std::vector<DTri> &tris = pk.tris; tris.assign((size_t)d.n_tris, DTri());
    std::vector<DTriAttr> &attrs = pk.attrs; attrs.assign((size_t)d.n_tris, DTriAttr());
    for (int k = 0; k < d.n_tris; k++) {
        pack_tri(k, tris[(size_t)k], attrs[(size_t)k]);
        tris[(size_t)k].rank = k; tris[(size_t)k].leaf_first = k;  // overwritten for triangles reachable from a tree
    }
    // per mesh: the reference's left-to-right leaf order gives every triangle its rank (exact-t ties are
    // resolved by it), then the fast BVH is built over the same triangles and they are re-ordered into its leaves
    std::vector<FNode> &fnodes = pk.fnodes;
    std::vector<DTri> &ftris = pk.ftris;
    std::vector<DTriAttr> &fattrs = pk.fattrs;
    fnodes.clear(); ftris.clear(); fattrs.clear();
    std::vector<int> mesh_froot((size_t)d.n_meshes, FNODE_NONE), mesh_tri_base((size_t)d.n_meshes, 0);
    std::vector<double> mesh_center(3 * (size_t)d.n_meshes, 0.0);
    std::vector<float> mesh_fbox(6 * (size_t)d.n_meshes, 0.0f);
    int fast_depth = 0;
    for (int m = 0; m < d.n_meshes; m++) {
        std::vector<int> ids;       // triangle indices in reference DFS order
        std::vector<int> todo;
        if (d.mesh_root[m] >= 0) todo.push_back(d.mesh_root[m]);
        while (!todo.empty()) {
            int k = todo.back();
            todo.pop_back();
            if (d.node_tri_count[k] >= 0) {
                int first_rank = (int)ids.size();
                for (int i = 0; i < d.node_tri_count[k]; i++) {
                    int t = d.node_first_tri[k] + i;
                    tris[(size_t)t].rank = (int)ids.size();
                    tris[(size_t)t].leaf_first = first_rank;
                    attrs[(size_t)t].leaf_node = k;
                    ids.push_back(t);
                }
            } else {
                if (d.node_right[k] >= 0) todo.push_back(d.node_right[k]);
                if (d.node_left[k] >= 0) todo.push_back(d.node_left[k]);
            }
        }
        FastBvh bvh;
        build_fast_bvh(d.tri_pos, ids, bvh);
        fast_depth = std::max(fast_depth, bvh.depth);
        const int node_base = (int)fnodes.size(), tri_base = (int)ftris.size();
        mesh_tri_base[(size_t)m] = tri_base;
        double ext = 0, ctr[3];
        for (int c = 0; c < 3; c++) { ctr[c] = 0.5 * (bvh.lo[c] + bvh.hi[c]); ext = std::max(ext, bvh.hi[c] - bvh.lo[c]); mesh_center[3 * (size_t)m + (size_t)c] = ctr[c]; }
        // mesh bounds relative to the mesh centre, rounded outward and inflated by 2^-18 of the mesh extent
        // (64x the fp32 rounding error of a ray origin inside them, trace_kernel.hip): root cull
        const double infl = std::max(ext, 1e-300) * (1.0 / 262144.0);
        float *mf = &mesh_fbox[6 * (size_t)m];  // flo[3], fhi[3]
        for (int c = 0; c < 3; c++) {
            mf[c] = round_down_f32((bvh.lo[c] - ctr[c]) - infl);
            mf[3 + c] = round_up_f32((bvh.hi[c] - ctr[c]) + infl);
        }
        // node boxes: mesh-centred fp32, inflated like the mesh bounds and rounded outward; the inflation covers every
        // fp32 rounding of the kernel's slab test (origin, reciprocal direction, one fused multiply-add per plane)
        auto fbox32 = [&](const double *lo, const double *hi, float *b) {
            for (int c = 0; c < 3; c++) {
                // (every child has a box: bvh_build.cpp gives the missing sibling of a one-leaf mesh that leaf's box and no
                // triangles; the kernel's box test relies on finite planes, so anything else is clamped to a point)
                const bool ok = lo[c] <= hi[c] && std::isfinite(lo[c]) && std::isfinite(hi[c]);
                b[2 * c] = ok ? round_down_f32((lo[c] - ctr[c]) - infl) : 0.0f;
                b[2 * c + 1] = ok ? round_up_f32((hi[c] - ctr[c]) + infl) : 0.0f;
            }
        };
        // inner references index the global node array; leaf references ~((first << 3) | count) index the
        // global BVH-ordered triangle array
        auto child = [&](int c) { return c == FNODE_NONE || c == FNODE_EMPTY ? FNODE_EMPTY : c >= 0 ? c + node_base : ~((((~c) >> 3) + tri_base) << 3 | ((~c) & 7)); };
        for (const FNodeBuild &nb : bvh.nodes) {
            FNode fn;
            std::memset(&fn, 0, sizeof fn);
            fbox32(nb.lo0, nb.hi0, fn.b0);
            fbox32(nb.lo1, nb.hi1, fn.b1);
            fn.c0 = child(nb.c0); fn.c1 = child(nb.c1);
            fnodes.push_back(fn);
        }
        if (bvh.root != FNODE_NONE) mesh_froot[(size_t)m] = bvh.root + node_base;
        for (int pos : bvh.order) {
            int t = ids[(size_t)pos];
            ftris.push_back(tris[(size_t)t]);
            fattrs.push_back(attrs[(size_t)t]);
        }
    }
    std::vector<DPlane> &planes = pk.planes; planes.assign((size_t)d.n_planes, DPlane());
    std::vector<DSphere> &spheres = pk.spheres; spheres.assign((size_t)d.n_spheres, DSphere());
    std::vector<DMesh> &meshes = pk.meshes; meshes.assign((size_t)d.n_meshes, DMesh());
    std::vector<DObject> &objects = pk.objects; objects.assign((size_t)d.n_objects, DObject());
    std::vector<char> plane_used((size_t)d.n_planes, 0), sphere_used((size_t)d.n_spheres, 0), mesh_used((size_t)d.n_meshes, 0);
    for (int i = 0; i < d.n_objects; i++) {
        DObject &o = objects[(size_t)i];
        std::memset(&o, 0, sizeof o);
        o.mat_kind = d.mat_kind[i];
        o.cmap_kind = d.cmap_kind[i];
        o.tex_w = d.tex_w ? d.tex_w[i] : 0;
        o.tex_h = d.tex_h ? d.tex_h[i] : 0;
        o.tex_off = d.tex_offset ? d.tex_offset[i] : 0;
        o.ior = d.mat_ior[i];
        o.fr = r0(1.0, o.ior);
        o.one_m_r0 = 1 - r0(o.ior, 1.0);
        for (int c = 0; c < 3; c++) { o.ce[c] = d.mat_ce[3 * i + c]; o.c1[c] = d.cmap_c1[3 * i + c]; o.c2[c] = d.cmap_c2[3 * i + c]; }
        o.size = d.cmap_size[i];
        const int occl = d.mat_kind[i] != RAYHS_MAT_EMMIT;  // isOccluder (src/RayHs.hs:81-82)
        const int g = d.obj_geom_index[i];
        if (d.obj_geom_kind[i] == RAYHS_GEOM_PLANE) {
            if (plane_used[(size_t)g]++) return set_error(RAYHS_ERR_INVALID, "plane %d referenced by two objects", g);
            DPlane &p = planes[(size_t)g];
            for (int c = 0; c < 3; c++) { p.p[c] = d.plane_point[3 * g + c]; p.n[c] = d.plane_normal[3 * g + c]; p.t[c] = d.plane_tangent[3 * g + c]; }
            // b = cross t n (src/Geometry.hs:76)
            p.b[0] = p.t[1] * p.n[2] - p.t[2] * p.n[1];
            p.b[1] = p.t[2] * p.n[0] - p.t[0] * p.n[2];
            p.b[2] = p.t[0] * p.n[1] - p.t[1] * p.n[0];
            p.obj = i; p.occluder = occl; p.shadow_skip = occl ? 0ull : ~0ull;
        } else if (d.obj_geom_kind[i] == RAYHS_GEOM_SPHERE) {
            if (sphere_used[(size_t)g]++) return set_error(RAYHS_ERR_INVALID, "sphere %d referenced by two objects", g);
            DSphere &s = spheres[(size_t)g];
            for (int c = 0; c < 3; c++) s.c[c] = d.sphere_center[3 * g + c];
            s.r2 = d.sphere_radius[g] * d.sphere_radius[g];  // r ^ 2 (src/Geometry.hs:89)
            s.obj = i; s.occluder = occl; s.shadow_skip = occl ? 0ull : ~0ull;
        } else {
            if (mesh_used[(size_t)g]++) return set_error(RAYHS_ERR_INVALID, "mesh %d referenced by two objects", g);
            DMesh &m = meshes[(size_t)g];
            std::memset(&m, 0, sizeof m);
            m.root = d.mesh_root[g]; m.obj = i; m.occluder = occl;
            m.froot = mesh_froot[(size_t)g]; m.tri_base = mesh_tri_base[(size_t)g];
            for (int c = 0; c < 3; c++) {
                // the reference root's own box (src/KDTree.hs:84); an Empty tree is never entered
                m.lo[c] = m.root >= 0 ? d.node_lo[3 * m.root + c] : 0;
                m.hi[c] = m.root >= 0 ? d.node_hi[3 * m.root + c] : 0;
                m.center[c] = mesh_center[3 * (size_t)g + (size_t)c];
                m.flo[c] = mesh_fbox[6 * (size_t)g + (size_t)c];
                m.fhi[c] = mesh_fbox[6 * (size_t)g + 3 + (size_t)c];
            }
        }
    }
    // geometry not referenced by any object is dropped from the query loops
    {
        std::vector<DPlane> p2; std::vector<DSphere> s2; std::vector<DMesh> m2;
        for (int g = 0; g < d.n_planes; g++) if (plane_used[(size_t)g]) p2.push_back(planes[(size_t)g]);
        for (int g = 0; g < d.n_spheres; g++) if (sphere_used[(size_t)g]) s2.push_back(spheres[(size_t)g]);
        for (int g = 0; g < d.n_meshes; g++) if (mesh_used[(size_t)g]) m2.push_back(meshes[(size_t)g]);
        planes.swap(p2); spheres.swap(s2); meshes.swap(m2);
    }
    // Planes grouped by the single non-zero component of their normal (x, y, z), general position last: for a finite ray
    // the two dot products of the plane test then consist of ONE product plus two exact zeros (trace_kernel.hip,
    // plane loop).  Closest-hit ties are decided by the object index, never by the storage order.
    int (&plane_axis_end)[3] = pk.plane_axis_end;
    plane_axis_end[0] = plane_axis_end[1] = plane_axis_end[2] = 0;
    {
        auto axis_of = [](const DPlane &p) {
            for (int c = 0; c < 3; c++)
                if (p.n[c] != 0 && std::isfinite(p.n[c]) && p.n[(c + 1) % 3] == 0 && p.n[(c + 2) % 3] == 0 && std::isfinite(p.p[0]) && std::isfinite(p.p[1]) &&
                    std::isfinite(p.p[2]))
                    return c;
            return 3;
        };
        std::vector<DPlane> grouped;
        for (int a = 0; a < 4; a++) {
            for (DPlane p : planes)
                if (axis_of(p) == a) {
                    p.nk = a < 3 ? p.n[a] : 0.0; p.pk = a < 3 ? p.p[a] : 0.0;
                    grouped.push_back(p);
                }
            if (a < 3) plane_axis_end[a] = (int)grouped.size();
        }
        planes.swap(grouped);
    }
    std::vector<DLight> &lights = pk.lights; lights.assign((size_t)d.n_lights, DLight());
    for (int i = 0; i < d.n_lights; i++) {
        DLight &l = lights[(size_t)i];
        l.kind = d.light_kind[i];
        for (int c = 0; c < 3; c++) { l.v[c] = d.light_vec[3 * i + c]; l.c[c] = d.light_color[3 * i + c]; }
        l.radius = d.light_radius[i];
        // a point light with an ordinary radius: the kernel may compute its falloff without the divisions' range fix-ups (trace_kernel.hip, rcp_in_range)
        l.short_ok = (l.kind == RAYHS_LIGHT_POINT && l.radius >= 1e-60 && l.radius <= 1e60) ? 1 : 0;
    }
    pk.texels.clear();
    if (d.n_texels > 0) pk.texels.assign(d.texels, d.texels + 3 * (size_t)d.n_texels);

    pk.n_objects = d.n_objects; pk.n_lights = d.n_lights; pk.n_nodes = d.n_nodes; pk.n_tris = d.n_tris;
    pk.tree_depth = tree_depth; pk.fast_depth = fast_depth;
    for (size_t i = 0; i < 9 * (size_t)d.n_tris && !pk.huge_coords; i++)
        pk.huge_coords = std::isfinite(d.tri_pos[i]) && std::fabs(d.tri_pos[i]) > 1e90;
    return RAYHS_OK;
}

// one device's copy of a packed scene (HBM-resident until destroy) plus its launch state
static int upload_scene(const PackedScene &pk, int device, rayhs_scene **out) {
    *out = nullptr;
    int ndev = rayhs_device_count();
    if (ndev <= 0) return set_error(RAYHS_ERR_NO_DEVICE, "no HIP device available (librayhs_hip has no CPU fallback)");
    if (device < 0 || device >= ndev) return set_error(RAYHS_ERR_INVALID, "device %d out of range (have %d)", device, ndev);
    using clk = std::chrono::steady_clock;
    const clk::time_point tu0 = clk::now();
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    const clk::time_point tu1 = clk::now();
    g_ms_malloc = g_ms_copy = 0;
    int rc;
    rayhs_scene *s = new rayhs_scene();
    s->device = device;
    s->cu_count = prop.multiProcessorCount;
    s->tree_depth = pk.tree_depth;
    s->fast_depth = pk.fast_depth;
    s->huge_coords = pk.huge_coords;
    auto fail = [&](int code) { rayhs_scene_destroy(s); return code; };
    if ((rc = upload(s, pk.nodes, &s->d.nodes))) return fail(rc);
    if ((rc = upload(s, pk.tris, &s->d.tris))) return fail(rc);
    if ((rc = upload(s, pk.attrs, &s->d.attrs))) return fail(rc);
    if ((rc = upload(s, pk.fnodes, &s->d.fnodes))) return fail(rc);
    if ((rc = upload(s, pk.ftris, &s->d.ftris))) return fail(rc);
    if ((rc = upload(s, pk.fattrs, &s->d.fattrs))) return fail(rc);
    if ((rc = upload(s, pk.planes, &s->d.planes))) return fail(rc);
    if ((rc = upload(s, pk.spheres, &s->d.spheres))) return fail(rc);
    if ((rc = upload(s, pk.meshes, &s->d.meshes))) return fail(rc);
    if ((rc = upload(s, pk.objects, &s->d.objects))) return fail(rc);
    if ((rc = upload(s, pk.lights, &s->d.lights))) return fail(rc);
    if ((rc = upload(s, pk.texels, &s->d.texels))) return fail(rc);
    s->d.n_planes = (int)pk.planes.size();
    for (int a = 0; a < 3; a++) s->d.plane_axis_end[a] = (int32_t)((uint32_t)pk.plane_axis_end[a] * (uint32_t)sizeof(DPlane));   // (byte offsets: what the query loop counts in)
    s->d.n_spheres = (int)pk.spheres.size();
    s->d.sphere_occluders = 0;
    for (const DSphere &sp : pk.spheres) s->d.sphere_occluders += sp.occluder ? 1 : 0;
    s->d.n_meshes = (int)pk.meshes.size();
    s->d.n_objects = pk.n_objects;
    s->d.n_lights = pk.n_lights;
    s->d.n_nodes = pk.n_nodes;
    s->d.n_tris = pk.n_tris;
    s->d.n_fnodes = (int)pk.fnodes.size();
    void *p = nullptr;
    if (hipMalloc(&p, sizeof(unsigned long long) * kTicketSlots * kTicketRings) != hipSuccess) return fail(set_error(RAYHS_ERR_HIP, "hipMalloc(tickets) failed"));
    s->tickets = (unsigned long long *)p;
    if (hipMalloc(&p, sizeof(unsigned long long) * CNT_SLOTS_DEV) != hipSuccess) return fail(set_error(RAYHS_ERR_HIP, "hipMalloc(counters) failed"));
    s->counters = (unsigned long long *)p;
    *out = s;
    if (g_cfg.debug_timing)
        std::fprintf(stderr, "upload_scene: wait for the runtime + device properties %.1f ms, 12 x hipMalloc %.1f ms, 12 x hipMemcpy %.1f ms, all %.1f ms\n",
                     std::chrono::duration<double, std::milli>(tu1 - tu0).count(), g_ms_malloc, g_ms_copy, std::chrono::duration<double, std::milli>(clk::now() - tu0).count());
    return RAYHS_OK;
}

extern "C" int rayhs_scene_create(const rayhs_scene_desc *desc, int device, rayhs_scene **out) {
    if (!out) return set_error(RAYHS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    // (validation and packing come first: a bad description is reported as such even where no device exists)
    using clk = std::chrono::steady_clock;
    const clk::time_point t0 = clk::now();
    PackedScene pk;
    int rc = pack_scene(desc, pk);
    if (rc) return rc;
    const clk::time_point t1 = clk::now();
    rc = upload_scene(pk, device, out);
    if (g_cfg.debug_timing)  // diagnostics: (the upload waits for rayhs_device_prepare when that runs on another thread)
        std::fprintf(stderr, "rayhs_scene_create: validate + flatten + BVH %.1f ms, device allocation + upload %.1f ms\n",
                     std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(clk::now() - t1).count());
    return rc;
}

extern "C" void rayhs_scene_destroy(rayhs_scene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (void *p : s->allocs) (void)hipFree(p);
    if (s->frames) (void)hipFree(s->frames);
    if (s->order_dev) (void)hipFree(s->order_dev);
    if (s->cost_dev) (void)hipFree(s->cost_dev);
    if (s->order_aux) (void)hipFree(s->order_aux);
    if (s->sbuf) (void)hipFree(s->sbuf);
    if (s->tickets) (void)hipFree(s->tickets);
    if (s->counters) (void)hipFree(s->counters);
    if (s->host_out_dev) (void)hipFree(s->host_out_dev);
    if (s->captured_dev) (void)hipFree(s->captured_dev);
    if (s->last_done) (void)hipEventDestroy(s->last_done);
    if (s->host_e0) (void)hipEventDestroy(s->host_e0);
    if (s->host_e1) (void)hipEventDestroy(s->host_e1);
    delete s;
}

// -------------------------------------------------------------------- render
extern "C" int rayhs_render_rows(const rayhs_render_params *p) {
    if (!p || p->row0 < 0 || p->row1 < p->row0 || p->stripe_count < 1 || p->stripe_rows < 1 || p->stripe_rank < 0 || p->stripe_rank >= p->stripe_count)
        return set_error(RAYHS_ERR_INVALID, "bad row/stripe selection");
    int total = p->row1 - p->row0, n = 0;
    int stripes = (total + p->stripe_rows - 1) / p->stripe_rows;
    for (int s = p->stripe_rank; s < stripes; s += p->stripe_count) n += std::min(p->stripe_rows, total - s * p->stripe_rows);
    return n;
}

// rayFromPixel's per-call constants (src/Projection.hs:22-46, src/Mat.hs:89-93)
static void make_camera(const rayhs_camera &c, int width, int height, DCamera &o) {
    std::memset(&o, 0, sizeof o);
    auto norm = [](const double v[3], double out[3]) {
        double l = 1 / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        out[0] = l * v[0]; out[1] = l * v[1]; out[2] = l * v[2];
    };
    auto crs = [](const double a[3], const double b[3], double out[3]) {
        out[0] = a[1] * b[2] - a[2] * b[1]; out[1] = a[2] * b[0] - a[0] * b[2]; out[2] = a[0] * b[1] - a[1] * b[0];
    };
    double diff[3] = {c.target[0] - c.position[0], c.target[1] - c.position[1], c.target[2] - c.position[2]};
    double fwd[3], right[3], up[3], tmp[3];
    norm(diff, fwd);
    crs(c.up, fwd, tmp);
    norm(tmp, right);
    crs(fwd, right, up);
    // fromColumns right up forward
    o.m[0] = right[0]; o.m[1] = up[0]; o.m[2] = fwd[0];
    o.m[3] = right[1]; o.m[4] = up[1]; o.m[5] = fwd[1];
    o.m[6] = right[2]; o.m[7] = up[2]; o.m[8] = fwd[2];
    for (int k = 0; k < 3; k++) o.pos[k] = c.position[k];
    o.w = (double)width; o.h = (double)height;
    o.half_w = o.w / 2; o.half_h = o.h / 2;
    double aspect = o.w / o.h;  // aspectSize (src/Projection.hs:41-46)
    if (aspect > 1) { o.apw = o.w; o.aph = o.w / aspect; } else { o.apw = aspect * o.h; o.aph = o.h; }
    o.f = 0.5 * o.h / (std::tan(0.5) * c.fovy);  // sic: (tan 0.5) * fovy (src/Projection.hs:35)
    o.proj_kind = c.proj_kind;
    // short_div.h in ray_from_pixel: width, height and the extents are positive and at most 2^24; the numerators are 0 or at least 2^-60 in magnitude;
    // f (a perspective camera's) must be finite and of moderate size, or the normalisation keeps the compiler's full sequences
    const double af = std::fabs(o.f);
    o.short_ok = (c.proj_kind == RAYHS_PROJ_ORTHOGRAPHIC || (std::isfinite(o.f) && af >= 0x1p-300 && af <= 0x1p300)) ? 1 : 0;
}

// An explicit list of the stripes one launch renders (rayhs_multi: cost-balanced ownership) instead of the round-robin
// rule of rayhs_render_params.  d_map: device array of global stripe indices, ascending; rows: image rows they cover;
// id: tells one list from another in the tile-order cache.
struct StripeList { const int32_t *d_map = nullptr; int n_stripes = 0, rows = 0, id = 0; };

static uint64_t mix64(uint64_t x);
static int prepare_launch(const rayhs_scene *s, const rayhs_camera *cam, const rayhs_render_params *p, const StripeList *sl, DCamera &dc, DParams &dp,
                          int &variant, bool &exact, int &grid, size_t &lds) {
    if (!s || !cam || !p) return set_error(RAYHS_ERR_INVALID, "NULL argument");
    if (p->width <= 0 || p->height <= 0 || p->max_depth < 0 || p->spp < 1) return set_error(RAYHS_ERR_INVALID, "bad width/height/maxDepth/spp");
    if (p->row1 > p->height) return set_error(RAYHS_ERR_INVALID, "row1 beyond the image");
    if (p->out_format < RAYHS_OUT_F32 || p->out_format > RAYHS_OUT_I16) return set_error(RAYHS_ERR_INVALID, "bad out_format");
    if (p->flags & ~RAYHS_FLAG_ALL) return set_error(RAYHS_ERR_INVALID, "unknown flag bits 0x%x (this library knows 0x%x)", p->flags & ~RAYHS_FLAG_ALL, RAYHS_FLAG_ALL);
    if (cam->proj_kind != RAYHS_PROJ_ORTHOGRAPHIC && cam->proj_kind != RAYHS_PROJ_PERSPECTIVE) return set_error(RAYHS_ERR_INVALID, "bad projection kind");
    int rows = sl ? sl->rows : rayhs_render_rows(p);
    if (rows < 0) return rows;
    if ((p->flags & RAYHS_FLAG_EXHAUSTIVE) && !(p->flags & RAYHS_FLAG_COUNT)) return set_error(RAYHS_ERR_INVALID, "RAYHS_FLAG_EXHAUSTIVE needs RAYHS_FLAG_COUNT");
    make_camera(*cam, p->width, p->height, dc);
    std::memset(&dp, 0, sizeof dp);
    dp.width = p->width; dp.height = p->height; dp.max_depth = p->max_depth; dp.spp = p->spp;
    dp.seed = mix64(p->seed + 0x9e3779b97f4a7c15ULL);  // the sampler's seed term (include/rayhs.h, rayhs_sample01), mixed once here instead of per sample
    dp.row0 = p->row0; dp.n_rows = rows;
    dp.stripe_rows = p->stripe_rows; dp.stripe_rank = p->stripe_rank; dp.stripe_count = p->stripe_count;
    if (sl) { dp.stripe_map = sl->d_map; dp.stripe_rank = sl->id; dp.stripe_count = 0; }  // rank / count are unused with a list (the rank slot keys the tile-order cache)
    dp.out_format = p->out_format;
    dp.flags_dev = (p->flags & RAYHS_FLAG_POW5_SQMUL) ? 2 : 0;
    // samples of a pixel are folded inside the wave when they all sit in one 64-item chunk (trace_kernel.hip)
    dp.wave_resolve = (p->spp > 1 && 64 % p->spp == 0 && !g_cfg.no_wave_resolve) ? 1 : 0;
    auto log2_exact = [](int v) { int s = 0; while ((1 << s) < v) s++; return (1 << s) == v ? s : -1; };
    dp.spp_shift = log2_exact(p->spp);
    dp.inv_spp = 1.0 / (double)p->spp;
    dp.stripe_shift = log2_exact(p->stripe_rows);
    dp.tiles_x = (p->width + 7) / 8;
    dp.inv_tiles_x = 1.0f / (float)dp.tiles_x;
    // the refill finds a tile's row and column with a float reciprocal plus one correction step (trace_kernel.hip)
    const int64_t n_tiles64 = (int64_t)dp.tiles_x * (int64_t)((rows + 7) / 8);  // checked before narrowing: a very wide frame overflows int
    if (n_tiles64 >= ((int64_t)1 << 24) || rows >= (1 << 23))
        return set_error(RAYHS_ERR_UNSUPPORTED, "frame too large: %lld tiles of 8x8 pixels (limit 2^24)", (long long)n_tiles64);
    dp.n_tiles = (int)n_tiles64;
    exact = (p->flags & RAYHS_FLAG_FMA) == 0;
    // kernel variants: 0 fast (16-bit stack entries, < 65536 BVH nodes), 3 fast (32-bit entries),
    // 1 counting, 2 exhaustive + counting (the reference's tree)
    variant = (p->flags & RAYHS_FLAG_EXHAUSTIVE) ? 2 : (p->flags & RAYHS_FLAG_COUNT) ? 1 : (s->d.n_fnodes < 65536 && s->d.n_nodes < 65536 ? 0 : 3);
    // A scene with finite vertex coordinates beyond 1e90 is rendered by the exhaustive kernel (the reference's own tree and fp64 boxes, the
    // compiler's full IEEE divisions), whatever was asked for: the fast walks' triangle tests take 1 / det without the division's range
    // fix-ups, which is the same double only while |det| < 2^1022 -- certain for coordinates below 1e90 and rays below 1e30 (begin()), not
    // beyond.  Same pixels (the exhaustive and the fast kernels agree bit for bit), slower, and nobody renders such a scene.
    if (s->huge_coords) variant = 2;
    dp.stack_cap = std::max(std::max(s->tree_depth, s->fast_depth) + 2, 4);  // either tree may be walked (trace_kernel.hip)
#ifdef RAYHS_DEV_KNOBS
    if (g_cfg.stack_cap > 0) dp.stack_cap = std::max(4, g_cfg.stack_cap);  // development knob (UNSAFE below the tree depth: exists only in -DRAYHS_DEV_KNOBS builds)
#endif
    const int entry = exact ? stack_entry_bytes_exact(variant) : stack_entry_bytes_fast(variant);
    // LDS per workgroup: context slots + stack.  160 KiB per CU / 4 workgroups = 40 KiB each; if the stack
    // leaves room, three more context slots keep the light term in LDS instead of the global workspace.
    const size_t stack_bytes = (size_t)dp.stack_cap * TRACE_BLOCK * (size_t)entry;
    dp.ctx_slots = CTX_SLOTS;
    bool term_in_lds = (size_t)(CTX_SLOTS + 3) * TRACE_BLOCK * sizeof(double) + stack_bytes <= 40 * 1024;
#ifdef RAYHS_DEV_KNOBS
    if (g_cfg.term_in_workspace) term_in_lds = false;
#endif
    if (term_in_lds) dp.ctx_slots = CTX_SLOTS + 3;
    lds = (size_t)dp.ctx_slots * TRACE_BLOCK * sizeof(double) + stack_bytes;
#ifdef RAYHS_PROFILE_LOOPS
    lds += 16;  // one word per wave of a workgroup: the per-lane-walk census of the diagnostic build (trace_kernel.hip, traverse())
#endif
    if (lds > 160 * 1024) return set_error(RAYHS_ERR_UNSUPPORTED, "tree depth %d needs %zu B of LDS per workgroup (limit 160 KiB)", dp.stack_cap - 2, lds);
    // persistent grid: as many workgroups as the chip keeps resident
    int *cache = s->grid_cache[exact ? 1 : 0][variant];
    if (cache[0] != (int)lds || cache[1] <= 0) {
        int per_cu = exact ? occupancy_trace_exact(variant, lds) : occupancy_trace_fast(variant, lds);
        if (per_cu <= 0) return set_error(RAYHS_ERR_HIP, "trace kernel cannot be resident (occupancy query returned %d)", per_cu);
#ifdef RAYHS_DEV_KNOBS
        if (g_cfg.blocks_per_cu > 0) per_cu = std::max(1, std::min(per_cu, g_cfg.blocks_per_cu));  // development knob
#endif
        cache[0] = (int)lds; cache[1] = per_cu;
        if (g_cfg.debug_counters)
            std::fprintf(stderr, "rayhs launch: variant %d, lds %zu B, %d blocks/CU, stack cap %d (ref depth %d, bvh depth %d)\n", variant, lds, per_cu,
                         dp.stack_cap, s->tree_depth, s->fast_depth);
    }
    int wanted = (dp.n_tiles + 3) / 4;  // 4 waves per workgroup, one tile per wave at a time
    grid = std::max(1, std::min(wanted, s->cu_count * cache[1]));
    // the continuation workspace (6 + 14 * maxDepth doubles per resident lane) is addressed with 32-bit byte offsets:
    // a very deep recursion limit runs on a smaller persistent grid rather than failing
    const size_t lane_bytes = ((size_t)LANE_SLOTS + (size_t)std::max(dp.max_depth, 1) * FRAME_SLOTS) * sizeof(double);
    const size_t max_blocks = (((size_t)1 << 32) - 1) / (lane_bytes * TRACE_BLOCK);
    if (max_blocks < 1) return set_error(RAYHS_ERR_UNSUPPORTED, "maxDepth %d needs %zu bytes of continuation frames per lane; the limit is 16 MiB", dp.max_depth, lane_bytes);
    grid = (int)std::min<size_t>((size_t)grid, max_blocks);
    dp.frame_stride = grid * TRACE_BLOCK;
    return RAYHS_OK;
}

static int ensure_workspace(const rayhs_scene *s, const DParams &dp, hipStream_t st) {
    size_t need = ((size_t)LANE_SLOTS + (size_t)std::max(dp.max_depth, 1) * FRAME_SLOTS) * (size_t)dp.frame_stride * sizeof(double);
#ifdef RAYHS_PROBE_WS_TWICE
    need = 2 * ((size_t)LANE_SLOTS + (size_t)std::max(dp.max_depth, 3) * FRAME_SLOTS) * (size_t)dp.frame_stride * sizeof(double);  // (maxDepth <= 3 only)
#endif
    if (need >= ((size_t)1 << 32) || dp.frame_stride >= (1 << 24))  // 32-bit byte offsets / 24-bit multiplies in the kernel
        return set_error(RAYHS_ERR_INVALID, "max_depth %d needs a %zu-byte continuation workspace; the limit is 4 GiB", dp.max_depth, need);
    if (need > s->frames_bytes) {
        if (s->frames) HIP_TRY(hipFree(s->frames));
        s->frames = nullptr; s->frames_bytes = 0;
        void *p = nullptr;
        HIP_TRY(hipMalloc(&p, need));
        s->frames = (double *)p; s->frames_bytes = need;
        // (Not touched here: a memset of the fresh ~100 MB in front of the first frame took the first-touch cost out of
        // that frame's kernel time -- 4-5 % of a 1080p frame, tools/first_frame.py -- but cost 3.4 ms of its own, serially,
        // where the kernel absorbs the touches it really makes in parallel: `rayhs --timing` render 7.9 -> 11.3 ms.)
        (void)st;
    }
    // spp > 1: one radiance triple per (pixel, sample), folded in sample order by rayhs_resolve
    size_t sneed = dp.spp > 1 && !dp.wave_resolve ? (size_t)dp.n_rows * (size_t)dp.width * (size_t)dp.spp * 3 * sizeof(double) : 0;
    if (sneed > s->sbuf_bytes) {
        if (s->sbuf) HIP_TRY(hipFree(s->sbuf));
        s->sbuf = nullptr; s->sbuf_bytes = 0;
        void *p = nullptr;
        HIP_TRY(hipMalloc(&p, sneed));
        s->sbuf = (double *)p; s->sbuf_bytes = sneed;
    }
    return RAYHS_OK;
}

// Feedback scheduling of the production kernels (why: rayhs_order_tiles in trace_kernel.hip).  Every frame records how
// long each of its chunks took (DParams.tile_cost) and, right behind it on the same stream, a one-workgroup kernel
// turns those times into the tile order of the NEXT frame of the same geometry (size, rows, stripes, spp, depth) --
// the same view in a progressive render or a benchmark, a neighbouring one in an animation.  The first frame of a
// geometry -- the only one a one-shot `rayhs -oX scene.json` renders -- is ordered by a static estimate instead
// (rayhs_static_costs: mesh-box tiles first, reflective surfaces next).  Nothing is read back to the host.
static int plan_tile_order(const rayhs_scene *s, const rayhs_render_params *p, int variant, DParams &dp, hipStream_t st, bool &record, bool &need_static, bool &reorder) {
    // (dp.stripe_rank / dp.stripe_count carry the stripe list's id / 0 when the launch has an explicit list)
    dp.tile_order = nullptr; dp.tile_cost = nullptr; record = false; need_static = false; reorder = false;
#ifdef RAYHS_PROFILE_TIME
    const bool production = variant != 2;  // the diagnostic build's "counting" kernel is the production kernel with time stamps
#else
    const bool production = variant == 0 || variant == 3;
#endif
    if (!production || dp.n_tiles < 512 || g_cfg.scan_order) return RAYHS_OK;
    const int key[10] = {p->width, p->height, p->row0, p->row1, p->stripe_rows, dp.stripe_rank, dp.stripe_count, p->spp, p->max_depth, dp.n_tiles};
    if (std::memcmp(key, s->order_key, sizeof key) != 0) {
        s->order_valid = false;
        s->frames_on_key = 0;
        if ((size_t)dp.n_tiles > s->order_cap) {
            if (s->order_dev) HIP_TRY(hipFree(s->order_dev));
            if (s->cost_dev) HIP_TRY(hipFree(s->cost_dev));
            s->order_dev = nullptr; s->cost_dev = nullptr; s->order_cap = 0;
            void *q = nullptr;
            HIP_TRY(hipMalloc(&q, (size_t)dp.n_tiles * sizeof(int32_t)));
            s->order_dev = (int32_t *)q;
            HIP_TRY(hipMalloc(&q, (size_t)dp.n_tiles * sizeof(float)));
            s->cost_dev = (float *)q;
            s->order_cap = (size_t)dp.n_tiles;
        }
        if (!s->order_aux) {
            void *q = nullptr;
            HIP_TRY(hipMalloc(&q, (size_t)ORDER_MAX_BLOCKS * 1024 * sizeof(unsigned)));  // (written whole by every ordering: never cleared)
            s->order_aux = (unsigned *)q;
        }
        HIP_TRY(hipMemsetAsync(s->cost_dev, 0, (size_t)dp.n_tiles * sizeof(float), st));  // once per geometry, in stream order; afterwards the ordering kernel clears it
        std::memcpy(s->order_key, key, sizeof key);
    }
    // no measured order yet: the launcher runs the static estimate + ordering in front of the trace kernel
    need_static = !s->order_valid && !g_cfg.no_static_order;
    if (s->order_valid || need_static) dp.tile_order = s->order_dev;
#ifdef RAYHS_DEV_KNOBS
    if (g_cfg.freeze_order_after >= 0)  // development: keep the order the n-th launch of the handle left, stop re-ordering
        if (s->order_valid && (long long)s->next_ticket > g_cfg.freeze_order_after) { dp.tile_cost = s->cost_dev; return RAYHS_OK; }  // (times still taken: the kernel does the same work)
#endif
    dp.tile_cost = s->cost_dev;
    record = true;
    // The order of a view that does not change settles within a few frames: the first four frames of a geometry re-order
    // every time, later ones every fourth frame (the chunk times of the frames in between accumulate: the cost of a tile is
    // the longest chunk seen since the last ordering).  Three frames in four then run without the two ordering kernels
    // behind them: 15 us + two launch gaps, 10 % of a 0.17 ms Cornell step.  RAYHS_REORDER_EVERY=1: every frame.
    reorder = s->frames_on_key < 4 || s->frames_on_key % g_cfg.reorder_every == 0;
    s->frames_on_key++;
    return RAYHS_OK;
}

static int launch(const rayhs_scene *s, const rayhs_camera *cam, const rayhs_render_params *p, void *d_out, hipStream_t st,
                  uint64_t *d_counters, hipEvent_t e0, hipEvent_t e1, const StripeList *sl = nullptr) {
    DCamera dc; DParams dp; int variant, grid; bool exact; size_t lds;
    int rc = prepare_launch(s, cam, p, sl, dc, dp, variant, exact, grid, lds);
    if (rc) return rc;
    if (!d_out) return set_error(RAYHS_ERR_INVALID, "d_out is NULL");
    if (variant == 2 && !d_counters && s->huge_coords) d_counters = (uint64_t *)s->counters;  // (the exhaustive kernel always counts: into the handle's own buffer then)
    if ((variant == 1 || variant == 2) && !d_counters) return set_error(RAYHS_ERR_INVALID, "RAYHS_FLAG_COUNT needs d_counters");
    HIP_TRY(hipSetDevice(s->device));
    // rayhs_scene_tile_order reports the order of the LAST launch: a launch that hands out no tiles (an empty row selection), or fails
    // before it has an order, must not leave the previous frame's order standing as its own
    if (s->capture_order) { s->captured_tiles = 0; s->captured_static = 0; }
    if (dp.n_tiles == 0) {  // nothing to trace: the caller's events still get recorded, so that it can wait on / time them
        if (e0) HIP_TRY(hipEventRecord(e0, st));
        if (e1) HIP_TRY(hipEventRecord(e1, st));
        return RAYHS_OK;
    }
    // launches of one handle share its workspace and tile-order buffers: a launch on another stream than the previous
    // one first waits for that one to finish (device-side wait, no host synchronisation)
    if (!s->last_done) HIP_TRY(hipEventCreateWithFlags(&s->last_done, hipEventDisableTiming));
    if (s->have_last && s->last_stream != st) HIP_TRY(hipStreamWaitEvent(st, s->last_done, 0));
    if ((rc = ensure_workspace(s, dp, st))) return rc;
    bool record_costs = false, static_order = false, reorder = false;
    if ((rc = plan_tile_order(s, p, variant, dp, st, record_costs, static_order, reorder))) return rc;
    // ticket counters of this launch: zeroed here, unless the ordering kernel behind the previous launch of this handle has
    // already done it (one fill kernel and one launch gap less per frame in a steady-state loop)
    const unsigned launch_no = s->next_ticket++;
    unsigned long long *ticket = s->tickets + (size_t)(launch_no % kTicketRings) * kTicketSlots;
    if ((long long)launch_no < s->ticket_clean_lo || (long long)launch_no > s->ticket_clean_hi)
        HIP_TRY(hipMemsetAsync(ticket, 0, sizeof(unsigned long long) * kTicketSlots, st));
    dp.sbuf = dp.spp > 1 && !dp.wave_resolve ? s->sbuf : nullptr;
    if (static_order) {  // first frame of this geometry: static tile classes -> work order (two tiny kernels, ~20 us)
        hipError_t es = exact ? launch_static_costs_exact(st, s->d, dc, dp, s->cost_dev) : launch_static_costs_fast(st, s->d, dc, dp, s->cost_dev);
        if (es == hipSuccess) es = exact ? launch_order_tiles_exact(st, s->cost_dev, dp.n_tiles, s->order_dev, s->order_aux, nullptr, 0, 0, 1, 0) : launch_order_tiles_fast(st, s->cost_dev, dp.n_tiles, s->order_dev, s->order_aux, nullptr, 0, 0, 1, 0);
        if (es != hipSuccess) return set_error(RAYHS_ERR_HIP, "static tile ordering launch failed: %s", hipGetErrorString(es));
    }
    if (s->capture_order) {  // rayhs_scene_capture_tile_order: keep the work order THIS frame is given (scan order: nothing to keep)
        s->captured_tiles = 0; s->captured_static = static_order ? 1 : 0;
        if (dp.tile_order != nullptr) {
            if ((size_t)dp.n_tiles > s->captured_cap) {
                if (s->captured_dev) HIP_TRY(hipFree(s->captured_dev));
                s->captured_dev = nullptr; s->captured_cap = 0;
                void *q = nullptr;
                HIP_TRY(hipMalloc(&q, (size_t)dp.n_tiles * sizeof(int32_t)));
                s->captured_dev = (int32_t *)q; s->captured_cap = (size_t)dp.n_tiles;
            }
            HIP_TRY(hipMemcpyAsync(s->captured_dev, s->order_dev, (size_t)dp.n_tiles * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
            s->captured_tiles = dp.n_tiles;
        }
    }
    if (e0) HIP_TRY(hipEventRecord(e0, st));
    hipError_t e = exact ? launch_trace_exact(variant, grid, lds, st, s->d, dc, dp, d_out, (unsigned long long *)d_counters, ticket, s->frames)
                         : launch_trace_fast(variant, grid, lds, st, s->d, dc, dp, d_out, (unsigned long long *)d_counters, ticket, s->frames);
    if (e != hipSuccess) return set_error(RAYHS_ERR_HIP, "trace kernel launch failed: %s", hipGetErrorString(e));
    if (e1) HIP_TRY(hipEventRecord(e1, st));
#ifdef RAYHS_DEV_KNOBS
    if (record_costs && !g_cfg.dump_tile_costs.empty()) {  // development: measured chunk times per tile, then the static estimate
        std::vector<float> h((size_t)dp.n_tiles);
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(h.data(), s->cost_dev, h.size() * sizeof(float), hipMemcpyDeviceToHost));
        const std::string dump_path = g_cfg.dump_tile_costs;
        if (FILE *f = std::fopen(dump_path.c_str(), "ab")) {
            int hdr[4] = {dp.n_tiles, dp.tiles_x, dp.spp, 0};
            std::fwrite(hdr, sizeof hdr, 1, f);
            std::fwrite(h.data(), sizeof(float), h.size(), f);
            void *tmp = nullptr;
            HIP_TRY(hipMalloc(&tmp, h.size() * sizeof(float)));
            DParams dpx = dp;
            dpx.flags_dev = g_cfg.dump_raw_feature ? 1 : 0;
            hipError_t es = exact ? launch_static_costs_exact(st, s->d, dc, dpx, (float *)tmp) : launch_static_costs_fast(st, s->d, dc, dpx, (float *)tmp);
            (void)es;
            HIP_TRY(hipStreamSynchronize(st));
            HIP_TRY(hipMemcpy(h.data(), tmp, h.size() * sizeof(float), hipMemcpyDeviceToHost));
            (void)hipFree(tmp);
            std::fwrite(h.data(), sizeof(float), h.size(), f);
            std::fclose(f);
        }
    }
#endif
    if (reorder) {  // the chunk times measured since the last ordering -> the tile order of the frames that follow; the kernel also
                    // zeroes the ticket counters of the next kTicketsAhead launches of this handle
        e = exact ? launch_order_tiles_exact(st, s->cost_dev, dp.n_tiles, s->order_dev, s->order_aux, s->tickets, (int)((launch_no + 1) % kTicketRings), kTicketsAhead, kTicketRings, kTicketSlots)
                  : launch_order_tiles_fast(st, s->cost_dev, dp.n_tiles, s->order_dev, s->order_aux, s->tickets, (int)((launch_no + 1) % kTicketRings), kTicketsAhead, kTicketRings, kTicketSlots);
        if (e != hipSuccess) return set_error(RAYHS_ERR_HIP, "tile ordering launch failed: %s", hipGetErrorString(e));
        s->order_valid = true;
    }
    if (dp.spp > 1 && !dp.wave_resolve) {
        e = exact ? launch_resolve_exact(st, s->sbuf, d_out, dp.n_rows * dp.width, dp.spp, dp.out_format)
                  : launch_resolve_fast(st, s->sbuf, d_out, dp.n_rows * dp.width, dp.spp, dp.out_format);
        if (e != hipSuccess) return set_error(RAYHS_ERR_HIP, "resolve kernel launch failed: %s", hipGetErrorString(e));
    }
    // (ticket_clean_lo..hi promise that those launches' counters are zeroed by work ORDERED BEHIND last_done: they are set
    // only once that event has been recorded, together with the stream it was recorded on, so that an error return in
    // between leaves the next launch to zero its own counters)
    const long long clean_hi_before = s->ticket_clean_hi;
    s->ticket_clean_hi = -1;
    HIP_TRY(hipEventRecord(s->last_done, st));
    s->last_stream = st; s->have_last = true;
    if (reorder) { s->ticket_clean_lo = (long long)launch_no + 1; s->ticket_clean_hi = (long long)launch_no + kTicketsAhead; }
    else s->ticket_clean_hi = clean_hi_before;  // (what an earlier ordering kernel zeroed stays zeroed: this launch used only its own counters)
    return RAYHS_OK;
}

extern "C" int rayhs_render_device(const rayhs_scene *s, const rayhs_camera *cam, const rayhs_render_params *p, void *d_out,
                                   void *hip_stream, uint64_t *d_counters) {
    return launch(s, cam, p, d_out, (hipStream_t)hip_stream, d_counters, nullptr, nullptr);
}

extern "C" int rayhs_render_device_events(const rayhs_scene *s, const rayhs_camera *cam, const rayhs_render_params *p, void *d_out,
                                          void *hip_stream, uint64_t *d_counters, void *ev_before_trace, void *ev_after_trace) {
    return launch(s, cam, p, d_out, (hipStream_t)hip_stream, d_counters, (hipEvent_t)ev_before_trace, (hipEvent_t)ev_after_trace);
}


// Diagnostics / tests: the order in which a frame's 8x8 tiles were handed to the waves (scheduling only, never a pixel).
extern "C" int rayhs_scene_capture_tile_order(rayhs_scene *s, int enable) {
    if (!s) return set_error(RAYHS_ERR_INVALID, "NULL argument");
    s->capture_order = enable != 0;
    if (!enable) s->captured_tiles = 0;
    return RAYHS_OK;
}
extern "C" int rayhs_scene_tile_order(const rayhs_scene *s, int32_t *order, int cap, int *from_static) {
    if (!s) return set_error(RAYHS_ERR_INVALID, "NULL argument");
    if (from_static) *from_static = s->captured_static;
    const int n = s->captured_tiles;
    if (n == 0 || !order) return n;
    if (cap < n) return set_error(RAYHS_ERR_INVALID, "order buffer holds %d entries, the captured order has %d", cap, n);
    HIP_TRY(hipSetDevice(s->device));
    if (s->have_last) HIP_TRY(hipEventSynchronize(s->last_done));  // the capture copy was enqueued in front of that launch's kernels
    HIP_TRY(hipMemcpy(order, s->captured_dev, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) order[i] &= (1 << 28) - 1;  // (the entries carry an issue-priority class above bit 28)
    return n;
}

extern "C" int rayhs_render(const rayhs_scene *s, const rayhs_camera *cam, const rayhs_render_params *p, void *out_host, rayhs_stats *stats) {
    if (!s || !p || !out_host) return set_error(RAYHS_ERR_INVALID, "NULL argument");
    int rows = rayhs_render_rows(p);
    if (rows < 0) return rows;
    if (p->width <= 0) return set_error(RAYHS_ERR_INVALID, "bad width");
    HIP_TRY(hipSetDevice(s->device));
    using clk = std::chrono::steady_clock;
    const clk::time_point tt0 = clk::now();
    size_t esz = p->out_format == RAYHS_OUT_F64 ? 8 : p->out_format == RAYHS_OUT_F32 ? 4 : 2;
    size_t bytes = (size_t)rows * (size_t)p->width * 3 * esz;
    // the device frame and the two timing events live with the handle (grow-only): repeated calls allocate nothing
    if (bytes > s->host_out_bytes || !s->host_out_dev) {
        if (s->host_out_dev) HIP_TRY(hipFree(s->host_out_dev));
        s->host_out_dev = nullptr; s->host_out_bytes = 0;
        HIP_TRY(hipMalloc(&s->host_out_dev, std::max<size_t>(bytes, 16)));
        s->host_out_bytes = std::max<size_t>(bytes, 16);
    }
    if (!s->host_e0) HIP_TRY(hipEventCreate(&s->host_e0));
    if (!s->host_e1) HIP_TRY(hipEventCreate(&s->host_e1));
    const bool count = (p->flags & RAYHS_FLAG_COUNT) != 0;
    if (count) HIP_TRY(hipMemsetAsync(s->counters, 0, sizeof(unsigned long long) * CNT_SLOTS_DEV, nullptr));
#ifdef RAYHS_DEV_KNOBS
    if (g_cfg.debug_timing && g_cfg.debug_sync_first) {  // development: is anything still pending from the uploads?
        const clk::time_point q0 = clk::now();
        (void)hipDeviceSynchronize();
        std::fprintf(stderr, "rayhs_render: hipDeviceSynchronize before the launch took %.1f ms\n", std::chrono::duration<double, std::milli>(clk::now() - q0).count());
    }
#endif
    const clk::time_point tt1 = clk::now();
    int rc = launch(s, cam, p, s->host_out_dev, nullptr, count ? (uint64_t *)s->counters : nullptr, s->host_e0, s->host_e1);
    if (rc) return rc;
    const clk::time_point tt2 = clk::now();
    clk::time_point tt3 = tt2, tt4 = tt2;
    if (g_cfg.debug_timing) {  // diagnostics: when the trace kernel starts and ends, seen from the host
        (void)hipEventSynchronize(s->host_e0); tt3 = clk::now();
        (void)hipEventSynchronize(s->host_e1); tt4 = clk::now();
    }
    // the copy is ordered behind the kernels on the NULL stream and blocks the host until the frame is there
    if (bytes) HIP_TRY(hipMemcpy(out_host, s->host_out_dev, bytes, hipMemcpyDeviceToHost));
    if (g_cfg.debug_timing) {
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        std::fprintf(stderr, "rayhs_render: output buffer + events %.1f ms, enqueue (workspace, order buffers, launches) %.1f ms, until the trace kernel starts %.1f ms, "
                             "until it ends %.1f ms, copy of %zu bytes to the host %.1f ms\n",
                     ms(tt0, tt1), ms(tt1, tt2), ms(tt2, tt3), ms(tt3, tt4), bytes, ms(tt4, clk::now()));
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
#if defined(RAYHS_DEV_KNOBS) && (defined(RAYHS_PROFILE_TIME) || defined(RAYHS_WAVE_ENDS))
    const std::string wave_ends_path = g_cfg.dump_wave_ends;
    if (const char *path = wave_ends_path.empty() ? nullptr : wave_ends_path.c_str()) {  // development (tools/wave_ends.py): per wave (start, end, last chunk drawn at, chunks), 100 MHz ticks
        std::vector<unsigned long long> w((size_t)s->cu_count * 16 * 4);  // (4 workgroups of 4 waves per CU)
        HIP_TRY(hipMemcpy(w.data(), s->frames, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        if (FILE *f = std::fopen(path, "wb")) { std::fwrite(w.data(), sizeof(unsigned long long), w.size(), f); std::fclose(f); }
    }
#endif
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, s->host_e0, s->host_e1));
        stats->kernel_ms = ms;
        if (count) {
            unsigned long long c[CNT_SLOTS_DEV];
            HIP_TRY(hipMemcpy(c, s->counters, sizeof c, hipMemcpyDeviceToHost));
            stats->rays_primary = c[CNT_PRIMARY]; stats->rays_reflect = c[CNT_REFLECT]; stats->rays_refract = c[CNT_REFRACT];
            stats->rays_refract_exit = c[CNT_REFRACT_EXIT]; stats->rays_shadow = c[CNT_SHADOW];
            stats->nodes_tested = c[CNT_NODES]; stats->tris_tested = c[CNT_TRIS]; stats->prims_tested = c[CNT_PRIMS];
            stats->shadow_skipped = c[CNT_SHADOW_SKIPPED];
            stats->shaded_hits = c[CNT_SHADED]; stats->pixels = c[CNT_PIXELS] / (unsigned long long)std::max(p->spp, 1);  // the kernel counts samples
            if (g_cfg.debug_counters)  // development: raw loop counters of a -DRAYHS_PROFILE_LOOPS build
                std::fprintf(stderr, "rayhs adv[0..3] %llu %llu %llu %llu span_ticks %llu latest_end %llu earliest_start %llu counters[6..15]: %llu %llu | %llu %llu | %llu %llu %llu %llu %llu %llu  kernel_ms %.3f\n", c[0], c[1], c[2], c[3], c[5] - ~c[4], c[5], ~c[4], c[6], c[7], c[8], c[9], c[10], c[11], c[12], c[13], c[14], c[15], (double)ms);
            if (g_cfg.debug_counters)  // (active lanes, wave executions) of: rounds, plane tests, plane divisions, sphere tests, hit, light, return sections, triangle tests
                std::fprintf(stderr, "rayhs sections[16..31]: %llu %llu | %llu %llu | %llu %llu | %llu %llu | %llu %llu | %llu %llu | %llu %llu | %llu %llu\n", c[16], c[17], c[18], c[19],
                             c[20], c[21], c[22], c[23], c[24], c[25], c[26], c[27], c[28], c[29], c[30], c[31]);
            if (g_cfg.debug_counters)
                std::fprintf(stderr, "rayhs traversal[32..39]: in-traversal %llu %llu | holding-leaf %llu %llu | lane-visits in passes <16 / <32 lanes: %llu %llu | entering %llu %llu\n",
                             c[32], c[33], c[34], c[35], c[36], c[37], c[38], c[39]);
            if (g_cfg.debug_counters)
                std::fprintf(stderr, "rayhs packet[40..43]: packet visits (lanes, waves) %llu %llu | %llu %llu\n", c[40], c[41], c[42], c[43]);
            if (g_cfg.debug_counters)  // per-lane-walk census of a -DRAYHS_PROFILE_LOOPS build (trace_kernel.hip, traverse())
                std::fprintf(stderr, "rayhs perlane[44..57]: walks by entering lanes <=4 / <=16 / <=32 / <=64: %llu %llu %llu %llu | node-loop iterations: %llu %llu %llu %llu | "
                                     "sibling waves of the workgroup inside a per-lane walk when one starts, 0 / 1 / 2 / 3: %llu %llu %llu %llu | lanes walking in those siblings %llu, "
                                     "entering lanes of the starting walks %llu\n", c[44], c[45], c[46], c[47], c[48], c[49], c[50], c[51], c[52], c[53], c[54], c[55], c[56], c[57]);
        }
    }
    return RAYHS_OK;
}

// ------------------------------------------------------------ several GPUs, one process
// The reference's only parallel construct is the pixel map (generatePixels, src/Image.hs:34-36).  Here the rows of one
// frame are cut into stripes, every GPU of the node traces its share from its own copy of the scene (no collective on
// the data path: stripes are independent), and pushes the finished stripes into the whole frame on the first GPU with
// peer writes over xGMI.  Stripe ownership is balanced by the static tile estimate (rayhs_static_costs), not dealt
// round-robin: the model sits in the middle rows.
struct rayhs_multi {
    std::vector<int> devices;
    std::vector<rayhs_scene *> scenes;
    std::vector<hipStream_t> streams;
    std::vector<hipEvent_t> e0, e1, done;
    std::vector<void *> out_dev;          // compact stripes of each device
    std::vector<size_t> out_bytes;
    std::vector<int32_t *> map_dev;
    std::vector<size_t> map_cap;
    std::vector<unsigned long long *> counters;
    std::vector<StripeList> lists;
    std::vector<double> est_share;         // estimated share of the frame's cost per device (last plan)
    std::vector<double> kernel_ms;         // trace-kernel time of each device in the last frame rendered with stats
    void *frame = nullptr;                 // whole frame, on devices[0]
    size_t frame_bytes = 0;
    float *cost0 = nullptr;                // static estimate of the whole frame, on devices[0]
    size_t cost_cap = 0;
    std::vector<unsigned char> plan_key;   // geometry + camera the stripe lists were planned for
    int next_list_id = 1;
};

extern "C" void rayhs_multi_destroy(rayhs_multi *m) {
    if (!m) return;
    for (size_t i = 0; i < m->devices.size(); i++) {
        (void)hipSetDevice(m->devices[i]);
        if (i < m->streams.size() && m->streams[i]) { (void)hipStreamSynchronize(m->streams[i]); (void)hipStreamDestroy(m->streams[i]); }
        if (i < m->e0.size() && m->e0[i]) (void)hipEventDestroy(m->e0[i]);
        if (i < m->e1.size() && m->e1[i]) (void)hipEventDestroy(m->e1[i]);
        if (i < m->done.size() && m->done[i]) (void)hipEventDestroy(m->done[i]);
        if (i < m->out_dev.size() && m->out_dev[i]) (void)hipFree(m->out_dev[i]);
        if (i < m->map_dev.size() && m->map_dev[i]) (void)hipFree(m->map_dev[i]);
        if (i < m->counters.size() && m->counters[i]) (void)hipFree(m->counters[i]);
        if (i < m->scenes.size() && m->scenes[i]) rayhs_scene_destroy(m->scenes[i]);
    }
    if (!m->devices.empty()) (void)hipSetDevice(m->devices[0]);
    if (m->frame) (void)hipFree(m->frame);
    if (m->cost0) (void)hipFree(m->cost0);
    delete m;
}

extern "C" int rayhs_multi_create(const rayhs_scene_desc *desc, const int *devices, int n_devices, rayhs_multi **out) {
    if (!out) return set_error(RAYHS_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return set_error(RAYHS_ERR_INVALID, "need 1..64 devices");
    int ndev = rayhs_device_count();
    if (ndev <= 0) return set_error(RAYHS_ERR_NO_DEVICE, "no HIP device available (librayhs_hip has no CPU fallback)");
    for (int i = 0; i < n_devices; i++)
        if (devices[i] < 0 || devices[i] >= ndev) return set_error(RAYHS_ERR_INVALID, "device %d out of range (have %d)", devices[i], ndev);
    rayhs_multi *m = new rayhs_multi();
    const size_t n = (size_t)n_devices;
    m->devices.assign(devices, devices + n_devices);
    m->scenes.assign(n, nullptr); m->streams.assign(n, nullptr); m->e0.assign(n, nullptr); m->e1.assign(n, nullptr); m->done.assign(n, nullptr);
    m->out_dev.assign(n, nullptr); m->out_bytes.assign(n, 0); m->map_dev.assign(n, nullptr); m->map_cap.assign(n, 0); m->counters.assign(n, nullptr);
    m->lists.assign(n, StripeList()); m->est_share.assign(n, 0.0); m->kernel_ms.assign(n, 0.0);
    auto fail = [&](int code) { rayhs_multi_destroy(m); return code; };
    // host preparation (validation, records, triangle ranks, SAH BVH) once for all entries ...
    PackedScene pk;
    {
        int rc = pack_scene(desc, pk);
        if (rc) return fail(rc);
    }
    // ... then one upload per entry, the entries side by side (one host thread each: the copies of different devices do
    // not wait for each other).  The same device may be listed more than once (a one-GPU rehearsal of the N-GPU path):
    // every entry gets its own scene copy, stream and buffers, exactly as N different GPUs would.
    std::vector<int> codes(n, RAYHS_OK);
    std::vector<std::string> messages(n);
    auto bring_up = [&](size_t i) {
        int rc = upload_scene(pk, devices[i], &m->scenes[i]);
        if (rc == RAYHS_OK &&
            (hipStreamCreateWithFlags(&m->streams[i], hipStreamNonBlocking) != hipSuccess || hipEventCreate(&m->e0[i]) != hipSuccess ||
             hipEventCreate(&m->e1[i]) != hipSuccess || hipEventCreateWithFlags(&m->done[i], hipEventDisableTiming) != hipSuccess))
            rc = set_error(RAYHS_ERR_HIP, "stream / event creation failed on device %d", devices[i]);
        void *p = nullptr;
        if (rc == RAYHS_OK && hipMalloc(&p, sizeof(unsigned long long) * CNT_SLOTS_DEV) != hipSuccess) rc = set_error(RAYHS_ERR_HIP, "hipMalloc(counters) failed");
        if (rc == RAYHS_OK) m->counters[i] = (unsigned long long *)p;
        if (rc == RAYHS_OK && devices[i] != devices[0]) {  // this device will write its stripes into the frame on devices[0]
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) != hipSuccess || !can)
                rc = set_error(RAYHS_ERR_UNSUPPORTED, "device %d has no peer access to device %d", devices[i], devices[0]);
            else {
                hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) rc = set_error(RAYHS_ERR_HIP, "hipDeviceEnablePeerAccess failed: %s", hipGetErrorString(e));
                (void)hipGetLastError();
            }
        }
        codes[i] = rc;
        if (rc) messages[i] = g_error;  // (the error text is thread-local: carried back to the calling thread below)
    };
    // One thread per device is opt-in (RAYHS_MULTI_PARALLEL_CREATE): concurrent peer enabling across DISTINCT devices has
    // not run on hardware yet (every box this build has had shows one GPU; tests/test_gpu_parity.py has the two-GPU test
    // that will).  A thread that cannot be started must not take the process down through the extern "C" boundary:
    // whatever was started is joined and the remaining devices are brought up on the calling thread.
    if (n == 1 || !g_cfg.multi_parallel_create || g_cfg.multi_serial_create) {
        for (size_t i = 0; i < n; i++) bring_up(i);
    } else {
        std::vector<std::thread> workers;
        size_t started = 0;
        try {
            workers.reserve(n);
            for (; started < n; started++) workers.emplace_back(bring_up, started);
        } catch (...) {  // std::system_error (no thread to be had) or bad_alloc
        }
        for (auto &w : workers) w.join();
        for (size_t i = started; i < n; i++) bring_up(i);
    }
    for (size_t i = 0; i < n; i++)
        if (codes[i]) { g_error = messages[i]; return fail(codes[i]); }
    (void)hipSetDevice(devices[0]);
    *out = m;
    return RAYHS_OK;
}

extern "C" int rayhs_multi_device_count(const rayhs_multi *m) { return m ? (int)m->devices.size() : 0; }

// Stripe lists for this frame geometry and camera: static estimate of the whole frame on devices[0] -> cost per stripe
// -> longest-processing-time-first assignment -> one ascending list per device, uploaded.  Once per geometry / camera.
static int plan_stripes(rayhs_multi *m, const rayhs_camera *cam, const rayhs_render_params *p) {
    std::vector<unsigned char> key(sizeof(rayhs_camera) + 8 * sizeof(int));
    std::memcpy(key.data(), cam, sizeof(rayhs_camera));
    const int geo[8] = {p->width, p->height, p->row0, p->row1, p->stripe_rows, p->spp, p->max_depth, (int)(p->flags & RAYHS_FLAG_FMA)};
    std::memcpy(key.data() + sizeof(rayhs_camera), geo, sizeof geo);
    if (key == m->plan_key) return RAYHS_OK;
    const size_t n = m->devices.size();
    const int total = p->row1 - p->row0, R = p->stripe_rows, n_stripes = (total + R - 1) / R;
    std::vector<double> stripe_cost((size_t)n_stripes, 1.0);
    if (n > 1 && total > 0 && !g_cfg.round_robin_stripes) {
        rayhs_render_params whole = *p;
        whole.stripe_rows = std::max(total, 1); whole.stripe_rank = 0; whole.stripe_count = 1; whole.flags &= RAYHS_FLAG_FMA;
        DCamera dc; DParams dp; int variant, grid; bool exact; size_t lds;
        int rc = prepare_launch(m->scenes[0], cam, &whole, nullptr, dc, dp, variant, exact, grid, lds);
        if (rc) return rc;
        HIP_TRY(hipSetDevice(m->devices[0]));
        if ((size_t)dp.n_tiles > m->cost_cap) {
            if (m->cost0) HIP_TRY(hipFree(m->cost0));
            m->cost0 = nullptr; m->cost_cap = 0;
            void *q = nullptr;
            HIP_TRY(hipMalloc(&q, (size_t)dp.n_tiles * sizeof(float)));
            m->cost0 = (float *)q; m->cost_cap = (size_t)dp.n_tiles;
        }
        hipError_t e = exact ? launch_static_costs_exact(m->streams[0], m->scenes[0]->d, dc, dp, m->cost0) : launch_static_costs_fast(m->streams[0], m->scenes[0]->d, dc, dp, m->cost0);
        if (e != hipSuccess) return set_error(RAYHS_ERR_HIP, "static cost launch failed: %s", hipGetErrorString(e));
        std::vector<float> tile_cost((size_t)dp.n_tiles);
        HIP_TRY(hipMemcpyAsync(tile_cost.data(), m->cost0, tile_cost.size() * sizeof(float), hipMemcpyDeviceToHost, m->streams[0]));
        HIP_TRY(hipStreamSynchronize(m->streams[0]));
        const int tile_rows = (total + 7) / 8;
        std::vector<double> row_cost((size_t)tile_rows, 0.0);
        for (int ty = 0; ty < tile_rows; ty++)
            for (int tx = 0; tx < dp.tiles_x; tx++) row_cost[(size_t)ty] += tile_cost[(size_t)ty * (size_t)dp.tiles_x + (size_t)tx];
        for (int st = 0; st < n_stripes; st++) {  // a stripe gets the tile rows it overlaps, by overlap
            const int a = st * R, b = std::min(total, a + R);
            double c = 0;
            for (int ty = a / 8; ty <= (b - 1) / 8; ty++) {
                const int lo = std::max(a, ty * 8), hi = std::min(b, ty * 8 + 8);
                c += row_cost[(size_t)ty] * (double)(hi - lo) / 8.0;
            }
            stripe_cost[(size_t)st] = c;
        }
    }
    // Interleaving (stripe k to device k mod n) hands every device an even sample of the frame and is hard to beat: on
    // the dragon frames its shares came out within 2 % of each other at 4K and beat a pure cost-sorted assignment, whose
    // estimate errors do not average out (profiles/r02/stripe_balance.jsonl).  It is kept whenever the estimate calls it
    // balanced (no share above 103 % of the mean); otherwise -- few stripes per device, a model that beats against the
    // interleave period -- stripes are dealt longest-processing-time-first by estimated cost.
    std::vector<int> by_cost((size_t)n_stripes);
    for (int i = 0; i < n_stripes; i++) by_cost[(size_t)i] = i;
    std::stable_sort(by_cost.begin(), by_cost.end(), [&](int a, int b) { return stripe_cost[(size_t)a] > stripe_cost[(size_t)b]; });
    std::vector<std::vector<int32_t>> lists(n);
    std::vector<double> load(n, 0.0);
    bool round_robin = n == 1 || g_cfg.round_robin_stripes;
    if (!round_robin && !g_cfg.lpt_stripes) {
        std::vector<double> rr(n, 0.0);
        double total_cost = 0;
        for (int k = 0; k < n_stripes; k++) { rr[(size_t)k % n] += stripe_cost[(size_t)k]; total_cost += stripe_cost[(size_t)k]; }
        double worst = 0;
        for (double v : rr) worst = std::max(worst, v);
        round_robin = total_cost > 0 && worst <= 1.03 * total_cost / (double)n;
    }
    for (int k = 0; k < n_stripes; k++) {
        const int st = round_robin ? k : by_cost[(size_t)k];
        size_t best = 0;
        if (round_robin) best = (size_t)k % n;
        else
            for (size_t i = 1; i < n; i++)
                if (load[i] < load[best]) best = i;
        lists[best].push_back(st);
        load[best] += stripe_cost[(size_t)st];
    }
    double sum = 0;
    for (double v : load) sum += v;
    for (size_t i = 0; i < n; i++) {
        std::sort(lists[i].begin(), lists[i].end());
        m->est_share[i] = sum > 0 ? load[i] / sum : 0.0;
        StripeList &sl = m->lists[i];
        sl.n_stripes = (int)lists[i].size();
        sl.rows = 0;
        for (int st : lists[i]) sl.rows += std::min(R, total - st * R);
        sl.id = m->next_list_id++;
        HIP_TRY(hipSetDevice(m->devices[i]));
        if (lists[i].size() > m->map_cap[i]) {
            if (m->map_dev[i]) HIP_TRY(hipFree(m->map_dev[i]));
            m->map_dev[i] = nullptr; m->map_cap[i] = 0;
            void *q = nullptr;
            HIP_TRY(hipMalloc(&q, lists[i].size() * sizeof(int32_t)));
            m->map_dev[i] = (int32_t *)q; m->map_cap[i] = lists[i].size();
        }
        if (!lists[i].empty()) {
            HIP_TRY(hipStreamSynchronize(m->streams[i]));  // a previous frame may still read the old list
            HIP_TRY(hipMemcpy(m->map_dev[i], lists[i].data(), lists[i].size() * sizeof(int32_t), hipMemcpyHostToDevice));
        }
        sl.d_map = m->map_dev[i];
    }
    m->plan_key = key;
    return RAYHS_OK;
}

// shares[i] = estimated fraction of the frame's cost given to device i by the last plan (diagnostics / tests)
extern "C" int rayhs_multi_shares(const rayhs_multi *m, double *shares, int *rows, double *kernel_ms, int n) {
    if (!m || n < (int)m->devices.size()) return set_error(RAYHS_ERR_INVALID, "bad argument");
    for (size_t i = 0; i < m->devices.size(); i++) {
        if (shares) shares[i] = m->est_share[i];
        if (rows) rows[i] = m->lists[i].rows;
        if (kernel_ms) kernel_ms[i] = m->kernel_ms[i];
    }
    return RAYHS_OK;
}

static int multi_render(rayhs_multi *m, const rayhs_camera *cam, const rayhs_render_params *p_in, void *out_host, void *d_out, rayhs_stats *stats) {
    if (!m || !cam || !p_in || (!out_host && !d_out)) return set_error(RAYHS_ERR_INVALID, "NULL argument");
    rayhs_render_params p = *p_in;
    if (p.stripe_rows < 1) p.stripe_rows = 8;
    p.stripe_rank = 0; p.stripe_count = 1;  // ownership is the library's here
    if (p.width <= 0 || p.height <= 0 || p.row0 < 0 || p.row1 < p.row0 || p.row1 > p.height) return set_error(RAYHS_ERR_INVALID, "bad width/height/row range");
    if (p.out_format < RAYHS_OUT_F32 || p.out_format > RAYHS_OUT_I16) return set_error(RAYHS_ERR_INVALID, "bad out_format");
    if (p.flags & RAYHS_FLAG_EXHAUSTIVE && !(p.flags & RAYHS_FLAG_COUNT)) return set_error(RAYHS_ERR_INVALID, "RAYHS_FLAG_EXHAUSTIVE needs RAYHS_FLAG_COUNT");
    const size_t n = m->devices.size();
    const int total = p.row1 - p.row0;
    const size_t esz = p.out_format == RAYHS_OUT_F64 ? 8 : p.out_format == RAYHS_OUT_F32 ? 4 : 2;
    const size_t row_bytes = (size_t)p.width * 3 * esz, bytes = row_bytes * (size_t)total;
    int rc = plan_stripes(m, cam, &p);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(m->devices[0]));
    void *frame = d_out;
    if (!frame) {
        if (bytes > m->frame_bytes || !m->frame) {
            if (m->frame) HIP_TRY(hipFree(m->frame));
            m->frame = nullptr; m->frame_bytes = 0;
            HIP_TRY(hipMalloc(&m->frame, std::max<size_t>(bytes, 16)));
            m->frame_bytes = std::max<size_t>(bytes, 16);
        }
        frame = m->frame;
    }
    const bool count = (p.flags & RAYHS_FLAG_COUNT) != 0, exact = (p.flags & RAYHS_FLAG_FMA) == 0;
    // One device's share: trace its stripes, push them into the frame on devices[0].  (A lambda so that a failure on
    // device i can be handled in ONE place below: the devices launched before it are still tracing and scattering
    // into the caller's frame, and the call must not return while they do.)
    auto launch_share = [&](size_t i) -> int {
        const StripeList &sl = m->lists[i];
        HIP_TRY(hipSetDevice(m->devices[i]));
        if (count) HIP_TRY(hipMemsetAsync(m->counters[i], 0, sizeof(unsigned long long) * CNT_SLOTS_DEV, m->streams[i]));
        const size_t need = row_bytes * (size_t)sl.rows;
        if (need > m->out_bytes[i] || !m->out_dev[i]) {
            if (m->out_dev[i]) { HIP_TRY(hipStreamSynchronize(m->streams[i])); HIP_TRY(hipFree(m->out_dev[i])); }
            m->out_dev[i] = nullptr; m->out_bytes[i] = 0;
            HIP_TRY(hipMalloc(&m->out_dev[i], std::max<size_t>(need, 16)));
            m->out_bytes[i] = std::max<size_t>(need, 16);
        }
        int rc_i = launch(m->scenes[i], cam, &p, m->out_dev[i], m->streams[i], count ? (uint64_t *)m->counters[i] : nullptr, m->e0[i], m->e1[i], &sl);
        if (rc_i) return rc_i;
        // push the finished stripes to their rows of the frame on devices[0] (peer write over xGMI)
        hipError_t e = exact ? launch_scatter_stripes_exact(m->streams[i], m->out_dev[i], frame, sl.d_map, sl.rows, p.stripe_rows, (long long)row_bytes)
                             : launch_scatter_stripes_fast(m->streams[i], m->out_dev[i], frame, sl.d_map, sl.rows, p.stripe_rows, (long long)row_bytes);
        if (e != hipSuccess) return set_error(RAYHS_ERR_HIP, "stripe scatter launch failed: %s", hipGetErrorString(e));
        HIP_TRY(hipEventRecord(m->done[i], m->streams[i]));
        // development (tools/stripe_balance.py): one share at a time, so that entries sharing a GPU are timed as if each had its own
        if (g_cfg.multi_serial) HIP_TRY(hipStreamSynchronize(m->streams[i]));
        return RAYHS_OK;
    };
    for (size_t i = 0; i < n; i++) {
        rc = launch_share(i);
        if (rc) {
            const std::string why = g_error;
            for (size_t k = 0; k <= i; k++)  // the shares already under way finish before the caller gets its buffer back
                if (hipSetDevice(m->devices[k]) == hipSuccess) (void)hipStreamSynchronize(m->streams[k]);
            (void)hipSetDevice(m->devices[0]);
            g_error = why;
            return rc;
        }
    }
    HIP_TRY(hipSetDevice(m->devices[0]));
    for (size_t i = 1; i < n; i++) HIP_TRY(hipStreamWaitEvent(m->streams[0], m->done[i], 0));
    if (out_host && bytes) HIP_TRY(hipMemcpyAsync(out_host, frame, bytes, hipMemcpyDeviceToHost, m->streams[0]));
    HIP_TRY(hipStreamSynchronize(m->streams[0]));
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (size_t i = 0; i < n; i++) {
            HIP_TRY(hipSetDevice(m->devices[i]));
            HIP_TRY(hipStreamSynchronize(m->streams[i]));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, m->e0[i], m->e1[i]));
            m->kernel_ms[i] = (double)ms;
            stats->kernel_ms = std::max(stats->kernel_ms, (double)ms);  // the slowest device's trace kernel
            if (count) {
                unsigned long long c[CNT_SLOTS_DEV];
                HIP_TRY(hipMemcpy(c, m->counters[i], sizeof c, hipMemcpyDeviceToHost));
                stats->rays_primary += c[CNT_PRIMARY]; stats->rays_reflect += c[CNT_REFLECT]; stats->rays_refract += c[CNT_REFRACT];
                stats->rays_refract_exit += c[CNT_REFRACT_EXIT]; stats->rays_shadow += c[CNT_SHADOW];
                stats->nodes_tested += c[CNT_NODES]; stats->tris_tested += c[CNT_TRIS]; stats->prims_tested += c[CNT_PRIMS];
                stats->shadow_skipped += c[CNT_SHADOW_SKIPPED];
                stats->shaded_hits += c[CNT_SHADED]; stats->pixels += c[CNT_PIXELS] / (unsigned long long)std::max(p.spp, 1);
            }
        }
        HIP_TRY(hipSetDevice(m->devices[0]));
    }
    return RAYHS_OK;
}

extern "C" int rayhs_multi_render(rayhs_multi *m, const rayhs_camera *cam, const rayhs_render_params *p, void *out_host, rayhs_stats *stats) {
    if (!out_host) return set_error(RAYHS_ERR_INVALID, "out_host is NULL");
    return multi_render(m, cam, p, out_host, nullptr, stats);
}
extern "C" int rayhs_multi_render_device(rayhs_multi *m, const rayhs_camera *cam, const rayhs_render_params *p, void *d_out, rayhs_stats *stats) {
    if (!d_out) return set_error(RAYHS_ERR_INVALID, "d_out is NULL");
    return multi_render(m, cam, p, nullptr, d_out, stats);
}

// ----------------------------------------------------------------- host prep
struct rayhs_render_desc { std::unique_ptr<RenderDescData> data; };

template <class F>
static int guarded(F f) {
    try {
        return f();
    } catch (const SceneError &e) {
        return set_error(e.code, "%s", e.what());
    } catch (const std::exception &e) {
        return set_error(RAYHS_ERR_INVALID, "%s", e.what());
    }
}

extern "C" int rayhs_render_desc_load(const char *json_path, rayhs_render_desc **out) {
    if (!json_path || !out) return set_error(RAYHS_ERR_INVALID, "NULL argument");
    *out = nullptr;
    return guarded([&] { auto d = load_render_desc(json_path); *out = new rayhs_render_desc{std::move(d)}; return (int)RAYHS_OK; });
}
extern "C" int rayhs_render_desc_parse(const char *json_text, rayhs_render_desc **out) {
    if (!json_text || !out) return set_error(RAYHS_ERR_INVALID, "NULL argument");
    *out = nullptr;
    return guarded([&] { auto d = decode_render_desc(json_text); *out = new rayhs_render_desc{std::move(d)}; return (int)RAYHS_OK; });
}
extern "C" void rayhs_render_desc_free(rayhs_render_desc *rd) { delete rd; }
extern "C" const rayhs_scene_desc *rayhs_render_desc_scene(const rayhs_render_desc *rd) { return rd ? &rd->data->flat.view : nullptr; }
extern "C" const rayhs_camera *rayhs_render_desc_camera(const rayhs_render_desc *rd) { return rd ? &rd->data->camera : nullptr; }
extern "C" int rayhs_render_desc_width(const rayhs_render_desc *rd) { return rd ? rd->data->width : 0; }
extern "C" int rayhs_render_desc_height(const rayhs_render_desc *rd) { return rd ? rd->data->height : 0; }
extern "C" int rayhs_render_desc_max_depth(const rayhs_render_desc *rd) { return rd ? rd->data->max_depth : 0; }
extern "C" int rayhs_render_desc_tree_depth(const rayhs_render_desc *rd) { return rd ? rd->data->tree_depth : -1; }

extern "C" int rayhs_write_ppm_i16(const char *path, int w, int h, const int16_t *q) {
    if (!path || !q || w <= 0 || h <= 0) return set_error(RAYHS_ERR_INVALID, "bad argument");
    return guarded([&] { write_ppm_i16(path, w, h, q); return (int)RAYHS_OK; });
}
extern "C" int rayhs_write_ppm_f64(const char *path, int w, int h, const double *rgb) {
    if (!path || !rgb || w <= 0 || h <= 0) return set_error(RAYHS_ERR_INVALID, "bad argument");
    return guarded([&] { write_ppm_f64(path, w, h, rgb); return (int)RAYHS_OK; });
}

static uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
extern "C" double rayhs_sample01(uint64_t seed, uint64_t pixel, uint32_t sample, uint32_t dim, uint32_t spp) {
    uint64_t ctr = (pixel * (uint64_t)spp + sample) * 2u + dim;
    uint64_t x = mix64(ctr + mix64(seed + 0x9e3779b97f4a7c15ULL));
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}
