// cli.cpp -- `rayhs [-oFILE|--output=FILE] SCENE.json`, the process-level
// drop-in for the reference executable (main, src/RayHs.hs:204-234).
//
// Same stdout lines, same default output name (out.ppm), same ASCII P3 file.
// Option parsing follows GetOpt RequireOrder with an OptArg: only -oFILE and
// --output=FILE bind a value; a bare -o keeps the default.  As an extension,
// `-o FILE SCENE.json` (the spelling BASELINE.json uses) is accepted when two
// more arguments follow.  Extra long options that the reference lacks:
//   --spp=N --seed=S   multi-sample path (distributedTracePixel semantics)
//   --device=N         HIP device (default 0)
//   --gpus=N           trace the frame on GPUs 0..N-1 of this node (stripes per GPU, assembled on GPU 0)
//   --devices=a,b,...  the same with an explicit device list (a device may repeat: one-GPU rehearsal)
//   --stats            work counters and kernel time on stderr
//   --timing           wall time of each stage (load, upload, render, write) on stderr
// The trace runs on the GPU through the C ABI; there is no CPU path.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rayhs.h"

// The HIP runtime, the device context and the kernels' code object come up on a second thread while this one decodes
// the scene and builds the trees (0.1-0.2 s of start-up against ~35 ms of host preparation: they used to run one after
// the other).  Joined before the process leaves by any path, so that it never exits under a half-initialised runtime.
static std::thread g_warmup;
static void finish_warmup() { if (g_warmup.joinable()) g_warmup.join(); }

[[noreturn]] static void die(const std::string &msg) {
    // `error` in the reference: message on stderr, exit status 1
    std::fprintf(stderr, "rayhs: %s\n", msg.c_str());
    finish_warmup();
    std::exit(1);
}

static const char *kUsage =
    "Usage: main [OPTION...]\n"
    "  -o[FILE]  --output[=FILE]  Specify the ouput filename.\n";

int main(int argc, char **argv) {
    std::string out_file = "out.ppm";
    bool have_output_flag = false;
    int spp = 1, device = 0;
    std::vector<int> devices;  // non-empty: the multi-GPU path
    unsigned long long seed = 24;  // src/RayHs.hs:239
    bool want_stats = false, want_timing = false;
    std::vector<std::string> errors;
    int i = 1;
    for (; i < argc; i++) {  // RequireOrder: options stop at the first non-option
        std::string a = argv[i];
        if (a == "--") { i++; break; }
        if (a.size() < 2 || a[0] != '-') break;
        std::string val;
        bool has_val = false;
        auto long_opt = [&](const char *name) {
            std::string n = std::string("--") + name;
            if (a == n) return true;
            if (a.compare(0, n.size() + 1, n + "=") == 0) { val = a.substr(n.size() + 1); has_val = true; return true; }
            return false;
        };
        if (a.compare(0, 2, "-o") == 0 && a[1] == 'o') {
            std::string v = a.substr(2);
            if (v.empty() && argc - i - 1 >= 2) v = argv[++i];  // extension: `-o FILE SCENE`
            if (!have_output_flag) out_file = v.empty() ? "out.ppm" : v;  // first Output flag wins (src/RayHs.hs:220-222)
            have_output_flag = true;
        } else if (long_opt("output")) {
            if (!have_output_flag) out_file = has_val ? val : "out.ppm";
            have_output_flag = true;
        } else if (long_opt("spp")) { spp = has_val ? std::atoi(val.c_str()) : 1; }
        else if (long_opt("seed")) { seed = has_val ? std::strtoull(val.c_str(), nullptr, 10) : 24; }
        else if (long_opt("device")) { device = has_val ? std::atoi(val.c_str()) : 0; }
        else if (long_opt("gpus")) {
            const int n = has_val ? std::atoi(val.c_str()) : 1;
            if (n < 1 || n > 64) errors.push_back("option `--gpus' needs a count between 1 and 64\n");
            else { devices.clear(); for (int d = 0; d < n; d++) devices.push_back(d); }
        } else if (long_opt("devices")) {
            devices.clear();
            for (size_t a0 = 0; has_val && a0 <= val.size();) {
                size_t b0 = val.find(',', a0);
                if (b0 == std::string::npos) b0 = val.size();
                if (b0 > a0) devices.push_back(std::atoi(val.substr(a0, b0 - a0).c_str()));
                a0 = b0 + 1;
            }
            if (devices.empty()) errors.push_back("option `--devices' needs a comma-separated device list\n");
        }
        else if (long_opt("stats")) { want_stats = true; }
        else if (long_opt("timing")) { want_timing = true; }
        else errors.push_back("unrecognized option `" + a + "'\n");
    }
    if (!errors.empty()) {
        std::string m;
        for (auto &e : errors) m += e;
        die(m + kUsage);
    }
    if (i >= argc) die("No input.");
    const std::string scene_file = argv[i];

    using clk = std::chrono::steady_clock;
    auto ms_since = [](clk::time_point t0) { return std::chrono::duration<double, std::milli>(clk::now() - t0).count(); };
    const clk::time_point t_start = clk::now();
    double ms_load = 0, ms_upload = 0, ms_render = 0, ms_write = 0, ms_kernel = 0;
    std::printf("Loading scene from %s...\n", scene_file.c_str());
    std::fflush(stdout);
    {
        std::vector<int> warm = devices.empty() ? std::vector<int>{device} : devices;
        g_warmup = std::thread([warm] { for (int d : warm) (void)rayhs_device_prepare(d); });  // errors surface from the calls below
    }
    rayhs_render_desc *rd = nullptr;
    int rc = rayhs_render_desc_load(scene_file.c_str(), &rd);
    if (rc == RAYHS_ERR_PARSE && std::strncmp(rayhs_last_error(), "Failed to read scene", 20) == 0) {
        std::fprintf(stderr, "rayhs: Failed to read scene\n");
        if (std::getenv("RAYHS_VERBOSE")) std::fprintf(stderr, "  %s\n", rayhs_last_error());
        finish_warmup();
        return 1;
    }
    // the reference prints "Rendering..." once the JSON decoded, before meshes are read
    std::printf("Rendering...\n");
    std::fflush(stdout);
    if (rc != RAYHS_OK) die(rayhs_last_error());

    ms_load = ms_since(t_start);
    clk::time_point t0 = clk::now();  // (the warm-up thread may still be running: the HIP runtime serialises what must be)
    rayhs_scene *scene = nullptr;
    rayhs_multi *multi = nullptr;
    if (devices.empty()) {
        if (rayhs_scene_create(rayhs_render_desc_scene(rd), device, &scene) != RAYHS_OK) die(rayhs_last_error());
    } else if (rayhs_multi_create(rayhs_render_desc_scene(rd), devices.data(), (int)devices.size(), &multi) != RAYHS_OK) die(rayhs_last_error());
    ms_upload = ms_since(t0);
    rayhs_render_params p;
    std::memset(&p, 0, sizeof p);
    p.width = rayhs_render_desc_width(rd);
    p.height = rayhs_render_desc_height(rd);
    p.max_depth = rayhs_render_desc_max_depth(rd);
    p.spp = spp;
    p.seed = seed;
    p.row0 = 0;
    p.row1 = p.height;
    p.stripe_rows = multi ? 8 : (p.height > 0 ? p.height : 1);
    p.stripe_rank = 0;
    p.stripe_count = 1;
    p.out_format = RAYHS_OUT_I16;
    p.flags = want_stats ? RAYHS_FLAG_COUNT : 0;
    if (p.width <= 0 || p.height <= 0) {
        // an empty image in the reference is just a header
        std::vector<int16_t> none;
        FILE *f = std::fopen(out_file.c_str(), "wb");
        if (!f) die(out_file + ": openFile: permission denied");
        std::fprintf(f, "P3\n%d %d\n255\n", p.width, p.height);
        std::fclose(f);
    } else {
        std::vector<int16_t> q((size_t)p.width * (size_t)p.height * 3);
        rayhs_stats st;
        t0 = clk::now();
        if ((multi ? rayhs_multi_render(multi, rayhs_render_desc_camera(rd), &p, q.data(), &st)
                   : rayhs_render(scene, rayhs_render_desc_camera(rd), &p, q.data(), &st)) != RAYHS_OK) die(rayhs_last_error());
        ms_render = ms_since(t0);
        ms_kernel = st.kernel_ms;  // the trace kernel of this one frame: a first frame, ordered by the static tile estimate
        t0 = clk::now();
        if (rayhs_write_ppm_i16(out_file.c_str(), p.width, p.height, q.data()) != RAYHS_OK) die(rayhs_last_error());
        ms_write = ms_since(t0);
        if (want_stats) {
            unsigned long long rays = st.rays_primary + st.rays_reflect + st.rays_refract + st.rays_refract_exit + st.rays_shadow;
            std::fprintf(stderr,
                         "{\"kernel_ms\": %.3f, \"rays\": %llu, \"rays_primary\": %llu, \"rays_reflect\": %llu, \"rays_refract\": %llu, "
                         "\"rays_refract_exit\": %llu, \"rays_shadow\": %llu, \"nodes_tested\": %llu, \"tris_tested\": %llu, "
                         "\"prims_tested\": %llu, \"shaded_hits\": %llu, \"pixels\": %llu}\n",
                         st.kernel_ms, rays, (unsigned long long)st.rays_primary, (unsigned long long)st.rays_reflect,
                         (unsigned long long)st.rays_refract, (unsigned long long)st.rays_refract_exit, (unsigned long long)st.rays_shadow,
                         (unsigned long long)st.nodes_tested, (unsigned long long)st.tris_tested, (unsigned long long)st.prims_tested,
                         (unsigned long long)st.shaded_hits, (unsigned long long)st.pixels);
        }
    }
    finish_warmup();
    rayhs_scene_destroy(scene);
    rayhs_multi_destroy(multi);
    rayhs_render_desc_free(rd);
    if (want_timing)
        std::fprintf(stderr, "{\"load_ms\": %.1f, \"scene_create_ms\": %.1f, \"render_ms\": %.1f, \"first_frame_kernel_ms\": %.3f, \"write_ms\": %.1f, \"total_ms\": %.1f}\n",
                     ms_load, ms_upload, ms_render, ms_kernel, ms_write, ms_since(t_start));
    std::printf("Done! Output written to %s\n", out_file.c_str());
    return 0;
}
