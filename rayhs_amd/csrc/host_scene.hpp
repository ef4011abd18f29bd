// host_scene.hpp -- C++ side of the host preparation (see host_scene.cpp).
#pragma once
#include <cstdint>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/rayhs.h"
#include "json_min.hpp"

namespace rayhs {

struct SceneError : std::runtime_error {
    int code;
    SceneError(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

// Owner of the SoA arrays that rayhs_scene_desc points into.
struct FlatScene {
    std::vector<int32_t> obj_geom_kind, obj_geom_index, mat_kind, cmap_kind, tex_w, tex_h;
    std::vector<int64_t> tex_offset;
    std::vector<double> mat_ior, mat_ce, cmap_c1, cmap_c2, cmap_size, texels;
    std::vector<double> plane_point, plane_normal, plane_tangent, sphere_center, sphere_radius;
    std::vector<int32_t> mesh_root, node_left, node_right, node_first_tri, node_tri_count, node_axis;
    std::vector<double> node_lo, node_hi, tri_pos, tri_nrm, tri_uv;
    std::vector<int32_t> light_kind;
    std::vector<double> light_vec, light_color, light_radius;
    rayhs_scene_desc view{};
    void refresh_view();
};

struct RenderDescData {
    FlatScene flat;
    rayhs_camera camera{};
    int width = 0, height = 0, max_depth = 0, tree_depth = -1;
};

std::unique_ptr<RenderDescData> decode_render_desc(const std::string &json_text);
std::unique_ptr<RenderDescData> load_render_desc(const std::string &path);

int to_int_c(double c);
void write_ppm_i16(const std::string &path, int w, int h, const int16_t *q);
void write_ppm_f64(const std::string &path, int w, int h, const double *rgb);

}  // namespace rayhs
