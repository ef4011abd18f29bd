// bvh_build.hpp -- host-side builder of the fast traversal structure (see bvh_build.cpp).
#pragma once
#include <cstdint>
#include <vector>

namespace rayhs {

struct FNodeBuild {
    double lo0[3], hi0[3], lo1[3], hi1[3];  // child boxes, exact fp64 bounds (quantised later)
    int32_t c0, c1;                          // >= 0 inner node, < 0 leaf ~((first << 3) | count), INT32_MIN empty
};

struct FastBvh {
    std::vector<FNodeBuild> nodes;  // node 0 is the root (if any)
    std::vector<int> order;         // BVH leaf order -> index of the input triangle
    int32_t root = INT32_MIN;       // INT32_MIN: no triangles
    int depth = 0;                  // deepest inner-node level
    double lo[3], hi[3];            // bounds of all triangles
};

// tri_pos: 9 doubles per triangle; builds over the triangles listed in tri_ids.
// out.order[i] is a position in tri_ids (leaf i-th slot holds triangle tri_ids[out.order[i]]).
void build_fast_bvh(const double *tri_pos, const std::vector<int> &tri_ids, FastBvh &out);

float round_down_f32(double v);
float round_up_f32(double v);

}  // namespace rayhs
