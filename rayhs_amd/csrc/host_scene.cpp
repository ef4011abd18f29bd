// host_scene.cpp -- host preparation for the trace kernels: decode the render
// description, load meshes, build the bounding-box trees and flatten everything
// into the SoA arrays of rayhs_scene_desc (include/rayhs.h).
//
// This is the part the reference does in Haskell before rayTrace runs
// (src/JSON.hs:22-141, src/Descriptors.hs:39-55, src/MaterialDescriptors.hs:27-45,
// src/Mesh.hs:84-221, src/Transform.hs, src/Mat.hs:52-87, src/KDTree.hs:16-37,68-90,
// src/Bitmap.hs:31-37).  No GHC exists in this image, so the host is C++; a
// Haskell host would produce the same arrays and call rayhs_scene_create.
// All arithmetic is fp64 in the reference's operation order (compiled with
// -ffp-contract=off) so that the flattened geometry is bit-identical to what
// the reference would hold.
#include "host_scene.hpp"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <sstream>
#include <thread>

namespace rayhs {

// ---------------------------------------------------------------- small math
struct D3 { double x, y, z; };
static inline D3 add(D3 a, D3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline D3 scl(double l, D3 a) { return {l * a.x, l * a.y, l * a.z}; }
static inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline D3 unit(D3 a) { return scl(1 / std::sqrt(dot(a, a)), a); }  // src/Vec.hs:124-126
// Haskell's Ord Double: min x y = if x <= y then x else y; max x y = if x <= y then y else x
static inline double hs_min(double x, double y) { return x <= y ? x : y; }
static inline double hs_max(double x, double y) { return x <= y ? y : x; }

struct M3 { double m[9]; };  // row-major, src/Mat.hs:15-18
static D3 mul(const M3 &a, D3 v) {  // src/Mat.hs:40-44
    return {a.m[0] * v.x + a.m[1] * v.y + a.m[2] * v.z, a.m[3] * v.x + a.m[4] * v.y + a.m[5] * v.z,
            a.m[6] * v.x + a.m[7] * v.y + a.m[8] * v.z};
}
static M3 mul(const M3 &p, const M3 &q) {  // src/Mat.hs:25-31
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.m[3 * i + j] = p.m[3 * i] * q.m[j] + p.m[3 * i + 1] * q.m[3 + j] + p.m[3 * i + 2] * q.m[6 + j];
    return r;
}
static M3 transposed(const M3 &a) { return {{a.m[0], a.m[3], a.m[6], a.m[1], a.m[4], a.m[7], a.m[2], a.m[5], a.m[8]}}; }
static M3 columns(D3 a, D3 b, D3 c) { return {{a.x, b.x, c.x, a.y, b.y, c.y, a.z, b.z, c.z}}; }  // src/Mat.hs:83-87
static M3 rot_x(double t) { return {{1, 0, 0, 0, std::cos(t), -(std::sin(t)), 0, std::sin(t), std::cos(t)}}; }  // src/Mat.hs:70-74
static M3 rot_y(double t) { return {{std::cos(t), 0, std::sin(t), 0, 1, 0, -(std::sin(t)), 0, std::cos(t)}}; }  // src/Mat.hs:64-68
static M3 rot_z(double t) { return {{std::cos(t), -(std::sin(t)), 0, std::sin(t), std::cos(t), 0, 0, 0, 1}}; }  // src/Mat.hs:58-62
static M3 rot_axis(D3 r, double angle) {  // src/Mat.hs:76-81 with orthonormal, src/Vec.hs:151-158
    D3 s;
    double ax = std::fabs(r.x), ay = std::fabs(r.y), az = std::fabs(r.z);
    if (ax < ay && ax < az) s = unit({0, -(r.z), r.y});
    else if (ay < ax && ay < az) s = unit({-(r.z), 0, r.x});
    else s = unit({-(r.y), r.x, 0});
    D3 t = cross(r, s);
    M3 m = columns(r, s, t);
    return mul(mul(transposed(m), rot_x(angle)), m);
}

// ------------------------------------------------------------ JSON decoding
// Mirrors aeson's `.:` -- a missing key or a value of the wrong shape fails the
// whole decode, which the reference reports as "Failed to read scene".
static const JsonValue &need(const JsonValue &o, const char *key) {
    if (o.kind != JsonValue::Object) throw JsonError(std::string("expected object holding '") + key + "'");
    const JsonValue *v = o.find(key);
    if (!v) throw JsonError(std::string("key '") + key + "' not present");
    return *v;
}
static double need_num(const JsonValue &o, const char *key) {
    const JsonValue &v = need(o, key);
    if (v.kind != JsonValue::Number) throw JsonError(std::string("'") + key + "' is not a number");
    return v.num;
}
static int need_int(const JsonValue &o, const char *key) {
    double d = need_num(o, key);
    if (!(std::floor(d) == d) || std::fabs(d) > 2147483647.0) throw JsonError(std::string("'") + key + "' is not an Int");
    return (int)d;
}
static const std::string &need_str(const JsonValue &o, const char *key) {
    const JsonValue &v = need(o, key);
    if (v.kind != JsonValue::String) throw JsonError(std::string("'") + key + "' is not a string");
    return v.str;
}
static D3 need_vec(const JsonValue &o, const char *key) {  // src/JSON.hs:22-27
    const JsonValue &v = need(o, key);
    return {need_num(v, "x"), need_num(v, "y"), need_num(v, "z")};
}
static D3 need_rgb(const JsonValue &o, const char *key) {  // src/JSON.hs:29-34
    const JsonValue &v = need(o, key);
    return {need_num(v, "r"), need_num(v, "g"), need_num(v, "b")};
}

struct Xform {  // src/Transform.hs:7-13
    enum Kind { Translate, Scale, Rotate, RotateX, RotateY, RotateZ, Sequence } kind;
    D3 v{0, 0, 0};
    double angle = 0;
    std::vector<Xform> seq;
};
static Xform decode_xform(const JsonValue &j) {  // src/JSON.hs:86-97
    const std::string &kind = need_str(j, "type");
    Xform x;
    if (kind == "scale") { x.kind = Xform::Scale; x.v = need_vec(j, "vector"); }
    else if (kind == "translate") { x.kind = Xform::Translate; x.v = need_vec(j, "vector"); }
    else if (kind == "rotateX") { x.kind = Xform::RotateX; x.angle = need_num(j, "angle"); }
    else if (kind == "rotateY") { x.kind = Xform::RotateY; x.angle = need_num(j, "angle"); }
    else if (kind == "rotateZ") { x.kind = Xform::RotateZ; x.angle = need_num(j, "angle"); }
    else if (kind == "rotate") { x.kind = Xform::Rotate; x.v = need_vec(j, "axis"); x.angle = need_num(j, "angle"); }
    else if (kind == "sequence") {
        x.kind = Xform::Sequence;
        const JsonValue &a = need(j, "transforms");
        if (a.kind != JsonValue::Array) throw JsonError("'transforms' is not an array");
        for (auto &e : a.arr) x.seq.push_back(decode_xform(e));
    } else throw JsonError("Unknown type for transform " + kind);
    return x;
}
static D3 apply(const Xform &t, D3 p) {  // src/Transform.hs:16-24
    switch (t.kind) {
    case Xform::Translate: return add(p, t.v);
    case Xform::Scale: return {t.v.x * p.x, t.v.y * p.y, t.v.z * p.z};
    case Xform::Rotate: return mul(rot_axis(t.v, t.angle), p);
    case Xform::RotateX: return mul(rot_x(t.angle), p);
    case Xform::RotateY: return mul(rot_y(t.angle), p);
    case Xform::RotateZ: return mul(rot_z(t.angle), p);
    case Xform::Sequence:
        for (auto &s : t.seq) p = apply(s, p);
        return p;
    }
    return p;
}

// ------------------------------------------------------------------- meshes
struct Corner { D3 p, n; double u, v; };  // Vertex, src/Mesh.hs:32
struct Face { Corner c[3]; };             // Triangle, src/Mesh.hs:48

static bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }

// `read :: String -> Double` accepts [-]digits[.digits][e[+-]digits]; other
// spellings make the reference die with "Prelude.read: no parse" (src/Mesh.hs:132-133).
static double hs_read_double(const char *b, const char *e) {
    const char *p = b;
    auto digits = [&](const char *&q) { const char *s = q; while (q < e && std::isdigit((unsigned char)*q)) q++; return q > s; };
    if (p < e && *p == '-') p++;
    bool ok = digits(p);
    if (ok && p < e && *p == '.') { p++; ok = digits(p); }
    if (ok && p < e && (*p == 'e' || *p == 'E')) { p++; if (p < e && (*p == '+' || *p == '-')) p++; ok = digits(p); }
    if (!ok || p != e) throw SceneError(RAYHS_ERR_PARSE, "Prelude.read: no parse");
    return strtod(std::string(b, e).c_str(), nullptr);
}
// readMaybe :: String -> Maybe Int (src/Mesh.hs:156-157)
static bool hs_read_int(const char *b, const char *e, long &out) {
    const char *p = b;
    if (p < e && *p == '-') p++;
    if (p == e || e - b > 18) return false;
    for (const char *q = p; q < e; q++) if (!std::isdigit((unsigned char)*q)) return false;
    out = strtol(std::string(b, e).c_str(), nullptr, 10);
    return true;
}

struct ObjIndex { long p; bool has_t; long t; bool has_n; long n; };  // OBJIndex, src/Mesh.hs:125

// readOBJ + buildMesh (src/Mesh.hs:118-221).  flattenVertices (src/Mesh.hs:195-212)
// only de-duplicates index triples; a face corner's value is mkV of its triple,
// so corners are materialised directly (no O(n^2) map) with identical values.
static std::vector<Face> load_obj(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw SceneError(RAYHS_ERR_IO, path + ": openFile: does not exist (No such file or directory)");
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    std::vector<D3> pos, nrm;
    std::vector<std::pair<double, double>> tex;
    std::vector<ObjIndex> idx;
    std::vector<std::pair<const char *, const char *>> words;
    size_t at = 0;
    while (at < text.size()) {
        size_t nl = text.find('\n', at);
        if (nl == std::string::npos) nl = text.size();
        const char *b = text.data() + at, *e = text.data() + nl;
        at = nl + 1;
        int kind = -1;
        if (e - b >= 2 && b[0] == 'v' && b[1] == ' ') { kind = 0; b += 2; }
        else if (e - b >= 3 && b[0] == 'v' && b[1] == 'n' && b[2] == ' ') { kind = 1; b += 3; }
        else if (e - b >= 3 && b[0] == 'v' && b[1] == 't' && b[2] == ' ') { kind = 2; b += 3; }
        else if (e - b >= 2 && b[0] == 'f' && b[1] == ' ') { kind = 3; b += 2; }
        if (kind < 0) continue;  // readLine _ = Nothing
        words.clear();
        for (const char *p = b; p < e;) {
            while (p < e && is_space(*p)) p++;
            if (p >= e) break;
            const char *w = p;
            while (p < e && !is_space(*p)) p++;
            words.emplace_back(w, p);
        }
        if (kind <= 2) {
            size_t want = kind == 2 ? 2 : 3;
            if (words.size() != want) continue;  // `case length floats of 3 -> ...; _ -> Nothing`
            double d[3] = {0, 0, 0};
            for (size_t k = 0; k < want; k++) d[k] = hs_read_double(words[k].first, words[k].second);
            if (kind == 0) pos.push_back({d[0], d[1], d[2]});
            else if (kind == 1) nrm.push_back({d[0], d[1], d[2]});
            else tex.emplace_back(d[0], d[1]);
        } else {
            std::vector<ObjIndex> got;
            for (auto &w : words) {  // readOBJIndex: splitOn "/" then parseIndex
                const char *part[3] = {nullptr, nullptr, nullptr}, *pend[3] = {nullptr, nullptr, nullptr};
                int np = 0;
                const char *s = w.first;
                for (const char *p = w.first;; p++) {
                    if (p == w.second || *p == '/') {
                        if (np < 3) { part[np] = s; pend[np] = p; }
                        np++;
                        s = p + 1;
                        if (p == w.second) break;
                    }
                }
                ObjIndex oi{0, false, 0, false, 0};
                if (!hs_read_int(part[0], pend[0], oi.p)) continue;
                if (np >= 2) oi.has_t = hs_read_int(part[1], pend[1], oi.t);
                if (np >= 3) oi.has_n = hs_read_int(part[2], pend[2], oi.n);
                got.push_back(oi);
            }
            if (got.size() > 2) idx.insert(idx.end(), got.begin(), got.end());  // is ++ f: polygons are NOT fanned
        }
    }
    if (idx.size() % 3 != 0)
        throw SceneError(RAYHS_ERR_PARSE, path + ": face indices do not group into triangles (Non-exhaustive patterns in function faces)");
    std::vector<Face> faces(idx.size() / 3);
    for (size_t k = 0; k < idx.size(); k++) {
        const ObjIndex &oi = idx[k];
        Corner c{{0, 0, 0}, {0, 0, 0}, 0, 0};
        if (oi.p < 1 || (size_t)oi.p > pos.size()) throw SceneError(RAYHS_ERR_PARSE, path + ": Prelude.!!: index out of range (position)");
        c.p = pos[(size_t)oi.p - 1];
        if (oi.has_n) {
            if (oi.n < 1 || (size_t)oi.n > nrm.size()) throw SceneError(RAYHS_ERR_PARSE, path + ": Prelude.!!: index out of range (normal)");
            c.n = nrm[(size_t)oi.n - 1];
        }
        if (oi.has_t) {
            if (oi.t < 1 || (size_t)oi.t > tex.size()) throw SceneError(RAYHS_ERR_PARSE, path + ": Prelude.!!: index out of range (uv)");
            c.u = tex[(size_t)oi.t - 1].first;
            c.v = tex[(size_t)oi.t - 1].second;
        }
        faces[k / 3].c[k % 3] = c;
    }
    return faces;
}

// Bitmap.readPPM (src/Bitmap.hs:31-37): exactly three header lines, size on the
// second, then every whitespace-separated number / 255 in groups of three.
static void load_texture(const std::string &path, int &w, int &h, std::vector<double> &texels) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw SceneError(RAYHS_ERR_IO, path + ": openFile: does not exist (No such file or directory)");
    std::string l0, l1, l2;
    std::getline(f, l0);
    std::getline(f, l1);
    std::getline(f, l2);
    {
        std::istringstream s(l1);
        if (!(s >> w >> h) || w <= 0 || h <= 0) throw SceneError(RAYHS_ERR_PARSE, path + ": bad PPM size line");
    }
    std::vector<double> vals;
    std::string tok;
    while (f >> tok) vals.push_back(hs_read_double(tok.data(), tok.data() + tok.size()) / 255);
    size_t n = vals.size() / 3;
    if (n < (size_t)w * (size_t)h) throw SceneError(RAYHS_ERR_PARSE, path + ": PPM holds fewer pixels than its header says");
    texels.assign(vals.begin(), vals.begin() + (std::ptrdiff_t)(n * 3));
}

// --------------------------------------------------------------- tree build
// buildNode (src/KDTree.hs:79-90), emitted straight into the flat arrays:
// nodes in DFS pre-order, leaf triangles appended in DFS (left-to-right) order.
struct TreeBuilder {
    FlatScene &fs;
    int max_depth = -1;
    explicit TreeBuilder(FlatScene &f) : fs(f) {}

    int build(const std::vector<const Face *> &tris, int depth, int axis) {
        if (tris.empty()) return -1;  // Empty
        // buildBoundingBox: foldl includeTriangle empty (src/KDTree.hs:19-31)
        const double inf = std::numeric_limits<double>::infinity();
        D3 lo{inf, inf, inf}, hi{-inf, -inf, -inf};
        for (const Face *t : tris)
            for (int k = 0; k < 3; k++) {
                const D3 &p = t->c[k].p;
                lo = {hs_min(lo.x, p.x), hs_min(lo.y, p.y), hs_min(lo.z, p.z)};
                hi = {hs_max(hi.x, p.x), hs_max(hi.y, p.y), hs_max(hi.z, p.z)};
            }
        int id = (int)fs.node_left.size();
        fs.node_lo.insert(fs.node_lo.end(), {lo.x, lo.y, lo.z});
        fs.node_hi.insert(fs.node_hi.end(), {hi.x, hi.y, hi.z});
        fs.node_left.push_back(-1);
        fs.node_right.push_back(-1);
        fs.node_first_tri.push_back(0);
        fs.node_tri_count.push_back(-1);
        fs.node_axis.push_back(axis);
        max_depth = std::max(max_depth, depth);
        if (tris.size() < 20 || depth >= 100) {  // Leaf
            fs.node_first_tri[(size_t)id] = (int)(fs.tri_pos.size() / 9);
            fs.node_tri_count[(size_t)id] = (int)tris.size();
            for (const Face *t : tris) {
                for (int k = 0; k < 3; k++) {
                    const Corner &c = t->c[k];
                    fs.tri_pos.insert(fs.tri_pos.end(), {c.p.x, c.p.y, c.p.z});
                    fs.tri_nrm.insert(fs.tri_nrm.end(), {c.n.x, c.n.y, c.n.z});
                    fs.tri_uv.insert(fs.tri_uv.end(), {c.u, c.v});
                }
            }
            return id;
        }
        auto comp = [axis](const D3 &v) { return axis == 0 ? v.x : axis == 1 ? v.y : v.z; };
        double split = 0.5 * (comp(hi) + comp(lo));
        std::vector<const Face *> left, right;
        for (const Face *t : tris) {
            D3 bc = scl(1.0 / 3, add(add(t->c[0].p, t->c[1].p), t->c[2].p));  // baryCenter, src/KDTree.hs:71-74
            double c = comp(bc);
            if (c < split) left.push_back(t);
            if (split <= c) right.push_back(t);
        }
        int next = (axis + 1) % 3;
        int l = build(left, depth + 1, next);
        int r = build(right, depth + 1, next);
        fs.node_left[(size_t)id] = l;
        fs.node_right[(size_t)id] = r;
        return id;
    }
};

// ------------------------------------------------------------ scene assembly
static void push3(std::vector<double> &v, D3 a) { v.insert(v.end(), {a.x, a.y, a.z}); }

static void decode_material(const JsonValue &m, FlatScene &fs) {  // src/JSON.hs:40-61, MaterialDescriptors.hs:27-45
    const std::string &kind = need_str(m, "type");
    int mk;
    double ior = 0;
    D3 ce{0, 0, 0}, c1{0, 0, 0}, c2{0, 0, 0};
    double size = 0;
    int ck = RAYHS_CMAP_FLAT, tw = 0, th = 0;
    int64_t toff = 0;
    auto colormap = [&](const JsonValue &cd) {
        const std::string &ct = need_str(cd, "type");
        if (ct == "flat") { ck = RAYHS_CMAP_FLAT; c1 = need_rgb(cd, "color"); }
        else if (ct == "checker") { ck = RAYHS_CMAP_CHECKER; c1 = need_rgb(cd, "color1"); c2 = need_rgb(cd, "color2"); size = need_num(cd, "size"); }
        else if (ct == "texture") {
            ck = RAYHS_CMAP_TEXTURE;
            std::vector<double> px;
            load_texture(need_str(cd, "fileName"), tw, th, px);
            toff = (int64_t)(fs.texels.size() / 3);
            fs.texels.insert(fs.texels.end(), px.begin(), px.end());
        } else throw JsonError("Unknown type for color map " + ct);
    };
    if (kind == "mirror") { mk = RAYHS_MAT_MIRROR; ior = need_num(m, "ior"); }
    else if (kind == "diffuse") { mk = RAYHS_MAT_DIFFUSE; colormap(need(m, "cd")); }
    else if (kind == "plastic") { mk = RAYHS_MAT_PLASTIC; colormap(need(m, "cd")); ior = need_num(m, "ior"); }
    else if (kind == "emmit") { mk = RAYHS_MAT_EMMIT; ce = need_rgb(m, "ce"); }
    else if (kind == "transparent") { mk = RAYHS_MAT_TRANSPARENT; ior = need_num(m, "ior"); }
    else throw JsonError("Unknown type for material " + kind);
    fs.mat_kind.push_back(mk);
    fs.mat_ior.push_back(ior);
    push3(fs.mat_ce, ce);
    fs.cmap_kind.push_back(ck);
    push3(fs.cmap_c1, c1);
    push3(fs.cmap_c2, c2);
    fs.cmap_size.push_back(size);
    fs.tex_w.push_back(tw);
    fs.tex_h.push_back(th);
    fs.tex_offset.push_back(toff);
}

struct PendingMesh { std::string file; Xform xf; };

void FlatScene::refresh_view() {
    rayhs_scene_desc &d = view;
    std::memset(&d, 0, sizeof d);
    d.n_objects = (int32_t)obj_geom_kind.size();
    d.obj_geom_kind = obj_geom_kind.data();
    d.obj_geom_index = obj_geom_index.data();
    d.mat_kind = mat_kind.data();
    d.mat_ior = mat_ior.data();
    d.mat_ce = mat_ce.data();
    d.cmap_kind = cmap_kind.data();
    d.cmap_c1 = cmap_c1.data();
    d.cmap_c2 = cmap_c2.data();
    d.cmap_size = cmap_size.data();
    d.tex_w = tex_w.data();
    d.tex_h = tex_h.data();
    d.tex_offset = tex_offset.data();
    d.n_texels = (int64_t)(texels.size() / 3);
    d.texels = texels.data();
    d.n_planes = (int32_t)(plane_point.size() / 3);
    d.plane_point = plane_point.data();
    d.plane_normal = plane_normal.data();
    d.plane_tangent = plane_tangent.data();
    d.n_spheres = (int32_t)sphere_radius.size();
    d.sphere_center = sphere_center.data();
    d.sphere_radius = sphere_radius.data();
    d.n_meshes = (int32_t)mesh_root.size();
    d.mesh_root = mesh_root.data();
    d.n_nodes = (int32_t)node_left.size();
    d.node_lo = node_lo.data();
    d.node_hi = node_hi.data();
    d.node_left = node_left.data();
    d.node_right = node_right.data();
    d.node_first_tri = node_first_tri.data();
    d.node_tri_count = node_tri_count.data();
    d.node_axis = node_axis.data();
    d.n_tris = (int32_t)(tri_pos.size() / 9);
    d.tri_pos = tri_pos.data();
    d.tri_nrm = tri_nrm.data();
    d.tri_uv = tri_uv.data();
    d.n_lights = (int32_t)light_kind.size();
    d.light_kind = light_kind.data();
    d.light_vec = light_vec.data();
    d.light_color = light_color.data();
    d.light_radius = light_radius.data();
}

// RenderDesc (src/JSON.hs:135-141) -> built + flattened scene (src/Descriptors.hs:39-55)
std::unique_ptr<RenderDescData> decode_render_desc(const std::string &json_text) {
    JsonValue root;
    try {
        root = JsonReader(json_text).parse_document();
    } catch (const JsonError &e) {
        throw SceneError(RAYHS_ERR_PARSE, std::string("Failed to read scene (") + e.what() + ")");
    }
    auto rd = std::make_unique<RenderDescData>();
    FlatScene &fs = rd->flat;
    std::vector<PendingMesh> meshes;
    try {
        // field order of the RenderDesc parser: scene, camera, width, height, maxDepth
        const JsonValue &scene = need(root, "scene");
        const JsonValue &objects = need(scene, "objects");
        if (objects.kind != JsonValue::Array) throw JsonError("'objects' is not an array");
        for (const JsonValue &o : objects.arr) {
            const JsonValue &g = need(o, "geometry");
            const std::string &kind = need_str(g, "type");
            if (kind == "sphere") {  // src/JSON.hs:112-114
                fs.obj_geom_kind.push_back(RAYHS_GEOM_SPHERE);
                fs.obj_geom_index.push_back((int)fs.sphere_radius.size());
                push3(fs.sphere_center, need_vec(g, "center"));
                fs.sphere_radius.push_back(need_num(g, "radius"));
            } else if (kind == "plane") {  // src/JSON.hs:115-118 + Descriptors.hs:51: net effect point/normal/tangent
                fs.obj_geom_kind.push_back(RAYHS_GEOM_PLANE);
                fs.obj_geom_index.push_back((int)(fs.plane_point.size() / 3));
                push3(fs.plane_point, need_vec(g, "point"));
                push3(fs.plane_normal, need_vec(g, "normal"));
                push3(fs.plane_tangent, need_vec(g, "tangent"));
            } else if (kind == "mesh") {  // src/JSON.hs:119-121
                fs.obj_geom_kind.push_back(RAYHS_GEOM_MESH);
                fs.obj_geom_index.push_back((int)meshes.size());
                meshes.push_back({need_str(g, "fileName"), decode_xform(need(g, "transform"))});
            } else throw JsonError("unknown geometry type " + kind);  // no fallback case in the reference
            decode_material(need(o, "material"), fs);
        }
        const JsonValue &lights = need(scene, "lights");
        if (lights.kind != JsonValue::Array) throw JsonError("'lights' is not an array");
        for (const JsonValue &l : lights.arr) {  // src/JSON.hs:63-72
            const std::string &kind = need_str(l, "type");
            if (kind == "directional") {
                fs.light_kind.push_back(RAYHS_LIGHT_DIRECTIONAL);
                push3(fs.light_vec, need_vec(l, "direction"));
                push3(fs.light_color, need_rgb(l, "color"));
                fs.light_radius.push_back(0);
            } else if (kind == "point") {
                fs.light_kind.push_back(RAYHS_LIGHT_POINT);
                push3(fs.light_vec, need_vec(l, "position"));
                push3(fs.light_color, need_rgb(l, "color"));
                fs.light_radius.push_back(need_num(l, "radius"));
            } else throw JsonError("Unknown type for light " + kind);
        }
        const JsonValue &cam = need(root, "camera");  // src/JSON.hs:99-106
        rayhs_camera &c = rd->camera;
        std::memset(&c, 0, sizeof c);
        D3 p = need_vec(cam, "position"), t = need_vec(cam, "target"), u = need_vec(cam, "up");
        c.position[0] = p.x; c.position[1] = p.y; c.position[2] = p.z;
        c.target[0] = t.x; c.target[1] = t.y; c.target[2] = t.z;
        c.up[0] = u.x; c.up[1] = u.y; c.up[2] = u.z;
        const JsonValue &proj = need(cam, "projection");  // src/JSON.hs:74-84
        const std::string &pk = need_str(proj, "type");
        if (pk == "orthographic") {
            c.proj_kind = RAYHS_PROJ_ORTHOGRAPHIC;
            c.proj_width = need_num(proj, "width");
            c.proj_height = need_num(proj, "height");
        } else if (pk == "perspective") {
            c.proj_kind = RAYHS_PROJ_PERSPECTIVE;
            c.fovy = need_num(proj, "fovy");
            c.proj_width = need_num(proj, "width");
            c.proj_height = need_num(proj, "height");
            c.near_ = need_num(proj, "near");
        } else throw JsonError("Unknown type for projection " + pk);
        rd->width = need_int(root, "width");
        rd->height = need_int(root, "height");
        rd->max_depth = need_int(root, "maxDepth");
    } catch (const JsonError &e) {
        throw SceneError(RAYHS_ERR_PARSE, std::string("Failed to read scene (") + e.what() + ")");
    }
    // buildScene: readOBJ -> Mesh.transform -> buildKDTree, per mesh (src/Descriptors.hs:53-55)
    TreeBuilder tb(fs);
    for (const PendingMesh &pm : meshes) {
        std::vector<Face> faces = load_obj(pm.file);
        // Mesh.transform (src/Mesh.hs:89-101): a bare Translate leaves normals alone;
        // any other transform is applied to normals as if they were points.
        const bool bare_translate = pm.xf.kind == Xform::Translate;
        for (Face &f : faces)
            for (Corner &c : f.c) {
                c.p = apply(pm.xf, c.p);
                if (!bare_translate) c.n = apply(pm.xf, c.n);
            }
        std::vector<const Face *> all(faces.size());
        for (size_t i = 0; i < faces.size(); i++) all[i] = &faces[i];
        fs.mesh_root.push_back(tb.build(all, 0, 0));
    }
    rd->tree_depth = tb.max_depth;
    fs.refresh_view();
    return rd;
}

std::unique_ptr<RenderDescData> load_render_desc(const std::string &path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw SceneError(RAYHS_ERR_IO, path + ": openBinaryFile: does not exist (No such file or directory)");
    std::stringstream ss;
    ss << f.rdbuf();
    return decode_render_desc(ss.str());
}

// ------------------------------------------------------------------- output
// toIntC (src/Image.hs:54-55): truncate (255 * min c 1); NaN -> 255.
int to_int_c(double c) {
    double v = 255 * hs_min(c, 1);
    if (v < -32768.0) return -32768;
    return (int)v;
}

// writePPM / formatPixelsPPM (src/Image.hs:60-75): "P3\nW H\n255\n", rows joined by '\n', every pixel
// "r g b" followed by two spaces, no trailing newline.  A 1080p frame is ~20 MB of ASCII (4K: ~80 MB): the
// digits come from a 256-entry table (toIntC yields 0..255 unless a colour is negative), bands of rows are
// formatted by a few threads into their own buffers, and the bands are written in order.
static inline char *put_int(char *p, int v) {
    static const struct Table {
        char txt[256][4];
        unsigned char len[256];
        Table() {
            for (int k = 0; k < 256; k++) {
                int n = 0;
                if (k >= 100) txt[k][n++] = (char)('0' + k / 100);
                if (k >= 10) txt[k][n++] = (char)('0' + (k / 10) % 10);
                txt[k][n++] = (char)('0' + k % 10);
                len[k] = (unsigned char)n;
                for (; n < 4; n++) txt[k][n] = ' ';
            }
        }
    } table;
    if ((unsigned)v < 256u) {
        std::memcpy(p, table.txt[v], 4);  // the bytes after the digits are overwritten by the caller
        return p + table.len[v];
    }
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    if (v < 0) *p++ = '-';
    char tmp[12];
    int n = 0;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) *p++ = tmp[--n];
    return p;
}
template <class Get>
static void write_ppm_impl(const std::string &path, int w, int h, Get get) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) throw SceneError(RAYHS_ERR_IO, path + ": openFile: permission denied or missing directory");
    const std::string head = "P3\n" + std::to_string(w) + " " + std::to_string(h) + "\n255\n";
    const size_t row_worst = (size_t)w * 23 + 8;  // <= 3 x "-32768" + separators per pixel, + newline and table slack
    const int hw = (int)std::thread::hardware_concurrency();
    const int n_bands = std::max(1, std::min(std::min(hw > 0 ? hw : 1, 16), h / 32));
    std::vector<std::vector<char>> bands((size_t)n_bands);
    auto format_band = [&](int b) {
        const int y0 = (int)((long long)h * b / n_bands), y1 = (int)((long long)h * (b + 1) / n_bands);
        std::vector<char> &buf = bands[(size_t)b];
        buf.resize((size_t)(y1 - y0) * ((size_t)w * 13 + 1) + row_worst);  // typical size; grown below if a row needs more
        size_t used = 0;
        for (int y = y0; y < y1; y++) {
            if (buf.size() - used < row_worst) buf.resize(buf.size() * 2 + row_worst);
            char *p = buf.data() + used;
            if (y) *p++ = '\n';
            for (int x = 0; x < w; x++) {
                size_t i = ((size_t)y * (size_t)w + (size_t)x) * 3;
                p = put_int(p, get(i)); *p++ = ' ';
                p = put_int(p, get(i + 1)); *p++ = ' ';
                p = put_int(p, get(i + 2)); *p++ = ' '; *p++ = ' ';
            }
            used = (size_t)(p - buf.data());
        }
        buf.resize(used);
    };
    std::vector<std::thread> workers;
    for (int b = 1; b < n_bands; b++) workers.emplace_back(format_band, b);
    format_band(0);
    for (std::thread &t : workers) t.join();
    bool ok = std::fwrite(head.data(), 1, head.size(), f) == head.size();
    for (const std::vector<char> &buf : bands) ok = ok && (buf.empty() || std::fwrite(buf.data(), 1, buf.size(), f) == buf.size());
    if (std::fclose(f) != 0 || !ok) throw SceneError(RAYHS_ERR_IO, path + ": write failed");
}
void write_ppm_i16(const std::string &path, int w, int h, const int16_t *q) {
    write_ppm_impl(path, w, h, [q](size_t i) { return (int)q[i]; });
}
void write_ppm_f64(const std::string &path, int w, int h, const double *rgb) {
    write_ppm_impl(path, w, h, [rgb](size_t i) { return to_int_c(rgb[i]); });
}

}  // namespace rayhs
