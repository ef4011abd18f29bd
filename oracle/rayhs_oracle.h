/*
 * rayhs_oracle.h -- C interface of the CPU parity oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a literal fp64 restatement of the
 * reference's per-pixel trace loop (oilandrust/rayhs, Haskell).  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product (rayhs_amd/, librayhs_hip.so, the rayhs CLI) never does.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, no known-answer
 * tests and no expected images, and GHC is not available
 * to run it.  The oracle is pinned only by analytic identities derived from the
 * reference source (tests/test_oracle.py) and by the tree statistics table of
 * SURVEY.md section 8.
 */
#ifndef RAYHS_ORACLE_H
#define RAYHS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Material.hs:13-19 */
enum { ORC_MAT_MIRROR = 0, ORC_MAT_DIFFUSE = 1, ORC_MAT_PLASTIC = 2, ORC_MAT_EMMIT = 3,
       ORC_MAT_TRANSPARENT = 4, ORC_MAT_SHOWNORMAL = 5, ORC_MAT_SHOWUV = 6 };
/* ColorMap.hs:11-16 */
enum { ORC_CMAP_FLAT = 0, ORC_CMAP_CHECKER = 1, ORC_CMAP_TEXTURE = 2 };
/* Transform.hs:7-13 (Sequence is flattened by the caller; order preserved) */
enum { ORC_XF_TRANSLATE = 0, ORC_XF_SCALE = 1, ORC_XF_ROTATE = 2, ORC_XF_ROTATEX = 3,
       ORC_XF_ROTATEY = 4, ORC_XF_ROTATEZ = 5 };
/* Projection.hs:10-15 */
enum { ORC_PROJ_ORTHOGRAPHIC = 0, ORC_PROJ_PERSPECTIVE = 1 };

typedef struct {
    int kind;          /* ORC_MAT_* */
    double ior;        /* mirror, plastic, transparent */
    double ce[3];      /* emmit */
    int cmap_kind;     /* ORC_CMAP_* (diffuse, plastic) */
    double c1[3];      /* flat colour / checker color1 */
    double c2[3];      /* checker color2 */
    double size;       /* checker size */
    int tex_w, tex_h;  /* texture */
    const double *tex; /* tex_w*tex_h*3 doubles, row-major, copied */
} orc_material;

typedef struct {
    int kind;      /* ORC_XF_* */
    double v[3];   /* translate/scale vector, rotate axis */
    double angle;  /* rotate* angle */
} orc_xform;

typedef struct {
    double pos[3], target[3], up[3];
    int proj_kind;            /* ORC_PROJ_* */
    double fovy, pw, ph, near_; /* pw/ph/near are parsed but unused (Projection.hs:34) */
} orc_camera;

/* One "ray" = one scene query. Shadow queries stop at the first
 * occluding object in scene order, as lazy evaluation does in the reference
 * (RayHs.hs:74-80, 94-96); inside an object traversal is exhaustive. */
typedef struct {
    uint64_t rays_primary, rays_reflect, rays_refract, rays_refract_exit, rays_shadow;
    uint64_t nodes_tested;   /* rayInterBox calls (KDTree.hs:39) */
    uint64_t tris_tested;    /* triangleIntersection calls (Mesh.hs:59) */
    uint64_t prims_tested;   /* plane/sphere tests (Geometry.hs:68) */
    uint64_t shaded_hits;    /* irradiance calls (RayHs.hs:107) */
} orc_stats;

typedef struct {
    int hit;        /* 0 = miss */
    int object;     /* scene object index */
    double t;
    double p[3], n[3], uv[2];
} orc_hit;

typedef struct {
    int tris, inner, leaves, empties, max_depth, leaf_min, leaf_max;
    double leaf_mean;
} orc_tree_stats;

typedef struct orc_scene orc_scene;

orc_scene *orc_scene_new(void);
void orc_scene_free(orc_scene *s);
const char *orc_last_error(void);

int orc_add_plane(orc_scene *s, const double point[3], const double normal[3],
                  const double tangent[3], const orc_material *m);
int orc_add_sphere(orc_scene *s, const double center[3], double radius, const orc_material *m);
/* top_is_translate: the JSON transform is a bare Translate (Mesh.hs:89-95). */
int orc_add_mesh_obj(orc_scene *s, const char *path, int top_is_translate, int n_xf,
                     const orc_xform *xf, const orc_material *m);
int orc_add_light_directional(orc_scene *s, const double dir[3], const double color[3]);
int orc_add_light_point(orc_scene *s, const double pos[3], const double color[3], double radius);

int orc_num_objects(const orc_scene *s);
int orc_mesh_tree_stats(const orc_scene *s, int object, orc_tree_stats *out);
/* number of triangles of a mesh object and a copy of them in file order:
 * 24 doubles per triangle = 3 x (pos[3], normal[3], uv[2]) after the transform. */
int orc_mesh_triangles(const orc_scene *s, int object, double *out, int max_tris);

/* rgb_out: (row1-row0)*w*3 doubles, row-major, row0 first.  spp==1 follows the
 * live path rayTrace (RayHs.hs:161-166); spp>1 follows distributedTracePixel
 * (RayHs.hs:173-188) with the build-defined counter sampler orc_sample01. */
int orc_render(const orc_scene *s, const orc_camera *cam, int w, int h, int max_depth,
               int spp, uint64_t seed, int row0, int row1, double *rgb_out,
               orc_stats *stats_or_null, int nthreads);

/* 1: shadow queries test every object (work-counter comparison with the
 * kernel's exhaustive mode); 0: stop at the first occluder like the reference. */
void orc_set_shadow_exhaustive(int on);

/* probes for unit tests */
void orc_ray_from_pixel(const orc_camera *cam, double w, double h, double px, double py,
                        double origin[3], double dir[3]);
void orc_closest_hit(const orc_scene *s, const double origin[3], const double dir[3], orc_hit *out);
/* same query but meshes are intersected by brute force (Mesh.hs:44-45) */
void orc_closest_hit_bruteforce(const orc_scene *s, const double origin[3], const double dir[3],
                                orc_hit *out);
void orc_trace_ray(const orc_scene *s, const double origin[3], const double dir[3], int depth,
                   int max_depth, double rgb[3]);
double orc_r0(double n1, double n2);
double orc_fresnel(double ior, double cos0);
/* 0 (default): x^5 = x*x*x*x*x (GHC -O RULES "^5/Int"); 1: ((x*x)*(x*x))*x (GHC.Real's (^) un-rewritten).  Process-wide. */
void orc_set_pow5_variant(int square_and_multiply);
double orc_mod_floor(double n, double d);
int orc_refract(const double i[3], const double n[3], double n1, double n2, double out[3]);
void orc_color_at(const orc_material *m, double u, double v, double rgb[3]);
double orc_sample01(uint64_t seed, uint64_t pixel, uint32_t sample, uint32_t dim, uint32_t spp);
int orc_to_int_c(double c);
/* writes the reference's P3 format (Image.hs:60-75) from doubles */
int orc_write_ppm(const char *path, int w, int h, const double *rgb);

#ifdef __cplusplus
}
#endif
#endif
