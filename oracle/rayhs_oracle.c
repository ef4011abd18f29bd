/*
 * rayhs_oracle.c -- CPU parity oracle for the rayhs per-pixel trace loop.
 *
 * TEST INFRASTRUCTURE ONLY (see rayhs_oracle.h).  PARITY UNPINNED: the
 * reference has no golden vectors and cannot be run here (no GHC).
 *
 * Plain C11, fp64 everywhere, compiled with -ffp-contract=off because GHC does
 * not fuse multiply-add, and with -fno-builtin-sin -fno-builtin-cos because GHC
 * calls libm's sin and cos each by itself (gcc -O2 merges the pair of a rotation
 * matrix into sincos(), whose results are not always theirs: oracle/Makefile).
 * Every function cites the reference file:line it
 * restates (paths relative to /root/reference).  Operation order follows the
 * Haskell source literally: left-associated sums, Haskell min/max semantics,
 * minimumBy's first-wins tie rule, GHC's (^) squaring chain.
 */
#include "rayhs_oracle.h"

#include <ctype.h>
#include <errno.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ errors */
static _Thread_local char g_err[512];
static int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return -1;
}
const char *orc_last_error(void) { return g_err; }

/* ------------------------------------------------------------------ Vec.hs */
typedef struct { double x, y, z; } vec;   /* Vec.hs:29-31 */
typedef struct { double u, v; } uv2;      /* Vec.hs:33-34 */
typedef vec color;                        /* Color.hs:19 (r,g,b) */

static inline vec V(double x, double y, double z) { vec r = {x, y, z}; return r; }
static inline vec vadd(vec a, vec b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }   /* Vec.hs:38 */
static inline vec vsub(vec a, vec b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }   /* Vec.hs:42 */
static inline vec vmulv(vec a, vec b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }  /* Vec.hs:40, Color.hs:25 */
static inline vec vneg(vec a) { return V(-a.x, -a.y, -a.z); }                         /* Vec.hs:44 */
static inline vec vmul(double l, vec a) { return V(l * a.x, l * a.y, l * a.z); }      /* Vec.hs:70 */
static inline double vdot(vec a, vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; } /* Vec.hs:105 */
static inline vec vcross(vec a, vec b) {                                              /* Vec.hs:109-110 */
    return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline double vsqrlen(vec a) { return vdot(a, a); }                            /* Vec.hs:114 */
static inline double vsqrdist(vec a, vec b) { return vsqrlen(vsub(a, b)); }           /* Vec.hs:118 */
static inline double vdist(vec a, vec b) { return sqrt(vsqrdist(a, b)); }             /* Vec.hs:122 */
static inline vec vnormalize(vec a) { return vmul(1 / sqrt(vsqrlen(a)), a); }         /* Vec.hs:126 */
static inline vec vreflect(vec v, vec n) { return vsub(v, vmul(2 * vdot(v, n), n)); } /* Vec.hs:130 */
static inline double vidx(vec a, int i) { return i == 0 ? a.x : i == 1 ? a.y : i == 2 ? a.z : 0; } /* Vec.hs:57-61 */

/* Haskell Ord Double defaults: min x y = if x <= y then x else y;
 * max x y = if x <= y then y else x (NaN behaviour differs from fmin/fmax). */
static inline double hmin(double x, double y) { return x <= y ? x : y; }
static inline double hmax(double x, double y) { return x <= y ? y : x; }

/* Vec.hs:132-140 */
static int vrefract(vec i, vec n, double n1, double n2, vec *out) {
    double n1n2 = n1 / n2;
    double cos0 = -(vdot(i, n));
    double sin20 = n1n2 * n1n2 * (1 - cos0 * cos0);
    if (sin20 > 1) return 0;
    double coeff = n1n2 * cos0 - sqrt(1.0 - sin20);
    *out = vadd(vmul(n1n2, i), vmul(coeff, n));
    return 1;
}
int orc_refract(const double i[3], const double n[3], double n1, double n2, double out[3]) {
    vec r;
    if (!vrefract(V(i[0], i[1], i[2]), V(n[0], n[1], n[2]), n1, n2, &r)) return 0;
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
    return 1;
}

static inline vec vminv(vec a, vec p) { return V(hmin(a.x, p.x), hmin(a.y, p.y), hmin(a.z, p.z)); } /* Vec.hs:144 */
static inline vec vmaxv(vec a, vec p) { return V(hmax(a.x, p.x), hmax(a.y, p.y), hmax(a.z, p.z)); } /* Vec.hs:148 */

/* Vec.hs:151-158 */
static void orthonormal(vec r, vec *s, vec *t) {
    if (fabs(r.x) < fabs(r.y) && fabs(r.x) < fabs(r.z)) *s = vnormalize(V(0, -(r.z), r.y));
    else if (fabs(r.y) < fabs(r.x) && fabs(r.y) < fabs(r.z)) *s = vnormalize(V(-(r.z), 0, r.x));
    else *s = vnormalize(V(-(r.y), r.x, 0));
    *t = vcross(r, *s);
}

/* ------------------------------------------------------------------ Mat.hs */
typedef struct { double a, b, c, d, e, f, g, h, i; } mat3; /* Mat.hs:15-18 row-major */

static vec mapply(mat3 m, vec v) { /* Mat.hs:40-44 */
    return V(m.a * v.x + m.b * v.y + m.c * v.z, m.d * v.x + m.e * v.y + m.f * v.z,
             m.g * v.x + m.h * v.y + m.i * v.z);
}
static mat3 mmul(mat3 p, mat3 q) { /* Mat.hs:25-31 */
    mat3 r = {p.a * q.a + p.b * q.d + p.c * q.g, p.a * q.b + p.b * q.e + p.c * q.h, p.a * q.c + p.b * q.f + p.c * q.i,
              p.d * q.a + p.e * q.d + p.f * q.g, p.d * q.b + p.e * q.e + p.f * q.h, p.d * q.c + p.e * q.f + p.f * q.i,
              p.g * q.a + p.h * q.d + p.i * q.g, p.g * q.b + p.h * q.e + p.i * q.h, p.g * q.c + p.h * q.f + p.i * q.i};
    return r;
}
static mat3 mtranspose(mat3 m) { mat3 r = {m.a, m.d, m.g, m.b, m.e, m.h, m.c, m.f, m.i}; return r; } /* Mat.hs:47-49 */
static mat3 from_columns(vec v1, vec v2, vec v3) { /* Mat.hs:83-87 */
    mat3 r = {v1.x, v2.x, v3.x, v1.y, v2.y, v3.y, v1.z, v2.z, v3.z};
    return r;
}
static mat3 rotate_z(double a) { mat3 r = {cos(a), -(sin(a)), 0, sin(a), cos(a), 0, 0, 0, 1}; return r; } /* Mat.hs:58-62 */
static mat3 rotate_y(double a) { mat3 r = {cos(a), 0, sin(a), 0, 1, 0, -(sin(a)), 0, cos(a)}; return r; } /* Mat.hs:64-68 */
static mat3 rotate_x(double a) { mat3 r = {1, 0, 0, 0, cos(a), -(sin(a)), 0, sin(a), cos(a)}; return r; } /* Mat.hs:70-74 */
static mat3 rotate_axis(vec axis, double angle) { /* Mat.hs:76-81: mt * rot * m */
    vec s, t;
    orthonormal(axis, &s, &t);
    mat3 m = from_columns(axis, s, t);
    mat3 mt = mtranspose(m);
    mat3 rot = rotate_x(angle);
    return mmul(mmul(mt, rot), m);
}
static mat3 look_at(vec pos, vec target, vec tup) { /* Mat.hs:89-93 */
    vec forward = vnormalize(vsub(target, pos));
    vec right = vnormalize(vcross(tup, forward));
    vec up = vcross(forward, right);
    return from_columns(right, up, forward);
}

/* ------------------------------------------------------------ Transform.hs */
static vec xf_apply(const orc_xform *t, vec p) { /* Transform.hs:16-23 */
    vec v = V(t->v[0], t->v[1], t->v[2]);
    switch (t->kind) {
    case ORC_XF_TRANSLATE: return vadd(p, v);
    case ORC_XF_SCALE: return V(v.x * p.x, v.y * p.y, v.z * p.z);
    case ORC_XF_ROTATE: return mapply(rotate_axis(v, t->angle), p);
    case ORC_XF_ROTATEX: return mapply(rotate_x(t->angle), p);
    case ORC_XF_ROTATEY: return mapply(rotate_y(t->angle), p);
    case ORC_XF_ROTATEZ: return mapply(rotate_z(t->angle), p);
    }
    return p;
}
static vec xf_seq(int n, const orc_xform *xf, vec p) { /* Transform.hs:24 foldl (flip transform) */
    for (int i = 0; i < n; i++) p = xf_apply(&xf[i], p);
    return p;
}

/* ----------------------------------------------------------- Geometry.hs */
typedef struct { vec o, d; } ray;                  /* Geometry.hs:24-25 */
typedef struct { vec p, n; uv2 uv; double t; } hit; /* Geometry.hs:39 */
static const double EPS = 0.000001;                /* Geometry.hs:31-32 */
static const double PI_ = 3.141592653589793;       /* Haskell pi :: Double */
#define PI_INV (1 / PI_)                           /* Math.hs:12-13 */

static inline vec ray_at(ray r, double t) { return vadd(r.o, vmul(t, r.d)); }            /* Geometry.hs:27-29 */
static inline ray ray_eps(vec p, vec n) { ray r = {vadd(p, vmul(EPS, n)), n}; return r; } /* Geometry.hs:34-36 */

/* Mesh.hs:32,48 */
typedef struct { vec p, n; uv2 uv; } vertex;
typedef struct { vertex a, b, c; } triangle;

/* KDTree.hs:13-14,59-61 */
typedef struct { vec lo, hi; } box;
enum { KD_EMPTY, KD_LEAF, KD_NODE };
typedef struct kdtree {
    int kind;
    box b;
    struct kdtree *l, *r;
    triangle *tris;
    int ntris;
} kdtree;

enum { GEOM_PLANE, GEOM_SPHERE, GEOM_MESH };
typedef struct {
    int kind;
    vec p, n, t;   /* plane point/normal/tangent, sphere centre in p */
    double radius;
    kdtree *tree;  /* mesh */
    triangle *tris;
    int ntris;
    orc_material mat;
    double *tex;
} object;

enum { LIGHT_DIRECTIONAL, LIGHT_POINT };
typedef struct { int kind; vec v; color c; double radius; } light; /* Light.hs:8-10 */

struct orc_scene {
    object *objs; int nobjs, cobjs;
    light *lights; int nlights, clights;
};

typedef struct { orc_stats st; } ctx;

/* Geometry.hs:68-79 */
static int inter_plane(ray r, const object *o, hit *h) {
    double dDotn = vdot(r.d, o->n);
    double time = vdot(o->n, vsub(o->p, r.o)) / dDotn;
    if (fabs(dDotn) > 0 && time > 0) {
        vec pos = ray_at(r, time);
        vec b = vcross(o->t, o->n);
        vec rel = vsub(pos, o->p);
        h->p = pos; h->n = o->n; h->uv.u = vdot(o->t, rel); h->uv.v = vdot(b, rel); h->t = time;
        return 1;
    }
    return 0;
}
/* Geometry.hs:96 */
static uv2 polar(vec p) { uv2 r = {PI_INV * atan(p.z / p.x), PI_INV * acos(p.y)}; return r; }
/* Geometry.hs:81-95 */
static int inter_sphere(ray r, const object *o, hit *h) {
    vec oc = vsub(r.o, o->p);
    double a = vdot(r.d, r.d);
    double b = 2.0 * vdot(r.d, oc);
    double c = vsqrlen(oc) - o->radius * o->radius;
    double delta = b * b - 4.0 * a * c;
    if (delta < 0.0) return 0;
    double t0 = 0.5 * ((-b) - sqrt(delta)) / a;
    if (t0 > 0) {
        vec p0 = ray_at(r, t0);
        vec n0 = vnormalize(vsub(p0, o->p));
        h->p = p0; h->n = n0; h->uv = polar(n0); h->t = t0;
        return 1;
    }
    double t1 = 0.5 * ((-b) + sqrt(delta)) / a;
    if (t1 > 0) {
        vec p1 = ray_at(r, t1);
        vec n1 = vnormalize(vsub(p1, o->p));
        h->p = p1; h->n = n1; h->uv = polar(n1); h->t = t1;
        return 1;
    }
    return 0;
}

/* Mesh.hs:59-82 (Moller-Trumbore), barycentricInterp Mesh.hs:53-57 */
static int inter_triangle(ray r, const triangle *tr, hit *h) {
    vec e1 = vsub(tr->b.p, tr->a.p);
    vec e2 = vsub(tr->c.p, tr->a.p);
    vec p = vcross(r.d, e2);
    double det = vdot(e1, p);
    double idet = 1 / det;
    vec t0 = vsub(r.o, tr->a.p);
    double u = idet * vdot(t0, p);
    vec q = vcross(t0, e1);
    double v = idet * vdot(r.d, q);
    double t = idet * vdot(e2, q);
    if (fabs(det) < EPS || u < 0 || u > 1 || v < 0 || (u + v) > 1 || t < EPS) return 0;
    double w = 1 - u - v;
    h->p = ray_at(r, t);
    h->n = vadd(vadd(vmul(u, tr->b.n), vmul(v, tr->c.n)), vmul(w, tr->a.n));
    h->uv.u = u * tr->b.uv.u + v * tr->c.uv.u + w * tr->a.uv.u;
    h->uv.v = u * tr->b.uv.v + v * tr->c.uv.v + w * tr->a.uv.v;
    h->t = t;
    return 1;
}

/* minimumBy (compare `on` t): foldl1 keeping x unless compare x y == GT, where
 * compare x y = LT if x<y, EQ if x==y, else GT (so a NaN on either side picks y).
 * Geometry.hs:54-57, RayHs.hs:67-71. */
static inline int keep_first(double tx, double ty) { return tx < ty || tx == ty; }

/* Geometry.hs:54-57 + Mesh.hs:44-45 / KDTree.hs:98-100 */
static int closest_in_list(ctx *cx, ray r, const triangle *tris, int n, hit *out) {
    int have = 0;
    hit best, h;
    for (int i = 0; i < n; i++) {
        cx->st.tris_tested++;
        if (inter_triangle(r, &tris[i], &h)) {
            if (!have) { best = h; have = 1; }
            else if (!keep_first(best.t, h.t)) best = h;
        }
    }
    if (have) *out = best;
    return have;
}

/* KDTree.hs:39-56 */
static int ray_inter_box(ray r, box b) {
    double idx = 1 / r.d.x, idy = 1 / r.d.y, idz = 1 / r.d.z;
    double t1 = idx * (b.lo.x - r.o.x), t2 = idx * (b.hi.x - r.o.x);
    double t3 = idy * (b.lo.y - r.o.y), t4 = idy * (b.hi.y - r.o.y);
    double t5 = idz * (b.lo.z - r.o.z), t6 = idz * (b.hi.z - r.o.z);
    double tmin = hmax(hmax(hmin(t1, t2), hmin(t3, t4)), hmin(t5, t6));
    double tmax = hmin(hmin(hmax(t1, t2), hmax(t3, t4)), hmax(t5, t6));
    return !(tmax < 0 || tmin > tmax);
}

/* KDTree.hs:96-115 */
static int ray_inter_tree(ctx *cx, ray r, const kdtree *k, hit *out) {
    if (k->kind == KD_EMPTY) return 0;
    cx->st.nodes_tested++;
    if (!ray_inter_box(r, k->b)) return 0;
    if (k->kind == KD_LEAF) return closest_in_list(cx, r, k->tris, k->ntris, out);
    hit lh, rh;
    int hl = ray_inter_tree(cx, r, k->l, &lh);
    int hr = ray_inter_tree(cx, r, k->r, &rh);
    if (!hl && !hr) return 0;
    if (hl && !hr) { *out = lh; return 1; }
    if (!hl && hr) { *out = rh; return 1; }
    *out = (lh.t < rh.t) ? lh : rh;  /* minMaybeHit: right wins ties */
    return 1;
}

static int inter_object(ctx *cx, ray r, const object *o, hit *h, int brute) {
    switch (o->kind) {
    case GEOM_PLANE: cx->st.prims_tested++; return inter_plane(r, o, h);
    case GEOM_SPHERE: cx->st.prims_tested++; return inter_sphere(r, o, h);
    default:
        if (brute) return closest_in_list(cx, r, o->tris, o->ntris, h);
        return ray_inter_tree(cx, r, o->tree, h);
    }
}

/* RayHs.hs:58-71 */
static int closest_intersection(ctx *cx, const orc_scene *s, ray r, hit *out, int *obj, int brute) {
    int have = 0;
    hit best, h;
    int bi = -1;
    for (int i = 0; i < s->nobjs; i++) {
        if (inter_object(cx, r, &s->objs[i], &h, brute)) {
            if (!have) { best = h; bi = i; have = 1; }
            else if (!keep_first(best.t, h.t)) { best = h; bi = i; }
        }
    }
    if (have) { *out = best; *obj = bi; }
    return have;
}

/* Counting mode for shadow queries.  0 (default): lazy, as the reference runs
 * -- stop at the first occluding object in scene order.  1: evaluate every
 * object (same answer, more work counted); matches the HIP kernel's
 * RAYHS_FLAG_EXHAUSTIVE run so that work counters can be compared exactly. */
static int g_shadow_exhaustive = 0;
void orc_set_shadow_exhaustive(int on) { g_shadow_exhaustive = on; }

/* RayHs.hs:74-87: does any object yield an occluder? Lazy evaluation in the
 * reference stops at the first one (only Just/Nothing is inspected, :94-96). */
static int shadow_intersection(ctx *cx, const orc_scene *s, const light *l, ray r) {
    hit h;
    int any = 0;
    for (int i = 0; i < s->nobjs; i++) {
        if (!inter_object(cx, r, &s->objs[i], &h, 0)) continue;
        int in_front = (l->kind == LIGHT_DIRECTIONAL) ? 1 : (vsqrdist(r.o, l->v) > vsqrdist(r.o, h.p));
        if (!in_front) continue;
        if (s->objs[i].mat.kind == ORC_MAT_EMMIT) continue;
        if (!g_shadow_exhaustive) return 1;
        any = 1;
    }
    return any;
}

/* ------------------------------------------------------ Material / ColorMap */
static double sq(double x) { return x * x; }              /* GHC (^) with exponent 2 */
/* x ^ (5 :: Int) as GHC compiles it with optimisation on (build.sh:1 uses -O2): GHC.Real carries the rewrite
 * RULES "^2/Int" .. "^5/Int" (GHC ticket #5237), which turn a literal small Int exponent into the FLAT
 * left-associated product u*u*u*u*u; only an un-optimised build runs the square-and-multiply definition
 * ((x*x)*(x*x))*x.  The two differ by at most an ulp or two of the Fresnel factor.  [from memory of base's
 * GHC/Real.hs -- no GHC here to confirm; DESIGN.md section 2] */
/* The other reading stays selectable (orc_set_pow5_variant(1), RAYHS_FLAG_POW5_SQMUL in the kernels) until one Fresnel
 * value from a real GHC -O2 build settles it: tests/golden/pow5_variants.json holds arguments on which the two differ. */
static int g_pow5_sqmul = 0;
void orc_set_pow5_variant(int square_and_multiply) { g_pow5_sqmul = square_and_multiply != 0; }
static double pow5(double x) {
    if (g_pow5_sqmul) return ((x * x) * (x * x)) * x;   /* GHC.Real (^) as defined: square-and-multiply */
    return x * x * x * x * x;                           /* RULES "^5/Int": the default */
}
double orc_r0(double n1, double n2) { return sq((n1 - n2) / (n1 + n2)); }  /* Material.hs:22-24 */
double orc_fresnel(double ior, double cos0) {                              /* Material.hs:26-29 */
    double r = orc_r0(1.0, ior);
    return r + (1 - r) * pow5(1 - cos0);
}

/* Data.Fixed.mod': n - fromInteger (floor (toRational n / toRational d)) * d.
 * The floor is of the EXACT quotient; fma gives the exact sign of n - q*d. */
double orc_mod_floor(double n, double d) {
    double q = floor(n / d);
    if (isfinite(q) && d != 0) {
        for (int it = 0; it < 4; it++) {
            double rem = fma(-q, d, n); /* sign(rem) exact */
            if (d > 0) {
                if (rem < 0) q -= 1; else if (rem >= d) q += 1; else break;
            } else {
                if (rem > 0) q -= 1; else if (rem <= d) q += 1; else break;
            }
        }
    }
    return n - q * d;
}

/* Haskell round: to nearest, ties to even */
static long long hround(double x) { return (long long)nearbyint(x); }
static long long imod(long long a, long long m) { long long r = a % m; if (r != 0 && ((r < 0) != (m < 0))) r += m; return r; }

/* ColorMap.hs:18-58 */
static color color_at(const orc_material *m, const double *tex, uv2 uv) {
    if (m->cmap_kind == ORC_CMAP_FLAT) return V(m->c1[0], m->c1[1], m->c1[2]);
    if (m->cmap_kind == ORC_CMAP_CHECKER) {
        double s = m->size;
        if ((orc_mod_floor(uv.u, s) - (0.5 * s)) * (orc_mod_floor(uv.v, s) - (0.5 * s)) < 0)
            return V(m->c1[0], m->c1[1], m->c1[2]);
        return V(m->c2[0], m->c2[1], m->c2[2]);
    }
    int w = m->tex_w, hh = m->tex_h;
    double u = orc_mod_floor(uv.u, 1) * (double)w;
    double v = orc_mod_floor(uv.v, 1) * (double)hh;
    long long ui = hround(u), vi = hround(v);
    long long x0 = imod(ui - 1, w), x1 = imod(ui, w), y0 = imod(vi - 1, hh), y1 = imod(vi, hh);
    double lx = u - (double)(ui - 1) - 0.5;
    double ly = v - (double)(vi - 1) - 0.5;
#define TEXEL(X, Y) V(tex[3 * ((X) + w * (Y))], tex[3 * ((X) + w * (Y)) + 1], tex[3 * ((X) + w * (Y)) + 2])
    color c0 = TEXEL(x0, y0), c1 = TEXEL(x1, y0), c2 = TEXEL(x0, y1), c3 = TEXEL(x1, y1);
#undef TEXEL
    color cx0 = vadd(vmul(lx, c1), vmul(1 - lx, c0));
    color cx1 = vadd(vmul(lx, c3), vmul(1 - lx, c2));
    return vadd(vmul(ly, cx1), vmul(1 - ly, cx0));
}
void orc_color_at(const orc_material *m, double u, double v, double rgb[3]) {
    uv2 uv = {u, v};
    color c = color_at(m, m->tex, uv);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* Light.hs:12-17 */
static void light_at(const light *l, vec p, vec *ld, color *lc) {
    if (l->kind == LIGHT_DIRECTIONAL) { *ld = l->v; *lc = l->c; return; }
    double d = vdist(l->v, p);
    double falloff = 1.0 / sq(1.0 + d / l->radius);
    *ld = vmul(1 / d, vsub(l->v, p));
    *lc = vmul(falloff, l->c);
}
/* Material.hs:31-33 */
static color diffuse(color cd, color lc, vec l, vec n) { return vmul(hmax(vdot(l, n), 0) * PI_INV, vmulv(cd, lc)); }

static color trace_ray(ctx *cx, const orc_scene *s, int depth, int max_depth, ray r, int kind);

/* RayHs.hs:89-97 */
static color accum_diffuse(ctx *cx, const orc_scene *s, vec p, vec n, color cd) {
    color c = V(0, 0, 0);
    for (int i = 0; i < s->nlights; i++) {
        vec ld; color lc;
        light_at(&s->lights[i], p, &ld, &lc);
        cx->st.rays_shadow++;
        color term = shadow_intersection(cx, s, &s->lights[i], ray_eps(p, ld)) ? V(0, 0, 0) : diffuse(cd, lc, ld, n);
        c = vadd(c, term);
    }
    return c;
}
/* RayHs.hs:99-104 */
static color specular(ctx *cx, const orc_scene *s, int depth, int max_depth, vec v, vec p, vec n) {
    if (depth < max_depth) {
        vec rf = vreflect(v, n);
        return vmul(vdot(rf, n), trace_ray(cx, s, depth + 1, max_depth, ray_eps(p, rf), 1));
    }
    return V(0, 0, 0);
}
/* RayHs.hs:107-147 */
static color irradiance(ctx *cx, int d, int max_depth, const orc_scene *s, const object *o, vec v, vec p, vec n, uv2 uv) {
    const orc_material *m = &o->mat;
    cx->st.shaded_hits++;
    switch (m->kind) {
    case ORC_MAT_DIFFUSE: {
        color cd = color_at(m, o->tex, uv);
        return vadd(vmul(0.2, cd), accum_diffuse(cx, s, p, n, cd));
    }
    case ORC_MAT_PLASTIC: {
        color cd = color_at(m, o->tex, uv);
        color a = accum_diffuse(cx, s, p, n, cd);
        return vadd(a, vmul(orc_fresnel(m->ior, vdot(n, vneg(v))), specular(cx, s, d, max_depth, v, p, n)));
    }
    case ORC_MAT_MIRROR:
        return vmul(orc_fresnel(m->ior, vdot(n, vneg(v))), specular(cx, s, d, max_depth, v, p, n));
    case ORC_MAT_EMMIT:
        return V(m->ce[0], m->ce[1], m->ce[2]);
    case ORC_MAT_TRANSPARENT: {
        int have = 0;
        color radiance = V(0, 0, 0);
        if (d != max_depth) {
            vec refdir;
            if (vrefract(v, n, 1.0, m->ior, &refdir)) {
                ray refr = ray_eps(p, refdir);
                hit oh; int oi;
                cx->st.rays_refract_exit++;
                if (closest_intersection(cx, s, refr, &oh, &oi, 0)) {
                    vec outdir;
                    if (vrefract(refdir, vneg(oh.n), m->ior, 1.0, &outdir)) {
                        radiance = trace_ray(cx, s, d + 1, max_depth, ray_eps(oh.p, outdir), 2);
                        have = 1;
                    }
                }
            }
        }
        color spec = vmul(orc_fresnel(m->ior, vdot(n, vneg(v))), specular(cx, s, d, max_depth, v, p, n));
        if (!have) return spec;
        return vadd(vmul(1 - orc_r0(m->ior, 1.0), radiance), spec);
    }
    case ORC_MAT_SHOWNORMAL: return n;
    case ORC_MAT_SHOWUV: return V(uv.u, uv.v, 0);
    }
    return V(0, 0, 0);
}
/* RayHs.hs:149-154.  kind: 0 primary, 1 reflection, 2 refraction (stats only) */
static color trace_ray(ctx *cx, const orc_scene *s, int depth, int max_depth, ray r, int kind) {
    hit h; int oi;
    if (kind == 0) cx->st.rays_primary++; else if (kind == 1) cx->st.rays_reflect++; else cx->st.rays_refract++;
    if (closest_intersection(cx, s, r, &h, &oi, 0))
        return irradiance(cx, depth, max_depth, s, &s->objs[oi], r.d, h.p, h.n, h.uv);
    return V(0, 0, 0);
}

/* ------------------------------------------------------------ Projection.hs */
/* Projection.hs:41-46 */
static void aspect_size(double w, double h, double *apw, double *aph) {
    double aspect = w / h;
    if (aspect > 1) { *apw = w; *aph = w / aspect; } else { *apw = aspect * h; *aph = h; }
}
/* Projection.hs:22-39 */
static ray ray_from_pixel(const orc_camera *c, double w, double h, double px, double py) {
    vec p = V(c->pos[0], c->pos[1], c->pos[2]);
    vec t = V(c->target[0], c->target[1], c->target[2]);
    vec up = V(c->up[0], c->up[1], c->up[2]);
    double apw, aph;
    aspect_size(w, h, &apw, &aph);
    vec o, d;
    if (c->proj_kind == ORC_PROJ_ORTHOGRAPHIC) {
        o = V(apw * (px - (w / 2)) / w, aph * ((-py) + (h / 2)) / h, 0);
        d = V(0, 0, 1);
    } else {
        double f = 0.5 * h / (tan(0.5) * c->fovy); /* sic: (tan 0.5) * fovy, Projection.hs:35 */
        vec vpp = V(apw * (px - (w / 2)) / w, aph * ((-py) + (h / 2)) / h, f);
        d = vnormalize(vpp);
        o = V(0, 0, 0);
    }
    mat3 m = look_at(p, t, up);
    ray r = {vadd(o, p), mapply(m, d)};
    return r;
}
void orc_ray_from_pixel(const orc_camera *cam, double w, double h, double px, double py, double origin[3], double dir[3]) {
    ray r = ray_from_pixel(cam, w, h, px, py);
    origin[0] = r.o.x; origin[1] = r.o.y; origin[2] = r.o.z;
    dir[0] = r.d.x; dir[1] = r.d.y; dir[2] = r.d.z;
}

/* ----------------------------------------------------------------- sampler */
/* Build-defined counter-based sampler (the reference's StdGen stream is
 * un-pinned third-party code).  splitmix64 finaliser over a
 * counter; 53-bit mantissa -> [0,1).  The HIP kernel implements the same. */
static uint64_t mix64(uint64_t x) {
    x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 27; x *= 0x94d049bb133111ebULL;
    x ^= x >> 31;
    return x;
}
double orc_sample01(uint64_t seed, uint64_t pixel, uint32_t sample, uint32_t dim, uint32_t spp) {
    uint64_t ctr = (pixel * (uint64_t)spp + sample) * 2u + dim;
    uint64_t x = mix64(ctr + mix64(seed + 0x9e3779b97f4a7c15ULL));
    return (double)(x >> 11) * (1.0 / 9007199254740992.0);
}

/* ----------------------------------------------------------------- Mesh.hs */
static void *xrealloc(void *p, size_t n) { void *q = realloc(p, n); if (!q) { fprintf(stderr, "oracle: out of memory\n"); abort(); } return q; }

/* Haskell `read :: String -> Double` on the decimal subset; anything else is
 * "Prelude.read: no parse" in the reference (Mesh.hs:132-133). */
static int hs_read_double(const char *s, size_t n, double *out) {
    char buf[128];
    if (n == 0 || n >= sizeof buf) return 0;
    size_t i = 0;
    if (s[i] == '-') i++;
    size_t d0 = i;
    while (i < n && isdigit((unsigned char)s[i])) i++;
    if (i == d0) return 0;
    if (i < n && s[i] == '.') {
        i++;
        size_t f0 = i;
        while (i < n && isdigit((unsigned char)s[i])) i++;
        if (i == f0) return 0;
    }
    if (i < n && (s[i] == 'e' || s[i] == 'E')) {
        i++;
        if (i < n && (s[i] == '-' || s[i] == '+')) i++;
        size_t e0 = i;
        while (i < n && isdigit((unsigned char)s[i])) i++;
        if (i == e0) return 0;
    }
    if (i != n) return 0;
    memcpy(buf, s, n); buf[n] = 0;
    *out = strtod(buf, NULL);
    return 1;
}
/* readMaybe :: String -> Maybe Int */
static int hs_read_int(const char *s, size_t n, long *out) {
    if (n == 0 || n > 18) return 0;
    size_t i = 0;
    if (s[i] == '-') i++;
    if (i == n) return 0;
    for (size_t k = i; k < n; k++) if (!isdigit((unsigned char)s[k])) return 0;
    char buf[32]; memcpy(buf, s, n); buf[n] = 0;
    *out = strtol(buf, NULL, 10);
    return 1;
}
static int hs_isspace(int c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\f' || c == '\v'; }

typedef struct { long pi; int has_t; long ti; int has_n; long ni; } objindex; /* Mesh.hs:125 */

typedef struct { const char *s; size_t n; } tok;
static int split_words(const char *s, size_t n, tok *out, int max) { /* Prelude.words */
    int k = 0; size_t i = 0;
    while (i < n) {
        while (i < n && hs_isspace((unsigned char)s[i])) i++;
        if (i >= n) break;
        size_t b = i;
        while (i < n && !hs_isspace((unsigned char)s[i])) i++;
        if (k < max) { out[k].s = s + b; out[k].n = i - b; }
        k++;
    }
    return k;
}

/* Mesh.hs:118-221: readOBJ + buildMesh.  flattenVertices only de-duplicates
 * index triples, so a triangle's three vertices are exactly mkV of its three
 * triples (Mesh.hs:205-212); the O(n^2) map is not needed to get the values. */
static int read_obj(const char *path, triangle **out_tris, int *out_n) {
    FILE *f = fopen(path, "rb");
    if (!f) return fail("%s: openFile: does not exist (No such file or directory)", path);
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    char *data = malloc((size_t)sz + 1);
    if (fread(data, 1, (size_t)sz, f) != (size_t)sz) { fclose(f); free(data); return fail("%s: read error", path); }
    fclose(f); data[sz] = 0;
    vec *vs = NULL, *ns = NULL; uv2 *ts = NULL; objindex *is = NULL;
    size_t nv = 0, nn = 0, nt = 0, ni = 0, cv = 0, cn = 0, ct = 0, ci = 0;
    int rc = 0;
    size_t pos = 0;
    enum { MAXTOK = 4096 };
    tok *toks = malloc(sizeof(tok) * MAXTOK);
    while (pos < (size_t)sz && rc == 0) {
        size_t e = pos;
        while (e < (size_t)sz && data[e] != '\n') e++;
        const char *ln = data + pos; size_t len = e - pos;
        pos = e + 1;
        int kind = -1; size_t skip = 0;
        if (len >= 2 && ln[0] == 'v' && ln[1] == ' ') { kind = 0; skip = 2; }
        else if (len >= 3 && ln[0] == 'v' && ln[1] == 'n' && ln[2] == ' ') { kind = 1; skip = 3; }
        else if (len >= 3 && ln[0] == 'v' && ln[1] == 't' && ln[2] == ' ') { kind = 2; skip = 3; }
        else if (len >= 2 && ln[0] == 'f' && ln[1] == ' ') { kind = 3; skip = 2; }
        if (kind < 0) continue;
        int nw = split_words(ln + skip, len - skip, toks, MAXTOK);
        if (nw > MAXTOK) { rc = fail("%s: line too long", path); break; }
        if (kind <= 2) {
            double d[3];
            int want = kind == 2 ? 2 : 3;
            /* `length floats` forces only the spine; elements are read only when the count matches */
            if (nw != want) continue;
            for (int k = 0; k < want; k++)
                if (!hs_read_double(toks[k].s, toks[k].n, &d[k])) { rc = fail("Prelude.read: no parse"); break; }
            if (rc) break;
            if (kind == 0) { if (nv == cv) { cv = cv ? cv * 2 : 1024; vs = xrealloc(vs, cv * sizeof *vs); } vs[nv++] = V(d[0], d[1], d[2]); }
            else if (kind == 1) { if (nn == cn) { cn = cn ? cn * 2 : 1024; ns = xrealloc(ns, cn * sizeof *ns); } ns[nn++] = V(d[0], d[1], d[2]); }
            else { if (nt == ct) { ct = ct ? ct * 2 : 1024; ts = xrealloc(ts, ct * sizeof *ts); } ts[nt].u = d[0]; ts[nt].v = d[1]; nt++; }
        } else {
            /* readFace: mapMaybe readOBJIndex (words s); needs > 2 parsed indices */
            objindex tmp[MAXTOK]; int k = 0;
            for (int w = 0; w < nw; w++) {
                /* splitOn "/" */
                const char *s = toks[w].s; size_t n = toks[w].n;
                tok parts[3]; int np = 0; size_t b = 0;
                for (size_t i = 0; i <= n; i++) {
                    if (i == n || s[i] == '/') { if (np < 3) { parts[np].s = s + b; parts[np].n = i - b; } np++; b = i + 1; }
                }
                objindex oi = {0, 0, 0, 0, 0};
                if (!hs_read_int(parts[0].s, parts[0].n, &oi.pi)) continue; /* parseIndex _ = Nothing */
                if (np >= 2) oi.has_t = hs_read_int(parts[1].s, parts[1].n, &oi.ti);
                if (np >= 3) oi.has_n = hs_read_int(parts[2].s, parts[2].n, &oi.ni);
                tmp[k++] = oi;
            }
            if (k > 2) {
                if (ni + (size_t)k > ci) { while (ni + (size_t)k > ci) ci = ci ? ci * 2 : 4096; is = xrealloc(is, ci * sizeof *is); }
                memcpy(is + ni, tmp, sizeof(objindex) * (size_t)k); ni += (size_t)k;
            }
        }
    }
    free(toks);
    triangle *tris = NULL; int ntr = 0;
    if (rc == 0) {
        if (ni % 3 != 0) rc = fail("%s: Non-exhaustive patterns in function faces (index count %zu not a multiple of 3)", path, ni);
        else {
            ntr = (int)(ni / 3);
            tris = malloc(sizeof(triangle) * (size_t)(ntr ? ntr : 1));
            for (size_t k = 0; k < ni && rc == 0; k++) {
                objindex oi = is[k];
                vertex vx;
                if (oi.pi < 1 || (size_t)oi.pi > nv) { rc = fail("%s: Prelude.!!: index out of range (position %ld)", path, oi.pi); break; }
                vx.p = vs[oi.pi - 1];
                vx.n = V(0, 0, 0); vx.uv.u = 0; vx.uv.v = 0;
                if (oi.has_n) { if (oi.ni < 1 || (size_t)oi.ni > nn) { rc = fail("%s: Prelude.!!: index out of range (normal %ld)", path, oi.ni); break; } vx.n = ns[oi.ni - 1]; }
                if (oi.has_t) { if (oi.ti < 1 || (size_t)oi.ti > nt) { rc = fail("%s: Prelude.!!: index out of range (uv %ld)", path, oi.ti); break; } vx.uv = ts[oi.ti - 1]; }
                vertex *dst = (k % 3 == 0) ? &tris[k / 3].a : (k % 3 == 1) ? &tris[k / 3].b : &tris[k / 3].c;
                *dst = vx;
            }
        }
    }
    free(vs); free(ns); free(ts); free(is); free(data);
    if (rc) { free(tris); return rc; }
    *out_tris = tris; *out_n = ntr;
    return 0;
}

/* --------------------------------------------------------------- KDTree.hs */
/* KDTree.hs:19-31 */
static box bounding_box(const triangle *t, int n) {
    box b = {V(INFINITY, INFINITY, INFINITY), V(-INFINITY, -INFINITY, -INFINITY)};
    for (int i = 0; i < n; i++) {
        b.lo = vminv(b.lo, t[i].a.p); b.hi = vmaxv(b.hi, t[i].a.p);
        b.lo = vminv(b.lo, t[i].b.p); b.hi = vmaxv(b.hi, t[i].b.p);
        b.lo = vminv(b.lo, t[i].c.p); b.hi = vmaxv(b.hi, t[i].c.p);
    }
    return b;
}
/* KDTree.hs:71-74 */
static vec bary_center(const triangle *t) { return vmul(1.0 / 3, vadd(vadd(t->a.p, t->b.p), t->c.p)); }

/* KDTree.hs:79-90 */
static kdtree *build_node(const triangle *tris, int n, int depth, int axis) {
    kdtree *k = calloc(1, sizeof *k);
    if (n == 0) { k->kind = KD_EMPTY; return k; }
    k->b = bounding_box(tris, n);
    if (n < 20 || depth >= 100) {
        k->kind = KD_LEAF;
        k->tris = malloc(sizeof(triangle) * (size_t)n);
        memcpy(k->tris, tris, sizeof(triangle) * (size_t)n);
        k->ntris = n;
        return k;
    }
    double split = 0.5 * (vidx(k->b.hi, axis) + vidx(k->b.lo, axis));
    triangle *L = malloc(sizeof(triangle) * (size_t)n), *R = malloc(sizeof(triangle) * (size_t)n);
    int nl = 0, nr = 0;
    for (int i = 0; i < n; i++) {
        double c = vidx(bary_center(&tris[i]), axis);
        if (c < split) L[nl++] = tris[i];
        if (split <= c) R[nr++] = tris[i];
    }
    int next = (axis + 1) % 3;
    k->kind = KD_NODE;
    k->l = build_node(L, nl, depth + 1, next);
    k->r = build_node(R, nr, depth + 1, next);
    free(L); free(R);
    return k;
}
static void free_tree(kdtree *k) {
    if (!k) return;
    free_tree(k->l); free_tree(k->r); free(k->tris); free(k);
}
static void tree_stats(const kdtree *k, int depth, orc_tree_stats *s, long *leaf_sum) {
    if (k->kind == KD_EMPTY) { s->empties++; return; }
    if (depth > s->max_depth) s->max_depth = depth;
    if (k->kind == KD_LEAF) {
        s->leaves++; *leaf_sum += k->ntris;
        if (k->ntris < s->leaf_min) s->leaf_min = k->ntris;
        if (k->ntris > s->leaf_max) s->leaf_max = k->ntris;
        return;
    }
    s->inner++;
    tree_stats(k->l, depth + 1, s, leaf_sum);
    tree_stats(k->r, depth + 1, s, leaf_sum);
}

/* ------------------------------------------------------------ scene assembly */
orc_scene *orc_scene_new(void) { return calloc(1, sizeof(orc_scene)); }
void orc_scene_free(orc_scene *s) {
    if (!s) return;
    for (int i = 0; i < s->nobjs; i++) { free_tree(s->objs[i].tree); free(s->objs[i].tris); free(s->objs[i].tex); }
    free(s->objs); free(s->lights); free(s);
}
static object *new_object(orc_scene *s, const orc_material *m) {
    if (s->nobjs == s->cobjs) { s->cobjs = s->cobjs ? s->cobjs * 2 : 16; s->objs = xrealloc(s->objs, sizeof(object) * (size_t)s->cobjs); }
    object *o = &s->objs[s->nobjs++];
    memset(o, 0, sizeof *o);
    o->mat = *m;
    if ((m->kind == ORC_MAT_DIFFUSE || m->kind == ORC_MAT_PLASTIC) && m->cmap_kind == ORC_CMAP_TEXTURE && m->tex) {
        size_t n = (size_t)m->tex_w * (size_t)m->tex_h * 3;
        o->tex = malloc(n * sizeof(double));
        memcpy(o->tex, m->tex, n * sizeof(double));
    }
    o->mat.tex = o->tex;
    return o;
}
int orc_add_plane(orc_scene *s, const double p[3], const double n[3], const double t[3], const orc_material *m) {
    object *o = new_object(s, m);
    o->kind = GEOM_PLANE; o->p = V(p[0], p[1], p[2]); o->n = V(n[0], n[1], n[2]); o->t = V(t[0], t[1], t[2]);
    return 0;
}
int orc_add_sphere(orc_scene *s, const double c[3], double radius, const orc_material *m) {
    object *o = new_object(s, m);
    o->kind = GEOM_SPHERE; o->p = V(c[0], c[1], c[2]); o->radius = radius;
    return 0;
}
int orc_add_mesh_obj(orc_scene *s, const char *path, int top_is_translate, int n_xf, const orc_xform *xf, const orc_material *m) {
    triangle *tris; int n;
    if (read_obj(path, &tris, &n)) return -1;
    /* Mesh.hs:89-101: a bare Translate moves positions only; anything else is
     * applied to positions AND normals (normals treated as points). */
    for (int i = 0; i < n; i++) {
        vertex *vx[3] = {&tris[i].a, &tris[i].b, &tris[i].c};
        for (int k = 0; k < 3; k++) {
            vx[k]->p = xf_seq(n_xf, xf, vx[k]->p);
            if (!top_is_translate) vx[k]->n = xf_seq(n_xf, xf, vx[k]->n);
        }
    }
    object *o = new_object(s, m);
    o->kind = GEOM_MESH; o->tris = tris; o->ntris = n;
    o->tree = build_node(tris, n, 0, 0); /* KDTree.hs:68-69 */
    return 0;
}
static light *new_light(orc_scene *s) {
    if (s->nlights == s->clights) { s->clights = s->clights ? s->clights * 2 : 8; s->lights = xrealloc(s->lights, sizeof(light) * (size_t)s->clights); }
    return &s->lights[s->nlights++];
}
int orc_add_light_directional(orc_scene *s, const double d[3], const double c[3]) {
    light *l = new_light(s); l->kind = LIGHT_DIRECTIONAL; l->v = V(d[0], d[1], d[2]); l->c = V(c[0], c[1], c[2]); l->radius = 0;
    return 0;
}
int orc_add_light_point(orc_scene *s, const double p[3], const double c[3], double radius) {
    light *l = new_light(s); l->kind = LIGHT_POINT; l->v = V(p[0], p[1], p[2]); l->c = V(c[0], c[1], c[2]); l->radius = radius;
    return 0;
}
int orc_num_objects(const orc_scene *s) { return s->nobjs; }
int orc_mesh_tree_stats(const orc_scene *s, int object, orc_tree_stats *out) {
    if (object < 0 || object >= s->nobjs || s->objs[object].kind != GEOM_MESH) return fail("object %d is not a mesh", object);
    memset(out, 0, sizeof *out);
    out->leaf_min = 1 << 30; out->tris = s->objs[object].ntris;
    long sum = 0;
    tree_stats(s->objs[object].tree, 0, out, &sum);
    out->leaf_mean = out->leaves ? (double)sum / out->leaves : 0;
    if (!out->leaves) out->leaf_min = 0;
    return 0;
}
int orc_mesh_triangles(const orc_scene *s, int object, double *out, int max_tris) {
    if (object < 0 || object >= s->nobjs || s->objs[object].kind != GEOM_MESH) return fail("object %d is not a mesh", object);
    int n = s->objs[object].ntris;
    for (int i = 0; i < n && i < max_tris; i++) {
        const vertex *vx[3] = {&s->objs[object].tris[i].a, &s->objs[object].tris[i].b, &s->objs[object].tris[i].c};
        for (int k = 0; k < 3; k++) {
            double *d = out + (size_t)i * 24 + k * 8;
            d[0] = vx[k]->p.x; d[1] = vx[k]->p.y; d[2] = vx[k]->p.z;
            d[3] = vx[k]->n.x; d[4] = vx[k]->n.y; d[5] = vx[k]->n.z;
            d[6] = vx[k]->uv.u; d[7] = vx[k]->uv.v;
        }
    }
    return n;
}

/* ------------------------------------------------------------------ probes */
static void fill_hit(orc_hit *out, int have, int oi, const hit *h) {
    memset(out, 0, sizeof *out);
    out->hit = have; out->object = have ? oi : -1;
    if (!have) return;
    out->t = h->t;
    out->p[0] = h->p.x; out->p[1] = h->p.y; out->p[2] = h->p.z;
    out->n[0] = h->n.x; out->n[1] = h->n.y; out->n[2] = h->n.z;
    out->uv[0] = h->uv.u; out->uv[1] = h->uv.v;
}
void orc_closest_hit(const orc_scene *s, const double o[3], const double d[3], orc_hit *out) {
    ctx cx; memset(&cx, 0, sizeof cx);
    ray r = {V(o[0], o[1], o[2]), V(d[0], d[1], d[2])};
    hit h; int oi = -1;
    int have = closest_intersection(&cx, s, r, &h, &oi, 0);
    fill_hit(out, have, oi, &h);
}
void orc_closest_hit_bruteforce(const orc_scene *s, const double o[3], const double d[3], orc_hit *out) {
    ctx cx; memset(&cx, 0, sizeof cx);
    ray r = {V(o[0], o[1], o[2]), V(d[0], d[1], d[2])};
    hit h; int oi = -1;
    int have = closest_intersection(&cx, s, r, &h, &oi, 1);
    fill_hit(out, have, oi, &h);
}
void orc_trace_ray(const orc_scene *s, const double o[3], const double d[3], int depth, int max_depth, double rgb[3]) {
    ctx cx; memset(&cx, 0, sizeof cx);
    ray r = {V(o[0], o[1], o[2]), V(d[0], d[1], d[2])};
    color c = trace_ray(&cx, s, depth, max_depth, r, 0);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

/* ------------------------------------------------------------------ render */
static void stats_add(orc_stats *a, const orc_stats *b) {
    a->rays_primary += b->rays_primary; a->rays_reflect += b->rays_reflect; a->rays_refract += b->rays_refract;
    a->rays_refract_exit += b->rays_refract_exit; a->rays_shadow += b->rays_shadow;
    a->nodes_tested += b->nodes_tested; a->tris_tested += b->tris_tested; a->prims_tested += b->prims_tested;
    a->shaded_hits += b->shaded_hits;
}

/* RayHs.hs:156-166 (spp==1) and RayHs.hs:169-188 (spp>1); Image.hs:31-36 */
int orc_render(const orc_scene *s, const orc_camera *cam, int w, int h, int max_depth, int spp, uint64_t seed,
               int row0, int row1, double *rgb_out, orc_stats *stats, int nthreads) {
    if (w <= 0 || h <= 0 || row0 < 0 || row1 > h || row0 > row1 || spp < 1) return fail("bad render arguments");
    orc_stats total; memset(&total, 0, sizeof total);
    double dw = (double)w, dh = (double)h;
#ifdef _OPENMP
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
#endif
    {
        ctx cx; memset(&cx, 0, sizeof cx);
#ifdef _OPENMP
#pragma omp for collapse(2) schedule(dynamic, 16)
#endif
        for (int y = row0; y < row1; y++) {
            for (int x = 0; x < w; x++) {
                /* pixelCoord: (i mod w, i div w) as Doubles, Image.hs:31-32 */
                double px = (double)x, py = (double)y;
                color c;
                if (spp == 1) {
                    c = trace_ray(&cx, s, 0, max_depth, ray_from_pixel(cam, dw, dh, px, py), 0);
                } else {
                    uint64_t pixel = (uint64_t)x + (uint64_t)w * (uint64_t)y;
                    color sum = V(0, 0, 0);
                    for (int k = 0; k < spp; k++) {
                        double oi = orc_sample01(seed, pixel, (uint32_t)k, 0, (uint32_t)spp) - 0.5;
                        double oj = orc_sample01(seed, pixel, (uint32_t)k, 1, (uint32_t)spp) - 0.5;
                        ray r = ray_from_pixel(cam, dw, dh, px + oi, py + oj);
                        sum = vadd(sum, trace_ray(&cx, s, 0, max_depth, r, 0));
                    }
                    c = vmul(1.0 / (double)spp, sum); /* average, RayHs.hs:169-171 */
                }
                double *o = rgb_out + ((size_t)(y - row0) * (size_t)w + (size_t)x) * 3;
                o[0] = c.x; o[1] = c.y; o[2] = c.z;
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        stats_add(&total, &cx.st);
    }
    if (stats) *stats = total;
    return 0;
}

/* ---------------------------------------------------------------- Image.hs */
/* Image.hs:54-55: truncate (255 * min c 1) */
int orc_to_int_c(double c) {
    double v = 255 * hmin(c, 1);
    return (int)v; /* C conversion truncates toward zero like Haskell truncate */
}
/* Image.hs:60-75 */
int orc_write_ppm(const char *path, int w, int h, const double *rgb) {
    FILE *f = fopen(path, "wb");
    if (!f) return fail("%s: cannot open for writing", path);
    fprintf(f, "P3\n%d %d\n255\n", w, h);
    for (int y = 0; y < h; y++) {
        if (y) fputc('\n', f);
        for (int x = 0; x < w; x++) {
            const double *p = rgb + ((size_t)y * (size_t)w + (size_t)x) * 3;
            fprintf(f, "%d %d %d  ", orc_to_int_c(p[0]), orc_to_int_c(p[1]), orc_to_int_c(p[2]));
        }
    }
    fclose(f);
    return 0;
}
