"""Second, independent restatement of the reference's trace path -- plain Python floats.

TEST INFRASTRUCTURE ONLY, like everything under oracle/.  PARITY UNPINNED: this file is
*restatement-derived* too (no GHC exists here, the reference holds no expected outputs); what it
removes is SINGLE-TRANSCRIPTION risk: it was written from the Haskell sources alone, function by
function in the reference's own shape (Maybe = None, lists, foldl, recursion), without consulting
oracle/rayhs_oracle.c, and `oracle/second/check_second.py` asserts that the two restatements agree
BIT FOR BIT on whole small frames.  It never travels into a product path and is not run on the GPU box.

A Python float is an IEEE binary64 and + - * / and math.sqrt are correctly rounded with no fused
multiply-add, i.e. exactly GHC's Double arithmetic; tan / atan / acos / sin / cos go to the same
glibc libm the C oracle calls.

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
import json
import math
from fractions import Fraction

INF = float("inf")                     # src/Math.hs:17-18
EPS = 0.000001                         # src/Geometry.hs:31-32
PI_INV = 1 / math.pi                   # src/Math.hs:12-13


# ----------------------------------------------------------------- Prelude semantics
def hs_min(x, y):                      # Ord Double: min x y = if x <= y then x else y
    return x if x <= y else y


def hs_max(x, y):                      # max x y = if x <= y then y else x
    return y if x <= y else x


def hs_compare_gt(x, y):
    """compare x y == GT for Double: LT if x < y, EQ if x == y, otherwise GT (NaN on either side -> GT)."""
    return not (x < y) and not (x == y)


def minimum_by_t(hits):
    """minimumBy (compare `on` t) = foldl1 (\\x y -> case cmp x y of GT -> y; _ -> x): the FIRST minimal
    element wins (src/Geometry.hs:54-57, src/RayHs.hs:67-71)."""
    best = hits[0]
    for h in hits[1:]:
        if hs_compare_gt(best["t"], h["t"]):
            best = h
    return best


def pow2(u):                           # x ^ (2 :: Int)
    return u * u


POW5_SQUARE_AND_MULTIPLY = False      # the other reading of x ^ (5 :: Int), kept selectable


def pow5(u):
    """x ^ (5 :: Int) under GHC -O (build.sh:1 uses -O2): GHC.Real's RULES "^5/Int" rewrite a literal Int
    exponent 5 to the flat product u*u*u*u*u (left-associated); see DESIGN.md section 2 for why this and not the
    square-and-multiply of the un-optimised definition."""
    if POW5_SQUARE_AND_MULTIPLY:
        return ((u * u) * (u * u)) * u
    return u * u * u * u * u


def mod_prime(n, d):
    """Data.Fixed.mod' n d = n - fromInteger (floor (toRational n / toRational d)) * d (src/ColorMap.hs:22,57-58)."""
    f = math.floor(Fraction(n) / Fraction(d))
    return n - float(f) * d


def hs_round(x):                       # RealFrac Double round: ties to even (Python's round does the same)
    return int(round(x))


# ----------------------------------------------------------------------- src/Vec.hs
def vadd(a, b): return (a[0] + b[0], a[1] + b[1], a[2] + b[2])          # :36
def vsub(a, b): return (a[0] - b[0], a[1] - b[1], a[2] - b[2])          # :40
def vmulv(a, b): return (a[0] * b[0], a[1] * b[1], a[2] * b[2])         # :38
def vneg(a): return (-a[0], -a[1], -a[2])                               # :42
def mul(l, a): return tuple(l * c for c in a)                           # Ext: :66-84 (Vec, UV and Color alike)
def dot(a, b): return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]           # :103-105
def cross(a, b):                                                        # :107-110
    return (a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0])
def sqr_len(v): return dot(v, v)                                        # :112-114
def sqr_dist(v, w): return sqr_len(vsub(v, w))                          # :116-118
def dist(v, w): return math.sqrt(sqr_dist(v, w))                        # :120-122
def normalize(v): return mul(_fdiv(1.0, math.sqrt(sqr_len(v))), v)      # :124-126
def reflect(v, n): return vsub(v, mul(2 * dot(v, n), n))                # :128-130


def refract(i, n, n1, n2):                                              # :132-140
    n1n2 = _fdiv(n1, n2)
    cos0 = -(dot(i, n))
    sin20 = n1n2 * n1n2 * (1 - cos0 * cos0)
    if sin20 > 1:
        return None
    coeff = n1n2 * cos0 - math.sqrt(1.0 - sin20)
    return vadd(mul(n1n2, i), mul(coeff, n))


def min_v(a, b): return (hs_min(a[0], b[0]), hs_min(a[1], b[1]), hs_min(a[2], b[2]))    # :142-144
def max_v(a, b): return (hs_max(a[0], b[0]), hs_max(a[1], b[1]), hs_max(a[2], b[2]))    # :146-148


def orthonormal(r):                                                     # :151-158
    x, y, z = r
    if abs(x) < abs(y) and abs(x) < abs(z):
        s = normalize((0.0, -z, y))
    elif abs(y) < abs(x) and abs(y) < abs(z):
        s = normalize((-z, 0.0, x))
    else:
        s = normalize((-y, x, 0.0))
    return r, s, cross(r, s)


def _fdiv(a, b):
    """Double division with IEEE results where Python raises (x/0 -> inf / nan)."""
    try:
        return a / b
    except ZeroDivisionError:
        if a != a or a == 0:
            return float("nan")
        return math.copysign(INF, a) * math.copysign(1.0, b)


# ----------------------------------------------------------------------- src/Mat.hs
def mat_mul(a, b):                                                      # :27-33, row-major 9-tuples
    a1, b1, c1, d1, e1, f1, g1, h1, i1 = a
    a2, b2, c2, d2, e2, f2, g2, h2, i2 = b
    return (a1 * a2 + b1 * d2 + c1 * g2, a1 * b2 + b1 * e2 + c1 * h2, a1 * c2 + b1 * f2 + c1 * i2,
            d1 * a2 + e1 * d2 + f1 * g2, d1 * b2 + e1 * e2 + f1 * h2, d1 * c2 + e1 * f2 + f1 * i2,
            g1 * a2 + h1 * d2 + i1 * g2, g1 * b2 + h1 * e2 + i1 * h2, g1 * c2 + h1 * f2 + i1 * i2)


def apply(m, v):                                                        # :40-44
    a, b, c, d, e, f, g, h, i = m
    x, y, z = v
    return (a * x + b * y + c * z, d * x + e * y + f * z, g * x + h * y + i * z)


def transpose(m):                                                       # :47-49
    a, b, c, d, r, f, g, h, i = m
    return (a, d, g, b, r, h, c, f, i)


def rotate_z(t): return (math.cos(t), -(math.sin(t)), 0.0, math.sin(t), math.cos(t), 0.0, 0.0, 0.0, 1.0)    # :58-62
def rotate_y(t): return (math.cos(t), 0.0, math.sin(t), 0.0, 1.0, 0.0, -(math.sin(t)), 0.0, math.cos(t))    # :64-68
def rotate_x(t): return (1.0, 0.0, 0.0, 0.0, math.cos(t), -(math.sin(t)), 0.0, math.sin(t), math.cos(t))    # :70-74


def from_columns(v1, v2, v3):                                           # :83-87
    return (v1[0], v2[0], v3[0], v1[1], v2[1], v3[1], v1[2], v2[2], v3[2])


def rotate(axis, angle):                                                # :76-81: mt * rot * m, (*) is infixl
    r, s, t = orthonormal(axis)
    m = from_columns(r, s, t)
    return mat_mul(mat_mul(transpose(m), rotate_x(angle)), m)


def look_at(pos, target, tup):                                          # :89-93
    forward = normalize(vsub(target, pos))
    right = normalize(cross(tup, forward))
    up = cross(forward, right)
    return from_columns(right, up, forward)


# ----------------------------------------------------------------- src/Transform.hs:16-24
def transform_point(t, p):
    k = t["type"]
    if k == "translate":
        return vadd(p, _vec(t["vector"]))
    if k == "scale":
        s = _vec(t["vector"])
        return (s[0] * p[0], s[1] * p[1], s[2] * p[2])
    if k == "rotate":
        return apply(rotate(_vec(t["axis"]), float(t["angle"])), p)
    if k == "rotateX":
        return apply(rotate_x(float(t["angle"])), p)
    if k == "rotateY":
        return apply(rotate_y(float(t["angle"])), p)
    if k == "rotateZ":
        return apply(rotate_z(float(t["angle"])), p)
    if k == "sequence":                                                 # foldl (flip transform) p transforms
        for s in t["transforms"]:
            p = transform_point(s, p)
        return p
    raise ValueError("Unknown type for transform " + k)


# ------------------------------------------------------------------ src/Geometry.hs
def ray_at(ray, t):                                                     # :27-29
    return vadd(ray[0], mul(t, ray[1]))


def ray_eps(p, n):                                                      # :34-36: offset ALONG n, direction n
    return (vadd(p, mul(EPS, n)), n)


def hit(p, n, uv, t):
    return {"p": p, "n": n, "uv": uv, "t": t}


def closest_hit(maybe_hits):                                            # :54-57
    xs = [h for h in maybe_hits if h is not None]
    return None if not xs else minimum_by_t(xs)


def inter_plane(ray, shape):                                            # :70-79
    o, d = ray
    p, n, t = shape["point"], shape["normal"], shape["tangent"]
    d_dot_n = dot(d, n)
    time = _fdiv(dot(n, vsub(p, o)), d_dot_n)
    if abs(d_dot_n) > 0 and time > 0:
        pos = ray_at(ray, time)
        b = cross(t, n)
        rel = vsub(pos, p)
        return hit(pos, n, (dot(t, rel), dot(b, rel)), time)
    return None


def polar(p):                                                           # :96
    return (PI_INV * math.atan(_fdiv(p[2], p[0])), PI_INV * _acos(p[1]))


def _acos(x):
    return math.acos(x) if -1.0 <= x <= 1.0 else float("nan")


def inter_sphere(ray, shape):                                           # :81-95
    o, d = ray
    ct, r = shape["center"], shape["radius"]
    a = dot(d, d)
    b = 2.0 * dot(d, vsub(o, ct))
    c = sqr_len(vsub(o, ct)) - pow2(r)
    delta = pow2(b) - 4.0 * a * c
    if delta < 0.0:
        return None
    sq = math.sqrt(delta) if delta >= 0 else float("nan")
    t0 = _fdiv(0.5 * ((-b) - sq), a)
    if t0 > 0:
        p0 = ray_at(ray, t0)
        n0 = normalize(vsub(p0, ct))
        return hit(p0, n0, polar(n0), t0)
    t1 = _fdiv(0.5 * ((-b) + sq), a)
    if t1 > 0:
        p1 = ray_at(ray, t1)
        n1 = normalize(vsub(p1, ct))
        return hit(p1, n1, polar(n1), t1)
    return None


# ---------------------------------------------------------------------- src/Mesh.hs
def barycentric_interp(a, p, b, q, c, r):                               # :53-57
    return tuple((a * pi + b * qi) + c * ri for pi, qi, ri in zip(p, q, r))


def triangle_intersection(ray, tri):                                    # :59-82
    o, d = ray
    (p0, n0, uv0), (p1, n1, uv1), (p2, n2, uv2) = tri
    e1 = vsub(p1, p0)
    e2 = vsub(p2, p0)
    p = cross(d, e2)
    det = dot(e1, p)
    idet = _fdiv(1.0, det)
    t0 = vsub(o, p0)
    u = idet * dot(t0, p)
    q = cross(t0, e1)
    v = idet * dot(d, q)
    t = idet * dot(e2, q)
    if abs(det) < EPS or u < 0 or u > 1 or v < 0 or (u + v) > 1 or t < EPS:
        return None
    w = 1 - u - v
    return hit(ray_at(ray, t), barycentric_interp(u, n1, v, n2, w, n0), barycentric_interp(u, uv1, v, uv2, w, uv0), t)


def read_obj(path):                                                     # :118-221
    def read_index(s):                                                  # readOBJIndex / parseIndex :146-156
        parts = []
        for f in s.split("/"):
            try:
                parts.append(int(f))
            except ValueError:
                parts.append(None)
        if parts and parts[0] is not None:
            if len(parts) >= 3:
                return (parts[0], parts[1], parts[2])
            if len(parts) == 2:
                return (parts[0], parts[1], None)
            return (parts[0], None, None)
        return None

    pos, norms, uvs, inds = [], [], [], []
    with open(path) as f:
        for line in f.read().split("\n"):                               # readLine :171-176
            if line.startswith("v "):
                fl = [float(w) for w in line[2:].split()]
                if len(fl) == 3:
                    pos.append(tuple(fl))
            elif line.startswith("vn "):
                fl = [float(w) for w in line[3:].split()]
                if len(fl) == 3:
                    norms.append(tuple(fl))
            elif line.startswith("vt "):
                fl = [float(w) for w in line[3:].split()]
                if len(fl) == 2:
                    uvs.append(tuple(fl))
            elif line.startswith("f "):
                idx = [i for i in (read_index(w) for w in line[2:].split()) if i is not None]
                if len(idx) > 2:
                    inds.extend(idx)                                    # all indices concatenated (:183)
    verts, ids, seen = [], [], {}
    for oi in inds:                                                     # flattenVertices :195-212
        if oi not in seen:
            pi, ui, ni = oi                                             # (position, texture, normal) order of v/vt/vn
            seen[oi] = len(verts)
            verts.append((pos[pi - 1], norms[ni - 1] if ni is not None else (0.0, 0.0, 0.0),
                          uvs[ui - 1] if ui is not None else (0.0, 0.0)))
        ids.append(seen[oi])
    return verts, ids


def mesh_transform(verts, trans):                                       # :89-101
    if trans["type"] == "translate":
        return [(transform_point(trans, p), n, uv) for p, n, uv in verts]
    return [(transform_point(trans, p), transform_point(trans, n), uv) for p, n, uv in verts]


def triangles(verts, ids):                                              # :105-109
    return [(verts[ids[k]], verts[ids[k + 1]], verts[ids[k + 2]]) for k in range(0, len(ids) - 2, 3)]


# -------------------------------------------------------------------- src/KDTree.hs
def build_bounding_box(tris):                                           # :16-31
    lo, hi = (INF, INF, INF), (-INF, -INF, -INF)
    for (a, _, _), (b, _, _), (c, _, _) in tris:
        for p in (a, b, c):
            lo, hi = min_v(lo, p), max_v(hi, p)
    return lo, hi


def ray_inter_box(ray, box):                                            # :39-56
    (ox, oy, oz), (dx, dy, dz) = ray
    (mx, my, mz), (mX, mY, mZ) = box
    idx, idy, idz = _fdiv(1.0, dx), _fdiv(1.0, dy), _fdiv(1.0, dz)
    t1, t2 = idx * (mx - ox), idx * (mX - ox)
    t3, t4 = idy * (my - oy), idy * (mY - oy)
    t5, t6 = idz * (mz - oz), idz * (mZ - oz)
    tmin = hs_max(hs_max(hs_min(t1, t2), hs_min(t3, t4)), hs_min(t5, t6))    # max3 a b = max (max a b)
    tmax = hs_min(hs_min(hs_max(t1, t2), hs_max(t3, t4)), hs_max(t5, t6))
    return None if (tmax < 0 or tmin > tmax) else tmin


def bary_center(tri):                                                   # :71-74
    (a, _, _), (b, _, _), (c, _, _) = tri
    return mul(1 / 3, vadd(vadd(a, b), c))


def build_node(tris, depth, axis):                                      # :79-90
    if not tris:
        return ("Empty",)
    box = build_bounding_box(tris)
    if len(tris) < 20 or depth >= 100:
        return ("Leaf", box, tris)
    split = 0.5 * (box[1][axis] + box[0][axis])
    nxt = (axis + 1) % 3
    left = build_node([t for t in tris if bary_center(t)[axis] < split], depth + 1, nxt)
    right = build_node([t for t in tris if split <= bary_center(t)[axis]], depth + 1, nxt)
    return ("Node", box, left, right)


def ray_inter(ray, tree):                                               # :96-115
    if tree[0] == "Empty":
        return None
    if ray_inter_box(ray, tree[1]) is None:
        return None
    if tree[0] == "Leaf":
        return closest_hit([triangle_intersection(ray, t) for t in tree[2]])
    l_hit = ray_inter(ray, tree[2])
    r_hit = ray_inter(ray, tree[3])
    if l_hit is None:
        return r_hit
    if r_hit is None:
        return l_hit
    return l_hit if l_hit["t"] < r_hit["t"] else r_hit                  # minMaybeHit


# ------------------------------------------------- src/Light.hs, Material.hs, ColorMap.hs
def light_at(light, p):                                                 # Light.hs:12-17
    if light["type"] == "directional":
        return light["direction"], light["color"]
    d = dist(light["position"], p)
    falloff = _fdiv(1.0, pow2(1.0 + _fdiv(d, light["radius"])))
    return mul(_fdiv(1.0, d), vsub(light["position"], p)), mul(falloff, light["color"])


def r0(n1, n2): return pow2(_fdiv(n1 - n2, n1 + n2))                    # Material.hs:22-24


def fresnel(ior, cos0):                                                 # Material.hs:26-29
    r = r0(1.0, ior)
    return r + (1 - r) * pow5(1 - cos0)


def diffuse(cd, lc, l, n):                                              # Material.hs:31-33
    return mul(hs_max(dot(l, n), 0) * PI_INV, vmulv(cd, lc))


def color_at(cmap, uv):                                                 # ColorMap.hs:18-39
    k = cmap["type"]
    if k == "flat":
        return cmap["color"]
    u, v = uv
    if k == "checker":
        s = cmap["size"]
        return cmap["color1"] if (mod_prime(u, s) - (0.5 * s)) * (mod_prime(v, s) - (0.5 * s)) < 0 else cmap["color2"]
    w, h, px = cmap["bitmap"]
    pu, pv = mod_prime(u, 1) * float(w), mod_prime(v, 1) * float(h)     # toPixel / repeatUV :47-58
    ui, vi = hs_round(pu), hs_round(pv)
    x0, x1, y0, y1 = (ui - 1) % w, ui % w, (vi - 1) % h, vi % h
    lx = pu - float(ui - 1) - 0.5
    ly = pv - float(vi - 1) - 0.5
    c0, c1, c2, c3 = px[x0 + w * y0], px[x1 + w * y0], px[x0 + w * y1], px[x1 + w * y1]
    cx0 = vadd(mul(lx, c1), mul(1 - lx, c0))                            # bilinearInterp :41-45
    cx1 = vadd(mul(lx, c3), mul(1 - lx, c2))
    return vadd(mul(ly, cx1), mul(1 - ly, cx0))


def read_bitmap_ppm(path):                                              # Bitmap.hs:31-37
    lines = open(path).read().split("\n")
    w, h = (int(x) for x in lines[1].split()[:2])
    vals = [float(t) / 255 for ln in lines[3:] for t in ln.split()]
    return w, h, [tuple(vals[i:i + 3]) for i in range(0, len(vals) - 2, 3)]


# --------------------------------------------------------------- src/Projection.hs:22-46
def aspect_size(w, h):
    aspect = w / h
    return (w, w / aspect) if aspect > 1 else (aspect * h, h)


def unproject(w, h, proj, px, py):
    apw, aph = aspect_size(w, h)
    if proj["type"] == "orthographic":
        return (apw * (px - (w / 2)) / w, aph * ((-py) + (h / 2)) / h, 0.0), (0.0, 0.0, 1.0)
    f = 0.5 * h / (math.tan(0.5) * proj["fovy"])                        # sic: (tan 0.5) * fovy
    return (0.0, 0.0, 0.0), normalize((apw * (px - (w / 2)) / w, aph * ((-py) + (h / 2)) / h, f))


def ray_from_pixel(w, h, cam, px, py):
    o, d = unproject(w, h, cam["projection"], px, py)
    return vadd(o, cam["position"]), apply(look_at(cam["position"], cam["target"], cam["up"]), d)


# ----------------------------------------------------------------- src/RayHs.hs:58-188
def intersection(ray, obj):                                             # :58-62
    g = obj["geometry"]
    h = inter_plane(ray, g) if g["kind"] == "plane" else inter_sphere(ray, g) if g["kind"] == "sphere" else ray_inter(ray, g["tree"])
    return None if h is None else dict(h, material=obj["material"])


def intersections(objects, ray):                                        # :64-65
    return [h for h in (intersection(ray, o) for o in objects) if h is not None]


def closest_intersection(objects, ray):                                 # :67-71
    hits = intersections(objects, ray)
    return None if not hits else minimum_by_t(hits)


def shadow_intersection(objects, light, ray):                           # :74-87
    def in_front_of_light(h):
        if light["type"] == "directional":
            return True
        return sqr_dist(ray[0], light["position"]) > sqr_dist(ray[0], h["p"])
    hits = [h for h in intersections(objects, ray) if in_front_of_light(h) and h["material"]["type"] != "emmit"]
    return None if not hits else minimum_by_t(hits)


def accum_diffuse(objects, lights, p, n, color):                        # :89-97
    c = (0.0, 0.0, 0.0)
    for light in lights:
        ld, lc = light_at(light, p)
        inter = shadow_intersection(objects, light, ray_eps(p, ld))
        c = vadd(c, (0.0, 0.0, 0.0) if inter is not None else diffuse(color, lc, ld, n))
    return c


def specular(scene, depth, max_depth, v, p, n):                         # :99-104
    if depth < max_depth:
        return mul(dot(reflect(v, n), n), trace_ray(scene, depth + 1, max_depth, ray_eps(p, reflect(v, n))))
    return (0.0, 0.0, 0.0)


def irradiance(d, max_depth, scene, mat, v, p, n, uv):                  # :107-147
    objects, lights = scene
    k = mat["type"]
    if k == "diffuse":
        cd = color_at(mat["cd"], uv)
        return vadd(mul(0.2, cd), accum_diffuse(objects, lights, p, n, cd))
    if k == "plastic":
        cd = color_at(mat["cd"], uv)
        return vadd(accum_diffuse(objects, lights, p, n, cd),
                    mul(fresnel(mat["ior"], dot(n, vneg(v))), specular(scene, d, max_depth, v, p, n)))
    if k == "mirror":
        return mul(fresnel(mat["ior"], dot(n, vneg(v))), specular(scene, d, max_depth, v, p, n))
    if k == "emmit":
        return mat["ce"]
    if k == "transparent":
        ior = mat["ior"]
        radiance = None
        if d != max_depth:
            ref_dir = refract(v, n, 1.0, ior)
            if ref_dir is not None:
                out = closest_intersection(objects, ray_eps(p, ref_dir))
                if out is not None:
                    out_dir = refract(ref_dir, vneg(out["n"]), ior, 1.0)
                    if out_dir is not None:
                        radiance = trace_ray(scene, d + 1, max_depth, ray_eps(out["p"], out_dir))
        spec = mul(fresnel(ior, dot(n, vneg(v))), specular(scene, d, max_depth, v, p, n))
        return spec if radiance is None else vadd(mul(1 - r0(ior, 1.0), radiance), spec)
    raise ValueError(k)


def trace_ray(scene, depth, max_depth, ray):                            # :149-154
    inter = closest_intersection(scene[0], ray)
    if inter is None:
        return (0.0, 0.0, 0.0)
    return irradiance(depth, max_depth, scene, inter["material"], ray[1], inter["p"], inter["n"], inter["uv"])


def splitmix_sample01(seed, pixel, sample, dim, spp):
    """The BUILD-defined counter sampler (include/rayhs.h: rayhs_sample01) -- not the reference's StdGen stream,
    which is un-pinned third-party code."""
    m = (1 << 64) - 1

    def mix(x):
        x ^= x >> 30; x = (x * 0xbf58476d1ce4e5b9) & m
        x ^= x >> 27; x = (x * 0x94d049bb133111eb) & m
        x ^= x >> 31
        return x
    ctr = ((pixel * spp + sample) * 2 + dim) & m
    return float(mix((ctr + mix((seed + 0x9e3779b97f4a7c15) & m)) & m) >> 11) * (1.0 / 9007199254740992.0)


def trace_pixel(scene, d, w, h, cam, i, j, spp=1, seed=24, width=None):  # :156-159 and :169-188
    if spp == 1:
        return trace_ray(scene, 0, d, ray_from_pixel(w, h, cam, i, j))
    pixel = int(i) + width * int(j)
    total = (0.0, 0.0, 0.0)                                             # average: foldl (+) black, then mul (1/n)
    for s in range(spp):
        oi = splitmix_sample01(seed, pixel, s, 0, spp) - 0.5
        oj = splitmix_sample01(seed, pixel, s, 1, spp) - 0.5
        total = vadd(total, trace_ray(scene, 0, d, ray_from_pixel(w, h, cam, i + oi, j + oj)))
    return mul(1.0 / float(spp), total)


# ---------------------------------------------------- src/JSON.hs:22-141, src/Descriptors.hs:39-55
def _vec(d): return (float(d["x"]), float(d["y"]), float(d["z"]))
def _rgb(d): return (float(d["r"]), float(d["g"]), float(d["b"]))


def build_color_map(cd):
    k = cd["type"]
    if k == "flat":
        return {"type": k, "color": _rgb(cd["color"])}
    if k == "checker":
        return {"type": k, "color1": _rgb(cd["color1"]), "color2": _rgb(cd["color2"]), "size": float(cd["size"])}
    if k == "texture":
        return {"type": k, "bitmap": read_bitmap_ppm(cd["fileName"])}
    raise ValueError("Unknown type for color map " + k)


def build_material(m):
    k = m["type"]
    if k == "mirror":
        return {"type": k, "ior": float(m["ior"])}
    if k == "diffuse":
        return {"type": k, "cd": build_color_map(m["cd"])}
    if k == "plastic":
        return {"type": k, "cd": build_color_map(m["cd"]), "ior": float(m["ior"])}
    if k == "emmit":
        return {"type": k, "ce": _rgb(m["ce"])}
    if k == "transparent":
        return {"type": k, "ior": float(m["ior"])}
    raise ValueError("Unknown type for material " + k)


def build_geometry(g):
    k = g["type"]
    if k == "sphere":
        return {"kind": k, "center": _vec(g["center"]), "radius": float(g["radius"])}
    if k == "plane":
        # the parser fills PlaneDesc positionally (point, normal, tangent) into a record declared (normal, point,
        # tangent) and buildGeometry passes the fields on positionally: net effect Plane point normal tangent
        # (src/JSON.hs:115-118, src/Descriptors.hs:29-31,51)
        return {"kind": k, "point": _vec(g["point"]), "normal": _vec(g["normal"]), "tangent": _vec(g["tangent"])}
    if k == "mesh":
        verts, ids = read_obj(g["fileName"])
        tris = triangles(mesh_transform(verts, g["transform"]), ids)
        return {"kind": k, "tree": build_node(tris, 0, 0)}
    raise ValueError(k)


def build_light(l):
    if l["type"] == "directional":
        return {"type": "directional", "direction": _vec(l["direction"]), "color": _rgb(l["color"])}
    if l["type"] == "point":
        return {"type": "point", "position": _vec(l["position"]), "color": _rgb(l["color"]), "radius": float(l["radius"])}
    raise ValueError("Unknown type for light " + l["type"])


def build_rendering(jd):                                                # src/RayHs.hs:47-50
    sd = jd["scene"]
    objects = [{"geometry": build_geometry(o["geometry"]), "material": build_material(o["material"])} for o in sd["objects"]]
    lights = [build_light(l) for l in sd["lights"]]
    c = jd["camera"]
    p = c["projection"]
    proj = {"type": p["type"], "fovy": float(p["fovy"])} if p["type"] == "perspective" else {"type": p["type"]}
    cam = {"position": _vec(c["position"]), "target": _vec(c["target"]), "up": _vec(c["up"]), "projection": proj}
    return (objects, lights), cam, int(jd["width"]), int(jd["height"]), int(jd["maxDepth"])


def render(jd, width=None, height=None, spp=1, seed=24, rows=None, cols=None):
    """rayTrace (src/RayHs.hs:161-166) over pixelCoord (src/Image.hs:31-32); returns rows of (r, g, b).
    `rows` / `cols` restrict the pixels traced (the frame geometry stays width x height)."""
    scene, cam, w, h, d = build_rendering(jd)
    w, h = width or w, height or h
    fw, fh = float(w), float(h)
    out = []
    for y in (range(h) if rows is None else rows):
        out.append([trace_pixel(scene, d, fw, fh, cam, float(x), float(y), spp, seed, w) for x in (range(w) if cols is None else cols)])
    return out


def render_file(path, **kw):
    return render(json.load(open(path)), **kw)
