"""ctypes front-end of the CPU parity oracle (oracle/rayhs_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product (rayhs_amd/).  PARITY
UNPINNED (see rayhs_oracle.h).

The scene JSON is decoded here with Python's json module, following the aeson
instances of the reference (src/JSON.hs:22-141) -- an independent route from
the product's C++ parser, so the two check each other.
"""
import ctypes as C
import json
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Material(C.Structure):
    _fields_ = [("kind", C.c_int), ("ior", C.c_double), ("ce", C.c_double * 3),
                ("cmap_kind", C.c_int), ("c1", C.c_double * 3), ("c2", C.c_double * 3),
                ("size", C.c_double), ("tex_w", C.c_int), ("tex_h", C.c_int),
                ("tex", C.POINTER(C.c_double))]


class Xform(C.Structure):
    _fields_ = [("kind", C.c_int), ("v", C.c_double * 3), ("angle", C.c_double)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("target", C.c_double * 3), ("up", C.c_double * 3),
                ("proj_kind", C.c_int), ("fovy", C.c_double), ("pw", C.c_double),
                ("ph", C.c_double), ("near_", C.c_double)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("rays_primary", "rays_reflect", "rays_refract", "rays_refract_exit", "rays_shadow",
                 "nodes_tested", "tris_tested", "prims_tested", "shaded_hits")]

    def as_dict(self):
        d = {n: int(getattr(self, n)) for n, _ in self._fields_}
        d["rays"] = sum(d[k] for k in d if k.startswith("rays_"))
        return d


class Hit(C.Structure):
    _fields_ = [("hit", C.c_int), ("object", C.c_int), ("t", C.c_double), ("p", C.c_double * 3),
                ("n", C.c_double * 3), ("uv", C.c_double * 2)]


class TreeStats(C.Structure):
    _fields_ = [("tris", C.c_int), ("inner", C.c_int), ("leaves", C.c_int), ("empties", C.c_int),
                ("max_depth", C.c_int), ("leaf_min", C.c_int), ("leaf_max", C.c_int),
                ("leaf_mean", C.c_double)]


MAT = {"mirror": 0, "diffuse": 1, "plastic": 2, "emmit": 3, "transparent": 4}
XF = {"translate": 0, "scale": 1, "rotate": 2, "rotateX": 3, "rotateY": 4, "rotateZ": 5}


def build(force=False):
    """Compile librayhs_oracle.so with the committed Makefile (gcc, -ffp-contract=off)."""
    so = os.path.join(_HERE, "librayhs_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("rayhs_oracle.c", "rayhs_oracle.h", "Makefile")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_scene_new.restype = C.c_void_p
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_last_error.restype = C.c_char_p
        d3 = C.c_double * 3
        L.orc_add_plane.argtypes = [C.c_void_p, d3, d3, d3, C.POINTER(Material)]
        L.orc_add_sphere.argtypes = [C.c_void_p, d3, C.c_double, C.POINTER(Material)]
        L.orc_add_mesh_obj.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(Xform),
                                       C.POINTER(Material)]
        L.orc_add_light_directional.argtypes = [C.c_void_p, d3, d3]
        L.orc_add_light_point.argtypes = [C.c_void_p, d3, d3, C.c_double]
        L.orc_num_objects.argtypes = [C.c_void_p]
        L.orc_mesh_tree_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(TreeStats)]
        L.orc_mesh_triangles.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(Camera), C.c_int, C.c_int, C.c_int, C.c_int,
                                 C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.POINTER(Stats), C.c_int]
        L.orc_ray_from_pixel.argtypes = [C.POINTER(Camera), C.c_double, C.c_double, C.c_double,
                                         C.c_double, d3, d3]
        L.orc_closest_hit.argtypes = [C.c_void_p, d3, d3, C.POINTER(Hit)]
        L.orc_closest_hit_bruteforce.argtypes = [C.c_void_p, d3, d3, C.POINTER(Hit)]
        L.orc_trace_ray.argtypes = [C.c_void_p, d3, d3, C.c_int, C.c_int, d3]
        L.orc_r0.restype = C.c_double
        L.orc_r0.argtypes = [C.c_double, C.c_double]
        L.orc_fresnel.restype = C.c_double
        L.orc_fresnel.argtypes = [C.c_double, C.c_double]
        L.orc_set_pow5_variant.argtypes = [C.c_int]
        L.orc_mod_floor.restype = C.c_double
        L.orc_mod_floor.argtypes = [C.c_double, C.c_double]
        L.orc_refract.argtypes = [d3, d3, C.c_double, C.c_double, d3]
        L.orc_color_at.argtypes = [C.POINTER(Material), C.c_double, C.c_double, d3]
        L.orc_sample01.restype = C.c_double
        L.orc_sample01.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.orc_to_int_c.argtypes = [C.c_double]
        L.orc_set_shadow_exhaustive.argtypes = [C.c_int]
        L.orc_write_ppm.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        _LIB = L
    return _LIB


def _v3(d, keys=("x", "y", "z")):
    return (C.c_double * 3)(*(float(d[k]) for k in keys))


def _rgb(d):
    return _v3(d, ("r", "g", "b"))


def read_texture_ppm(path):
    """Bitmap.readPPM (src/Bitmap.hs:31-37): 3 header lines, then numbers / 255."""
    with open(path) as f:
        lines = f.read().split("\n")
    w, h = (int(x) for x in lines[1].split()[:2])
    vals = np.array([float(t) for ln in lines[3:] for t in ln.split()], dtype=np.float64) / 255
    n = (len(vals) // 3) * 3
    return w, h, np.ascontiguousarray(vals[:n])


def make_material(m, base_dir="."):
    """MaterialDesc / ColorMapDesc (src/JSON.hs:40-61, src/MaterialDescriptors.hs:27-45)."""
    out = Material()
    keep = None
    kind = m["type"]
    out.kind = MAT[kind]
    if kind in ("mirror", "plastic", "transparent"):
        out.ior = float(m["ior"])
    if kind == "emmit":
        out.ce = _rgb(m["ce"])
    if kind in ("diffuse", "plastic"):
        cd = m["cd"]
        if cd["type"] == "flat":
            out.cmap_kind = 0
            out.c1 = _rgb(cd["color"])
        elif cd["type"] == "checker":
            out.cmap_kind = 1
            out.c1 = _rgb(cd["color1"])
            out.c2 = _rgb(cd["color2"])
            out.size = float(cd["size"])
        elif cd["type"] == "texture":
            out.cmap_kind = 2
            w, h, px = read_texture_ppm(os.path.join(base_dir, cd["fileName"]))
            out.tex_w, out.tex_h = w, h
            keep = px
            out.tex = px.ctypes.data_as(C.POINTER(C.c_double))
        else:
            raise ValueError("Unknown type for color map " + cd["type"])
    return out, keep


def flatten_transform(t):
    """Transform (src/JSON.hs:86-97). Nested Sequences fold in list order
    (src/Transform.hs:24), so flattening preserves the result."""
    kind = t["type"]
    if kind == "sequence":
        out = []
        for s in t["transforms"]:
            out.extend(flatten_transform(s))
        return out
    x = Xform()
    x.kind = XF[kind]
    if kind in ("translate", "scale"):
        x.v = _v3(t["vector"])
    elif kind == "rotate":
        x.v = _v3(t["axis"])
        x.angle = float(t["angle"])
    else:
        x.angle = float(t["angle"])
    return [x]


def make_camera(c):
    """Camera / Projection (src/JSON.hs:74-106)."""
    cam = Camera()
    cam.pos = _v3(c["position"])
    cam.target = _v3(c["target"])
    cam.up = _v3(c["up"])
    p = c["projection"]
    if p["type"] == "perspective":
        cam.proj_kind = 1
        cam.fovy = float(p["fovy"])
        cam.near_ = float(p["near"])
    elif p["type"] == "orthographic":
        cam.proj_kind = 0
    else:
        raise ValueError("Unknown type for projection " + p["type"])
    cam.pw = float(p["width"])
    cam.ph = float(p["height"])
    return cam


class Scene:
    """A built Scene (src/Descriptors.hs:39-55) held by the C oracle."""

    def __init__(self, scene_desc, base_dir="."):
        L = lib()
        self._h = C.c_void_p(L.orc_scene_new())
        self._keep = []
        for o in scene_desc["objects"]:
            g = o["geometry"]
            mat, keep = make_material(o["material"], base_dir)
            self._keep.append(keep)
            if g["type"] == "sphere":
                L.orc_add_sphere(self._h, _v3(g["center"]), float(g["radius"]), C.byref(mat))
            elif g["type"] == "plane":
                L.orc_add_plane(self._h, _v3(g["point"]), _v3(g["normal"]), _v3(g["tangent"]), C.byref(mat))
            elif g["type"] == "mesh":
                xf = flatten_transform(g["transform"])
                arr = (Xform * max(1, len(xf)))(*xf)
                top = 1 if g["transform"]["type"] == "translate" else 0
                path = os.path.join(base_dir, g["fileName"]).encode()
                if L.orc_add_mesh_obj(self._h, path, top, len(xf), arr, C.byref(mat)) != 0:
                    raise RuntimeError(L.orc_last_error().decode())
            else:
                raise ValueError("unknown geometry type " + g["type"])
        for l in scene_desc["lights"]:
            if l["type"] == "directional":
                L.orc_add_light_directional(self._h, _v3(l["direction"]), _rgb(l["color"]))
            elif l["type"] == "point":
                L.orc_add_light_point(self._h, _v3(l["position"]), _rgb(l["color"]), float(l["radius"]))
            else:
                raise ValueError("Unknown type for light " + l["type"])

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scene_free(self._h)
            self._h = None

    def tree_stats(self, obj):
        ts = TreeStats()
        if lib().orc_mesh_tree_stats(self._h, obj, C.byref(ts)) != 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return {n: getattr(ts, n) for n, _ in ts._fields_}

    def mesh_triangles(self, obj):
        n = lib().orc_mesh_triangles(self._h, obj, None, 0)
        out = np.empty((n, 3, 8), dtype=np.float64)
        lib().orc_mesh_triangles(self._h, obj, out.ctypes.data, n)
        return out

    def closest_hit(self, o, d, brute=False):
        h = Hit()
        f = lib().orc_closest_hit_bruteforce if brute else lib().orc_closest_hit
        f(self._h, (C.c_double * 3)(*o), (C.c_double * 3)(*d), C.byref(h))
        return h

    def trace_ray(self, o, d, depth, max_depth):
        out = (C.c_double * 3)()
        lib().orc_trace_ray(self._h, (C.c_double * 3)(*o), (C.c_double * 3)(*d), depth, max_depth, out)
        return np.array(out[:])

    def render(self, cam, w, h, max_depth, spp=1, seed=24, row0=0, row1=None, nthreads=None):
        row1 = h if row1 is None else row1
        nthreads = nthreads or os.cpu_count() or 1
        out = np.empty((row1 - row0, w, 3), dtype=np.float64)
        st = Stats()
        if lib().orc_render(self._h, C.byref(cam), w, h, max_depth, spp, seed, row0, row1,
                            out.ctypes.data, C.byref(st), nthreads) != 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return out, st.as_dict()


class pow5_variant:
    """with pow5_variant(True): ... renders with fresnel's x ^ (5 :: Int) as square-and-multiply (GHC.Real's (^) un-rewritten)
    instead of the default flat product (GHC -O RULES "^5/Int"); the kernels' RAYHS_FLAG_POW5_SQMUL is the same switch."""

    def __init__(self, square_and_multiply):
        self.flag = 1 if square_and_multiply else 0

    def __enter__(self):
        lib().orc_set_pow5_variant(self.flag)

    def __exit__(self, *exc):
        lib().orc_set_pow5_variant(0)


def load_render_desc(path):
    """RenderDesc (src/JSON.hs:135-141)."""
    with open(path) as f:
        return json.load(f)


def render_file(path, base_dir=".", width=None, height=None, max_depth=None, mesh_file=None, **kw):
    """Render a scene JSON; width/height/max_depth/mesh_file override the file's values
    (the reference has no CLI override)."""
    jd = load_render_desc(path)
    if mesh_file is not None:
        for o in jd["scene"]["objects"]:
            if o["geometry"]["type"] == "mesh":
                o["geometry"]["fileName"] = mesh_file
    sc = Scene(jd["scene"], base_dir)
    cam = make_camera(jd["camera"])
    w = width or jd["width"]
    h = height or jd["height"]
    d = jd["maxDepth"] if max_depth is None else max_depth
    img, st = sc.render(cam, w, h, d, **kw)
    return img, st


def ray_from_pixel(cam, w, h, px, py):
    o = (C.c_double * 3)()
    d = (C.c_double * 3)()
    lib().orc_ray_from_pixel(C.byref(cam), w, h, px, py, o, d)
    return np.array(o[:]), np.array(d[:])


def write_ppm(path, img):
    img = np.ascontiguousarray(img, dtype=np.float64)
    h, w, _ = img.shape
    if lib().orc_write_ppm(path.encode(), w, h, img.ctypes.data) != 0:
        raise RuntimeError(lib().orc_last_error().decode())
