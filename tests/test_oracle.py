"""CPU tests of the parity oracle (oracle/rayhs_oracle.c).

The reference has no tests, golden vectors or expected images, so
the oracle is PARITY UNPINNED.  What can be checked is checked here: analytic
values worked out by hand from the reference's formulas (file:line cited per
test), agreement of the tree with brute force (src/Mesh.hs:44-45 vs
src/KDTree.hs:96-107), the tree statistics table of SURVEY.md section 8, and the
restatement-derived fixtures in tests/golden/.
"""
import ctypes as C
import json
import math
import os

import numpy as np
import pytest

from oracle import oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def d3(*v):
    return (C.c_double * 3)(*v)


def cam_default(fovy=0.9272952180016123, pos=(0, 0, -2), target=(0, 0, 0), persp=True):
    return O.make_camera({"position": dict(zip("xyz", pos)), "target": dict(zip("xyz", target)),
                          "up": {"x": 0, "y": 1, "z": 0},
                          "projection": {"type": "perspective" if persp else "orthographic", "fovy": fovy,
                                         "width": 2, "height": 2, "near": 0.1}})


# ---- Material.hs:22-29
def test_r0_known_values():
    L = O.lib()
    assert L.orc_r0(1.0, 1.5) == pytest.approx(0.04, abs=1e-17)
    assert L.orc_r0(1.0, 1.9) == ((1.0 - 1.9) / (1.0 + 1.9)) ** 2
    assert L.orc_r0(1.0, 1.0) == 0.0


def test_fresnel_schlick_and_power_chain():
    L = O.lib()
    r = L.orc_r0(1.0, 1.9)
    assert L.orc_fresnel(1.9, 1.0) == r                      # (1 - 1)^5 = 0
    assert L.orc_fresnel(1.9, 0.0) == r + (1 - r) * 1.0      # grazing: 1
    x = 1 - 0.3
    assert L.orc_fresnel(1.9, 0.3) == r + (1 - r) * ((((x * x) * x) * x) * x)  # GHC -O: RULES "^5/Int" -> u*u*u*u*u
    assert L.orc_fresnel(1.9, -0.5) > 1.0                    # cos0 < 0 exceeds 1


def test_both_readings_of_the_fifth_power_against_hand_computed_vectors():
    # tests/golden/pow5_variants.json (plain float arithmetic, make_pow5_variants.py): arguments on which the flat product
    # x*x*x*x*x (GHC -O RULES "^5/Int", the default) and square-and-multiply ((x*x)*(x*x))*x (GHC.Real's (^)) differ.
    # Both restatements switch together, so that one value from a real GHC -O2 build can settle the reading later
    # without touching the kernels' sources (RAYHS_FLAG_POW5_SQMUL is the same switch on the GPU).
    import json
    from oracle.second import rayhs_second as S2
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "pow5_variants.json")))
    assert gold["default"] == "flat" and len(gold["vectors"]) >= 16
    L = O.lib()
    for v in gold["vectors"]:
        ior, c = float.fromhex(v["ior"]), float.fromhex(v["cos0"])
        flat, sqmul = float.fromhex(v["flat"]), float.fromhex(v["sqmul"])
        assert flat != sqmul
        assert L.orc_fresnel(ior, c) == flat and S2.fresnel(ior, c) == flat
        with O.pow5_variant(True):
            assert L.orc_fresnel(ior, c) == sqmul
        S2.POW5_SQUARE_AND_MULTIPLY = True
        try:
            assert S2.fresnel(ior, c) == sqmul
        finally:
            S2.POW5_SQUARE_AND_MULTIPLY = False
        assert L.orc_fresnel(ior, c) == flat          # the switch is back at the default


# ---- Vec.hs:128-140
def test_refract_straight_through_and_total_internal_reflection():
    L = O.lib()
    out = d3(0, 0, 0)
    assert L.orc_refract(d3(0, 0, 1), d3(0, 0, -1), 1.0, 1.5, out) == 1
    assert list(out) == pytest.approx([0, 0, 1])
    s = math.sin(math.radians(60))
    c = math.cos(math.radians(60))
    assert L.orc_refract(d3(s, 0, c), d3(0, 0, -1), 1.5, 1.0, out) == 0      # sin^2 = 2.25*0.75 > 1
    assert L.orc_refract(d3(s, 0, c), d3(0, 0, -1), 1.0, 1.5, out) == 1
    sin_t = s / 1.5                                                          # Snell
    assert out[0] == pytest.approx(sin_t, rel=1e-14)
    assert out[2] == pytest.approx(math.sqrt(1 - sin_t ** 2), rel=1e-14)


# ---- Data.Fixed.mod' (ColorMap.hs:21-24, 57-58)
def test_mod_floor_is_floor_modulo():
    L = O.lib()
    assert L.orc_mod_floor(0.75, 0.5) == 0.25
    assert L.orc_mod_floor(-0.25, 0.5) == 0.25
    assert L.orc_mod_floor(-0.5, 0.5) == 0.0
    assert L.orc_mod_floor(2.0, 1.0) == 0.0
    # the exact quotient 0.3/0.1 is just below 3, so floor is 2 although 0.3/0.1 rounds to 3.0
    from fractions import Fraction
    n, d = 0.3, 0.1
    f = math.floor(Fraction(n) / Fraction(d))
    assert f == 2
    assert L.orc_mod_floor(n, d) == n - f * d
    rng = np.random.default_rng(1)
    for n, d in zip(rng.uniform(-50, 50, 300), rng.uniform(0.01, 3, 300)):
        f = math.floor(Fraction(float(n)) / Fraction(float(d)))
        assert L.orc_mod_floor(float(n), float(d)) == float(n) - float(f) * float(d)


def test_checker_colour_map():
    m = O.Material()
    m.kind, m.cmap_kind = 1, 1
    m.c1 = d3(1, 0, 0)
    m.c2 = d3(0, 1, 0)
    m.size = 0.5
    out = d3(0, 0, 0)
    L = O.lib()
    L.orc_color_at(C.byref(m), 0.1, 0.3, out)      # (0.1-0.25)*(0.3-0.25) < 0 -> color1
    assert list(out) == [1, 0, 0]
    L.orc_color_at(C.byref(m), 0.1, 0.1, out)      # both negative -> product > 0 -> color2
    assert list(out) == [0, 1, 0]
    L.orc_color_at(C.byref(m), -0.1, 0.1, out)     # mod' -0.1 0.5 = 0.4 -> (0.15)*(-0.15) < 0
    assert list(out) == [1, 0, 0]


def test_texture_bilinear_with_bankers_rounding_and_wrap():
    # 2x2 texture; ColorMap.hs:26-39
    tex = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=np.float64)   # (x,y): (0,0),(1,0),(0,1),(1,1)
    m = O.Material()
    m.kind, m.cmap_kind, m.tex_w, m.tex_h = 1, 2, 2, 2
    m.tex = tex.ctypes.data_as(C.POINTER(C.c_double))
    out = d3(0, 0, 0)
    # uv=(0.25,0.25): u=0.5 -> round 0 (ties to even) -> x0=(-1 mod 2)=1, x1=0, lx = 0.5-(-1)-0.5 = 1.0 -> all weight on x1,y1 = texel (0,0)
    O.lib().orc_color_at(C.byref(m), 0.25, 0.25, out)
    assert list(out) == [0, 0, 0]
    # uv=(0.75,0.25): u=1.5 -> round 2 -> x0=1, x1=0, lx = 1.5-1-0.5 = 0 -> texel (x0=1, y1=0) = red
    O.lib().orc_color_at(C.byref(m), 0.75, 0.25, out)
    assert list(out) == [1, 0, 0]
    # uv=(0.5,0.5): u=1 -> round 1, x0=0,x1=1, lx=0.5; v likewise -> mean of the four texels
    O.lib().orc_color_at(C.byref(m), 0.5, 0.5, out)
    assert list(out) == [0.25, 0.25, 0.25]


# ---- Projection.hs:22-46, Mat.hs:89-93
def test_centre_pixel_looks_along_z():
    cam = cam_default()
    o, d = O.ray_from_pixel(cam, 512.0, 512.0, 256.0, 256.0)
    assert list(o) == [0, 0, -2]
    assert list(d) == [0, 0, 1]


def test_focal_length_quirk_tan_half_times_fovy():
    # f = 0.5*h/((tan 0.5)*fovy), NOT 0.5*h/tan(0.5*fovy)  (Projection.hs:35)
    fovy = 0.9272952180016123
    cam = cam_default(fovy)
    w = h = 512.0
    _, d = O.ray_from_pixel(cam, w, h, 0.0, 256.0)          # px=0 -> x = 512*(0-256)/512 = -256, y = 0
    f = 0.5 * h / (math.tan(0.5) * fovy)
    assert f == pytest.approx(0.98700 * h, rel=1e-4)
    n = math.sqrt(256.0 ** 2 + f ** 2)
    assert d[0] == pytest.approx(-256.0 / n, rel=1e-15)
    assert d[2] == pytest.approx(f / n, rel=1e-15)
    assert d[1] == 0.0


def test_pixel_y_is_flipped_and_aspect_uses_width():
    cam = cam_default()
    _, d_top = O.ray_from_pixel(cam, 1920.0, 1080.0, 960.0, 0.0)
    assert d_top[1] > 0                                       # row 0 is the top (Projection.hs:37)
    # aspect > 1: (apw, aph) = (w, w/aspect) = (1920, 1080): x = 1920*(px-960)/1920
    _, d_left = O.ray_from_pixel(cam, 1920.0, 1080.0, 0.0, 540.0)
    f = 0.5 * 1080.0 / (math.tan(0.5) * 0.9272952180016123)
    assert d_left[0] / d_left[2] == pytest.approx(-960.0 / f, rel=1e-14)


def test_orthographic_origin_is_not_rotated():
    cam = cam_default(persp=False, pos=(1, 2, 3), target=(1, 2, 4))
    o, d = O.ray_from_pixel(cam, 100.0, 100.0, 75.0, 25.0)
    assert list(d) == [0, 0, 1]
    assert list(o) == [1 + 100 * (75 - 50) / 100, 2 + 100 * (-25 + 50) / 100, 3]


# ---- Geometry.hs:68-96
def _scene(objs, lights=()):
    return O.Scene({"objects": list(objs), "lights": list(lights)})


MIRROR = {"type": "mirror", "ior": 1.5}


def test_sphere_hit_front_and_from_inside():
    sc = _scene([{"geometry": {"type": "sphere", "center": {"x": 0, "y": 0, "z": 5}, "radius": 1}, "material": MIRROR}])
    h = sc.closest_hit((0, 0, 0), (0, 0, 1))
    assert h.hit and h.t == 4.0 and list(h.p) == [0, 0, 4] and list(h.n) == [0, 0, -1]
    h = sc.closest_hit((0, 0, 5), (0, 0, 2))                  # inside, non-unit direction: t1 = 0.5
    assert h.hit and h.t == 0.5 and list(h.n) == [0, 0, 1]
    assert not sc.closest_hit((0, 0, 0), (0, 1, 0)).hit
    assert not sc.closest_hit((0, 0, 7), (0, 0, 1)).hit       # both roots negative


def test_sphere_polar_uv():
    sc = _scene([{"geometry": {"type": "sphere", "center": {"x": 0, "y": 0, "z": 0}, "radius": 1}, "material": MIRROR}])
    h = sc.closest_hit((3, 0, 0), (-1, 0, 0))                 # n = (1,0,0): u = atan(0/1)/pi = 0, v = acos(0)/pi = 0.5
    assert h.uv[0] == 0.0 and h.uv[1] == pytest.approx(0.5, abs=1e-16)


def test_plane_hit_uv_and_unflipped_normal():
    pl = {"geometry": {"type": "plane", "point": {"x": 0, "y": -1, "z": 0}, "normal": {"x": 0, "y": 1, "z": 0},
                       "tangent": {"x": 0, "y": 0, "z": 1}}, "material": MIRROR}
    sc = _scene([pl])
    h = sc.closest_hit((0.5, 1, 0.25), (0, -1, 0))
    assert h.hit and h.t == 2.0 and list(h.n) == [0, 1, 0]
    # u = t.rel = 0.25; b = t x n = (0,0,1)x(0,1,0) = (-1,0,0); v = b.rel = -0.5
    assert list(h.uv) == [0.25, -0.5]
    h = sc.closest_hit((0, -3, 0), (0, 1, 0))                 # from below: still hit, normal NOT flipped
    assert h.hit and list(h.n) == [0, 1, 0]
    assert not sc.closest_hit((0, 1, 0), (1, 0, 0)).hit       # parallel: |d.n| = 0


def test_first_object_wins_exact_ties():
    # two identical planes: minimumBy keeps the first (RayHs.hs:67-71)
    pl = {"geometry": {"type": "plane", "point": {"x": 0, "y": 0, "z": 1}, "normal": {"x": 0, "y": 0, "z": -1},
                       "tangent": {"x": 1, "y": 0, "z": 0}}, "material": MIRROR}
    sc = _scene([pl, pl])
    assert sc.closest_hit((0, 0, 0), (0, 0, 1)).object == 0


# ---- Mesh.hs:59-82
def _write_obj(tmp_path, text, name="m.obj"):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


IDENT = {"type": "translate", "vector": {"x": 0, "y": 0, "z": 0}}


def test_triangle_is_two_sided_and_normal_is_not_normalised(tmp_path):
    obj = _write_obj(tmp_path, "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 2\nf 1//1 2//1 3//1\n")
    sc = _scene([{"geometry": {"type": "mesh", "fileName": obj, "transform": IDENT}, "material": MIRROR}])
    for oz, dz in ((1.0, -1.0), (-1.0, 1.0)):
        h = sc.closest_hit((0.25, 0.25, oz), (0, 0, dz))
        assert h.hit and h.t == 1.0 and list(h.n) == [0, 0, 2]        # u*n1 + v*n2 + (1-u-v)*n0, unnormalised
    assert not sc.closest_hit((0.75, 0.75, 1), (0, 0, -1)).hit        # u+v > 1
    assert not sc.closest_hit((0.25, 0.25, 5e-7), (0, 0, -1)).hit     # t < eps = 1e-6


def test_obj_quad_is_not_fanned_and_missing_attributes_default(tmp_path):
    # 4 + 5 indices are concatenated and regrouped by 3 (Mesh.hs:183, 105-109)
    obj = _write_obj(tmp_path, "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nv 0 0 1\nf 1 2 3 4\nf 1 2 3 4 5\n")
    sc = _scene([{"geometry": {"type": "mesh", "fileName": obj, "transform": IDENT}, "material": MIRROR}])
    tri = sc.mesh_triangles(0)
    assert tri.shape == (3, 3, 8)
    idx = [[int(round(v[0] + 2 * v[1] + 4 * v[2])) for v in t[:, :3]] for t in tri]   # encode positions
    enc = {0: 1, 1: 2, 3: 3, 2: 4, 4: 5}
    assert [[enc[i] for i in t] for t in idx] == [[1, 2, 3], [4, 1, 2], [3, 4, 5]]
    assert np.all(tri[:, :, 3:] == 0)                                   # no vn / vt -> zeros


def test_obj_dangling_indices_are_an_error(tmp_path):
    obj = _write_obj(tmp_path, "v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")
    with pytest.raises(RuntimeError, match="faces"):
        _scene([{"geometry": {"type": "mesh", "fileName": obj, "transform": IDENT}, "material": MIRROR}])


def test_mesh_transform_translates_normals_unless_bare_translate(tmp_path):
    obj = _write_obj(tmp_path, "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    t = {"type": "translate", "vector": {"x": 1, "y": 2, "z": 3}}
    bare = _scene([{"geometry": {"type": "mesh", "fileName": obj, "transform": t}, "material": MIRROR}]).mesh_triangles(0)
    seq = _scene([{"geometry": {"type": "mesh", "fileName": obj, "transform": {"type": "sequence", "transforms": [t]}},
                   "material": MIRROR}]).mesh_triangles(0)
    assert list(bare[0, 0, :6]) == [1, 2, 3, 0, 0, 1]
    assert list(seq[0, 0, :6]) == [1, 2, 3, 1, 2, 4]                    # Mesh.hs:96-101: normal moved like a point


# ---- KDTree.hs
SURVEY_TREES = {  # tris, inner, leaves, Empty, max depth, leaf min/max
    "cube": (12, 0, 1, 0, 0, 12, 12), "cone": (62, 5, 4, 2, 3, 14, 17), "torus": (576, 38, 39, 0, 6, 8, 19),
    "dragon_superlow": (3402, 279, 280, 0, 11, 3, 19), "dragon_low": (6807, 547, 548, 0, 12, 1, 19),
    "dragon": (27228, 2199, 2200, 0, 16, 1, 19),
}


@pytest.mark.parametrize("mesh", sorted(SURVEY_TREES))
def test_tree_statistics_match_survey_table(mesh):
    sc = _scene([{"geometry": {"type": "mesh", "fileName": f"data/{mesh}.obj", "transform": IDENT}, "material": MIRROR}])
    s = sc.tree_stats(0)
    assert (s["tris"], s["inner"], s["leaves"], s["empties"], s["max_depth"], s["leaf_min"], s["leaf_max"]) == SURVEY_TREES[mesh]


@pytest.mark.parametrize("mesh", ["torus", "cone", "dragon_superlow"])
def test_tree_traversal_agrees_with_brute_force(mesh):
    sc = _scene([{"geometry": {"type": "mesh", "fileName": f"data/{mesh}.obj", "transform": IDENT}, "material": MIRROR}])
    rng = np.random.default_rng(7)
    hits = 0
    for _ in range(400):
        o = rng.normal(size=3)
        o = 3 * o / np.linalg.norm(o)
        tgt = rng.uniform(-0.4, 0.4, size=3)
        d = tgt - o
        a, b = sc.closest_hit(o, d), sc.closest_hit(o, d, brute=True)
        assert a.hit == b.hit
        if a.hit:
            hits += 1
            assert a.t == b.t                    # same closest t
    assert hits > 50


# ---- Light.hs / RayHs.hs shading
def test_diffuse_plane_under_point_light_by_hand():
    # one plane z=1 facing the camera, one point light at the origin side
    pl = {"geometry": {"type": "plane", "point": {"x": 0, "y": 0, "z": 1}, "normal": {"x": 0, "y": 0, "z": -1},
                       "tangent": {"x": 1, "y": 0, "z": 0}},
          "material": {"type": "diffuse", "cd": {"type": "flat", "color": {"r": 0.5, "g": 1, "b": 2}}}}
    light = {"type": "point", "position": {"x": 0, "y": 0, "z": 0}, "color": {"r": 10, "g": 10, "b": 10}, "radius": 0.5}
    sc = _scene([pl], [light])
    rgb = sc.trace_ray((0, 0, 0), (0, 0, 1), 0, 3)
    d = 1.0
    falloff = 1.0 / (1.0 + d / 0.5) ** 2              # Light.hs:16
    lam = max(1.0, 0) * (1 / math.pi)                # l.n = (0,0,-1).(0,0,-1) = 1 (Material.hs:31-33)
    want = [0.2 * c + lam * (c * (falloff * 10)) for c in (0.5, 1, 2)]   # 0.2*cd + sum (RayHs.hs:111-114)
    assert list(rgb) == pytest.approx(want, rel=1e-15)


def test_directional_light_is_not_negated_or_normalised_and_emitters_do_not_occlude():
    pl = {"geometry": {"type": "plane", "point": {"x": 0, "y": 0, "z": 1}, "normal": {"x": 0, "y": 0, "z": -1},
                       "tangent": {"x": 1, "y": 0, "z": 0}},
          "material": {"type": "diffuse", "cd": {"type": "flat", "color": {"r": 1, "g": 1, "b": 1}}}}
    em = {"geometry": {"type": "sphere", "center": {"x": 0, "y": 0, "z": 0.5}, "radius": 0.1},
          "material": {"type": "emmit", "ce": {"r": 1, "g": 1, "b": 1}}}
    blocker = {"geometry": {"type": "sphere", "center": {"x": 0.3, "y": 0, "z": 0.5}, "radius": 0.1}, "material": MIRROR}
    light = {"type": "directional", "direction": {"x": 0, "y": 0, "z": -2}, "color": {"r": 1, "g": 1, "b": 1}}
    sc = _scene([pl, em, blocker], [light])
    # hit at (-0.3,0,1): shadow ray goes along (0,0,-2); nothing is in the way
    rgb = sc.trace_ray((-0.3, 0, 0.9), (0, 0, 1), 0, 0)
    assert rgb[0] == pytest.approx(0.2 + 2.0 / math.pi, rel=1e-15)       # l.n = 2 (unnormalised), Light.hs:14
    # under the emitter: it is hit by the shadow ray but Emmit is never an occluder (RayHs.hs:81)
    rgb = sc.trace_ray((0.0, 0.2, 0.9), (0, -2, 1), 0, 0)
    assert rgb[0] == pytest.approx(0.2 + 2.0 / math.pi, rel=1e-15)
    # under the mirror sphere: occluded -> ambient only
    rgb = sc.trace_ray((0.3, 0, 0.9), (0, 0, 1), 0, 0)
    assert rgb[0] == pytest.approx(0.2, rel=1e-15)


def test_max_depth_zero_disables_specular_and_transmission():
    sph = {"geometry": {"type": "sphere", "center": {"x": 0, "y": 0, "z": 3}, "radius": 1},
           "material": {"type": "transparent", "ior": 1.5}}
    sc = _scene([sph])
    assert list(sc.trace_ray((0, 0, 0), (0, 0, 1), 0, 0)) == [0, 0, 0]   # RayHs.hs:104,137


# ---- Image.hs:54-75
def test_to_int_c_truncates_clamps_high_and_maps_nan_to_255():
    L = O.lib()
    assert [L.orc_to_int_c(v) for v in (0.0, 0.999, 1.0, 7.5, 0.5, -0.001, -0.5)] == [0, 254, 255, 255, 127, 0, -127]
    assert L.orc_to_int_c(float("nan")) == 255


def test_ppm_byte_layout(tmp_path):
    img = np.array([[[0, 0.5, 1], [2, -0.5, 0.25]], [[1, 1, 1], [0, 0, 0]]], dtype=np.float64)
    p = tmp_path / "o.ppm"
    O.write_ppm(str(p), img)
    assert p.read_bytes() == b"P3\n2 2\n255\n0 127 255  255 -127 63  \n255 255 255  0 0 0  "


# ---- sampler + fixtures
def test_sampler_known_answers_and_range():
    meta = json.load(open(os.path.join(GOLD, "golden.json")))
    for seed, pixel, sample, dim, spp, hexv in meta["sampler"]:
        assert O.lib().orc_sample01(seed, pixel, sample, dim, spp) == float.fromhex(hexv)
    v = np.array([O.lib().orc_sample01(24, p, s, d, 8) for p in range(200) for s in range(8) for d in range(2)])
    assert v.min() >= 0 and v.max() < 1 and abs(v.mean() - 0.5) < 0.02


@pytest.mark.parametrize("name", ["cornell_96", "cornell_256", "dragon_superlow_96", "dragon_superlow_96_spp4", "outscene_128x72",
                                  "transform_64", "texture_earth_96"])
def test_oracle_reproduces_golden_fixture(name):
    meta = json.load(open(os.path.join(GOLD, "golden.json")))[name]
    img, st = O.render_file(meta["scene"], spp=meta["spp"], seed=meta["seed"])
    assert np.array_equal(img, np.load(os.path.join(GOLD, name + ".npy")))
    assert st == meta["stats"]


def test_single_thread_equals_multi_thread_and_row_ranges():
    a, _ = O.render_file("scenes/cornell_96.json", nthreads=1)
    b, _ = O.render_file("scenes/cornell_96.json", nthreads=4)
    c, _ = O.render_file("scenes/cornell_96.json", row0=40, row1=50)
    assert np.array_equal(a, b) and np.array_equal(a[40:50], c)


# ---- whole pixels derived by hand from the Haskell formulas (exact equality: same operations, same order)
def _render_text(jd, **kw):
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(jd, f)
    try:
        return O.render_file(f.name, **kw)[0]
    finally:
        os.unlink(f.name)


def _camera_json(fovy=0.9272952180016123):
    return {"position": {"x": 0, "y": 0, "z": -2}, "target": {"x": 0, "y": 0, "z": 0}, "up": {"x": 0, "y": 1, "z": 0},
            "projection": {"type": "perspective", "fovy": fovy, "width": 2, "height": 2, "near": 0.1}}


def test_centre_pixel_of_a_plastic_sphere_under_a_point_light_by_hand():
    # 4x4 image: pixel (2, 2) has corner coordinates (w/2, h/2) -> view-plane position (0, 0, f) (Projection.hs:34-39).
    # Sphere centre (0,0,1) r 0.5, plastic ior 1.5 flat (0.8,0.6,0.4); point light ON the camera axis at (0,0,-1);
    # maxDepth 0, so specular is black and the pixel is  accumDiffuse + fresnel * black  (RayHs.hs:104,117-120).
    w = h = 4.0
    fovy = 0.9272952180016123
    f = 0.5 * h / (math.tan(0.5) * fovy)
    dz = (1 / math.sqrt(0.0 * 0.0 + 0.0 * 0.0 + f * f)) * f            # normalize (Vec 0 0 f); lookAt here is the identity
    o = -2.0
    a = 0.0 * 0.0 + 0.0 * 0.0 + dz * dz                                 # Geometry.hs:86-91
    b = 2.0 * (0.0 * 0.0 + 0.0 * 0.0 + dz * (o - 1.0))
    c = (0.0 + 0.0 + (o - 1.0) * (o - 1.0)) - 0.5 * 0.5
    delta = b * b - 4.0 * a * c
    t0 = 0.5 * ((-b) - math.sqrt(delta)) / a
    assert t0 > 0
    pz = o + t0 * dz                                                    # rayAt
    nz = (1 / math.sqrt(0.0 + 0.0 + (pz - 1.0) * (pz - 1.0))) * (pz - 1.0)   # normalize (p0 - ct)
    assert nz == -1.0
    d = math.sqrt(0.0 + 0.0 + (-1.0 - pz) * (-1.0 - pz))                # Light.hs:15-17: dist p0 p
    ldz = (1 / d) * (-1.0 - pz)
    falloff = 1.0 / ((1.0 + d / 0.2) * (1.0 + d / 0.2))
    lam = max(0.0 * 0.0 + 0.0 * 0.0 + ldz * nz, 0) * (1 / math.pi)      # Material.hs:31-33
    cd, lc = (0.8, 0.6, 0.4), 30.0
    r = ((1.0 - 1.5) / (1.0 + 1.5)) ** 2                                # r0 1.0 ior (u*u)
    cos0 = 0.0 * -0.0 + 0.0 * -0.0 + nz * (-dz)                         # dot n (-v)
    x = 1 - cos0
    fres = r + (1 - r) * (x * x * x * x * x)
    want = [(0.0 + lam * (k * (falloff * lc))) + fres * 0.0 for k in cd]
    sph = {"geometry": {"type": "sphere", "center": {"x": 0, "y": 0, "z": 1}, "radius": 0.5},
           "material": {"type": "plastic", "ior": 1.5, "cd": {"type": "flat", "color": {"r": 0.8, "g": 0.6, "b": 0.4}}}}
    light = {"type": "point", "position": {"x": 0, "y": 0, "z": -1}, "color": {"r": 30, "g": 30, "b": 30}, "radius": 0.2}
    img = _render_text({"width": 4, "height": 4, "maxDepth": 0, "camera": _camera_json(fovy), "scene": {"objects": [sph], "lights": [light]}})
    assert list(img[2, 2]) == want and want[0] > 0.05


def test_checker_cell_and_ambient_only_pixel_by_hand():
    # a checker floor y = -1 (tangent x, so u = x, v = cross t n . rel = -z... evaluated below) seen by pixel (1, 3) of a
    # 4x4 image, no lights: the pixel is 0.2 * colorAt (RayHs.hs:111-114), and colorAt picks color1 iff
    # (mod' u s - 0.5 s) * (mod' v s - 0.5 s) < 0 (ColorMap.hs:21-24)
    from fractions import Fraction
    w = h = 4.0
    fovy = 0.9272952180016123
    f = 0.5 * h / (math.tan(0.5) * fovy)
    vx, vy = w * (1.0 - (w / 2)) / w, h * ((-3.0) + (h / 2)) / h        # aspect 1: apw = aspect*h = 4, aph = 4
    inv = 1 / math.sqrt(vx * vx + vy * vy + f * f)
    dx, dy, dz = inv * vx, inv * vy, inv * f
    ddn = dx * 0.0 + dy * 1.0 + dz * 0.0                                # Geometry.hs:70-79, n = (0,1,0), p = (0,-1,0)
    time = (0.0 * (0.0 - 0.0) + 1.0 * (-1.0 - 0.0) + 0.0 * (0.0 - -2.0)) / ddn
    assert abs(ddn) > 0 and time > 0
    px, py, pz = 0.0 + time * dx, 0.0 + time * dy, -2.0 + time * dz
    relx, rely, relz = px - 0.0, py - -1.0, pz - 0.0
    t = (1.0, 0.0, 0.0)
    bvec = (t[1] * 0.0 - t[2] * 1.0, t[2] * 0.0 - t[0] * 0.0, t[0] * 1.0 - t[1] * 0.0)   # cross t n
    u = t[0] * relx + t[1] * rely + t[2] * relz
    v = bvec[0] * relx + bvec[1] * rely + bvec[2] * relz
    s = 0.3

    def mod_prime(n, d):
        return n - float(math.floor(Fraction(n) / Fraction(d))) * d
    first = (mod_prime(u, s) - (0.5 * s)) * (mod_prime(v, s) - (0.5 * s)) < 0
    c1, c2 = (1.0, 0.25, 0.5), (0.125, 2.0, 0.75)
    want = [0.2 * k for k in (c1 if first else c2)]
    floor = {"geometry": {"type": "plane", "point": {"x": 0, "y": -1, "z": 0}, "normal": {"x": 0, "y": 1, "z": 0}, "tangent": {"x": 1, "y": 0, "z": 0}},
             "material": {"type": "diffuse", "cd": {"type": "checker", "color1": dict(zip("rgb", c1)), "color2": dict(zip("rgb", c2)), "size": s}}}
    img = _render_text({"width": 4, "height": 4, "maxDepth": 3, "camera": _camera_json(fovy), "scene": {"objects": [floor], "lights": []}})
    assert list(img[3, 1]) == want
    # and the neighbouring cell along u flips the colour
    other = [0.2 * k for k in (c2 if first else c1)]
    row = _render_text({"width": 64, "height": 64, "maxDepth": 3, "camera": _camera_json(fovy), "scene": {"objects": [floor], "lights": []}})[60]
    assert any(list(p) == want for p in row) and any(list(p) == other for p in row)


# ---- the second, independent restatement (oracle/second/, plain Python written from the Haskell alone)
SECOND = json.load(open(os.path.join(GOLD, "second.json")))


@pytest.mark.parametrize("name", [n for n, m in SECOND.items() if m["rows"] is not None])
def test_oracle_equals_the_second_restatement_on_windows_of_the_baseline_frames(name):
    # fixtures written by oracle/second/check_second.py: they did NOT come from the C oracle
    m = SECOND[name]
    want = np.load(os.path.join(GOLD, f"second_{name}.npy"))
    for i, y in enumerate(m["rows"]):
        row, _ = O.render_file(m["scene"], width=m["width"], height=m["height"], spp=m["spp"], seed=m["seed"], row0=y, row1=y + 1)
        assert np.array_equal(row[0][m["cols"]], want[i]), (name, y)


def test_second_restatement_agrees_live_on_a_small_frame():
    # the Python restatement itself, run here (seconds): a 24x24 frame of the dragon scene with every kind of ray
    from oracle.second import rayhs_second as S
    rows = S.render_file("scenes/dragon_superlow_96.json", width=24, height=24, spp=2, seed=24)
    first, _ = O.render_file("scenes/dragon_superlow_96.json", width=24, height=24, spp=2, seed=24)
    assert np.array_equal(np.array(rows), first)


@pytest.mark.parametrize("seed", range(60))
def test_the_two_restatements_agree_on_random_scenes(tmp_path, seed):
    # differential test of the two transcriptions where no fixture reaches: random materials, colour maps, nested
    # transforms, light kinds, projections, axis-aligned cameras (tests/scene_fuzz.py), 10x6 pixels, 1-3 samples.
    # NaN colours are legal (0/0 in a degenerate scene) and must appear in the same places.
    import sys
    from oracle.second import rayhs_second as S
    from scene_fuzz import rand_scene
    sys.setrecursionlimit(10000)
    rng = np.random.default_rng(9000 + seed)
    jd = rand_scene(rng)
    jd["width"], jd["height"] = 10, 6
    spp = int(rng.choice([1, 2, 3]))
    p = tmp_path / "f.json"
    p.write_text(json.dumps(jd))
    first, _ = O.render_file(str(p), spp=spp)
    second = np.array(S.render_file(str(p), spp=spp), dtype=np.float64)
    same = (first == second) | (np.isnan(first) & np.isnan(second))
    assert same.all(), (seed, int((~same).any(axis=2).sum()), float(np.nanmax(np.abs(first - second))))
