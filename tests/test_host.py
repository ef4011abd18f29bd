"""CPU tests of the product's host side: the C ABI loads and exports what
include/rayhs.h declares, the C++ scene loader (JSON -> OBJ -> transform -> tree
-> SoA) agrees with the oracle's independent route, error behaviour mirrors the
reference CLI (src/RayHs.hs:215-234), and compute entry points refuse to run
without a GPU instead of falling back to a CPU path.
"""
import ctypes as C
import json
import os
import re
import subprocess

import numpy as np
import pytest

import rayhs_amd
from rayhs_amd import RenderDesc, lib as L
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_GPU = L.load().rayhs_device_count() > 0


def test_library_is_in_tree_and_exports_every_declared_symbol():
    assert os.path.commonpath([L.LIB_PATH, ROOT]) == ROOT and os.path.exists(L.LIB_PATH)
    header = open(os.path.join(ROOT, "include", "rayhs.h")).read()
    declared = set(re.findall(r"\b(rayhs_[a-z0-9_]+)\s*\(", header))
    assert declared == {name for name, _, _ in L.SYMBOLS}
    lib = C.CDLL(L.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert L.load().rayhs_abi_version() == 5


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    # the drop-in boundary is a C ABI: include/rayhs.h must compile as strict C99 and link without any C++ or HIP header
    src = tmp_path / "client.c"
    src.write_text('#include "rayhs.h"\n#include <stdio.h>\n'
                   'int main(void) { rayhs_render_params p; rayhs_stats st; (void)p; (void)st;\n'
                   '  printf("%d %d\\n", rayhs_abi_version(), rayhs_device_count()); return rayhs_last_error() == 0; }\n')
    exe = tmp_path / "client"
    bindir = os.path.dirname(L.LIB_PATH)
    cc = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                         "-L", bindir, "-lrayhs_hip", "-Wl,-rpath," + bindir], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr
    run = subprocess.run([str(exe)], capture_output=True, text=True)
    assert run.returncode == 0 and run.stdout.split()[0] == "5"


def test_struct_sizes_match_the_header():
    # layout guard for the ctypes mirrors (x86-64 SysV)
    assert C.sizeof(L.Camera) == 3 * 24 + 8 + 4 * 8
    assert C.sizeof(L.RenderParams) == 56
    assert C.sizeof(L.Stats) == 12 * 8
    assert C.sizeof(L.SceneDesc) == 328
    # ... and against the library itself (rayhs_sizeof): a library of another layout must never be written into
    for which, cls in enumerate((L.SceneDesc, L.Camera, L.RenderParams, L.Stats)):
        assert L.load().rayhs_sizeof(which) == C.sizeof(cls)
    assert L.load().rayhs_sizeof(99) == 0


def test_loader_refuses_a_library_of_another_abi_version_or_layout(monkeypatch):
    # ABI version 2 grew rayhs_stats by 8 bytes; a caller that mirrors version 1 would be overrun by 8 bytes when it
    # asks for stats (include/rayhs.h, "ABI history").  The binding checks version and struct sizes when it loads.
    monkeypatch.setattr(L, "_LIB", None)
    monkeypatch.setattr(L, "ABI_VERSION", 1)
    with pytest.raises(ImportError, match="ABI version 5"):
        L.load()
    monkeypatch.setattr(L, "ABI_VERSION", 5)

    class Stats88(C.Structure):  # the version-1 layout: no shadow_skipped
        _fields_ = L.Stats._fields_[:-1]
    assert C.sizeof(Stats88) == 88
    monkeypatch.setattr(L, "Stats", Stats88)
    with pytest.raises(ImportError, match=r"sizeof\(rayhs_stats\) is 96"):
        L.load()
    monkeypatch.undo()
    # a library that lacks a declared entry point (an older build) is refused with the same kind of message
    monkeypatch.setattr(L, "_LIB", None)
    monkeypatch.setattr(L, "SYMBOLS", L.SYMBOLS + [("rayhs_entry_point_of_a_later_version", C.c_int, [])])
    with pytest.raises(ImportError, match="does not export rayhs_entry_point_of_a_later_version"):
        L.load()
    monkeypatch.undo()
    assert L.load().rayhs_abi_version() == 5


def test_diagnostic_entry_points_refuse_a_null_scene_without_touching_a_device():
    # rayhs_scene_capture_tile_order / rayhs_scene_tile_order (ABI 4) validate their handle first; rayhs_config_reload only re-reads
    # the environment: none of them needs a GPU
    lib = L.load()
    assert lib.rayhs_scene_capture_tile_order(None, 1) == L.ERR_INVALID
    assert lib.rayhs_scene_tile_order(None, None, 0, None) == L.ERR_INVALID and b"NULL" in lib.rayhs_last_error()
    lib.rayhs_config_reload()


def test_sampler_matches_oracle_definition():
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))["sampler"]
    for seed, pixel, sample, dim, spp, hexv in gold:
        assert L.load().rayhs_sample01(seed, pixel, sample, dim, spp) == float.fromhex(hexv)


# ------------------------------------------------------------- scene decoding
@pytest.mark.parametrize("scene", ["cornellBox", "dragon", "outScene", "transform"])
def test_flattened_scene_matches_python_json_and_oracle_meshes(scene):
    path = f"data/{scene}.json"
    rd = RenderDesc(path)
    a = rd.arrays()
    jd = json.load(open(path))
    assert (rd.width, rd.height, rd.max_depth) == (jd["width"], jd["height"], jd["maxDepth"])
    objs = jd["scene"]["objects"]
    kinds = {"plane": 0, "sphere": 1, "mesh": 2}
    assert list(a["obj_geom_kind"]) == [kinds[o["geometry"]["type"]] for o in objs]
    mats = {"mirror": 0, "diffuse": 1, "plastic": 2, "emmit": 3, "transparent": 4}
    assert list(a["mat_kind"]) == [mats[o["material"]["type"]] for o in objs]
    for i, o in enumerate(objs):
        g, gi = o["geometry"], a["obj_geom_index"][i]
        if g["type"] == "sphere":
            assert list(a["sphere_center"][gi]) == [g["center"][k] for k in "xyz"] and a["sphere_radius"][gi] == g["radius"]
        elif g["type"] == "plane":  # net effect of JSON.hs:115-118 + Descriptors.hs:51: point, normal, tangent
            assert list(a["plane_point"][gi]) == [g["point"][k] for k in "xyz"]
            assert list(a["plane_normal"][gi]) == [g["normal"][k] for k in "xyz"]
            assert list(a["plane_tangent"][gi]) == [g["tangent"][k] for k in "xyz"]
        m = o["material"]
        if "ior" in m:
            assert a["mat_ior"][i] == m["ior"]
        if m["type"] in ("diffuse", "plastic") and m["cd"]["type"] == "checker":
            assert a["cmap_kind"][i] == 1 and a["cmap_size"][i] == m["cd"]["size"]
            assert list(a["cmap_c1"][i]) == [m["cd"]["color1"][k] for k in "rgb"]
    lights = jd["scene"]["lights"]
    assert list(a["light_kind"]) == [0 if l["type"] == "directional" else 1 for l in lights]
    cam = rd.camera
    assert list(cam.position) == [jd["camera"]["position"][k] for k in "xyz"] and cam.fovy == jd["camera"]["projection"]["fovy"]
    # meshes: same triangles (bit for bit) as the oracle's own OBJ loader + transform, leaves in DFS order
    sc = O.Scene(jd["scene"])
    for i, o in enumerate(objs):
        if o["geometry"]["type"] != "mesh":
            continue
        want = sc.mesh_triangles(i)
        n = len(want)
        root = a["mesh_root"][a["obj_geom_index"][i]]
        order, stack = [], [root]
        while stack:
            k = stack.pop()
            if k < 0:
                continue
            if a["node_tri_count"][k] >= 0:
                order += list(range(a["node_first_tri"][k], a["node_first_tri"][k] + a["node_tri_count"][k]))
            else:
                stack += [a["node_right"][k], a["node_left"][k]]
        assert order == list(range(order[0], order[0] + n))
        got = np.concatenate([a["tri_pos"][order], a["tri_nrm"][order], a["tri_uv"][order]], axis=2).reshape(n, -1)
        ref = want.reshape(n, -1)
        assert np.array_equal(got[np.lexsort(got.T[::-1])], ref[np.lexsort(ref.T[::-1])])


SURVEY_TREES = {"cube": (12, 0, 1, 0, 0), "cone": (62, 5, 4, 2, 3), "torus": (576, 38, 39, 0, 6),
                "dragon_superlow": (3402, 279, 280, 0, 11), "dragon_low": (6807, 547, 548, 0, 12), "dragon": (27228, 2199, 2200, 0, 16)}


def _mesh_scene(fname, transform=None, w=8, h=8):
    t = transform or {"type": "translate", "vector": {"x": 0, "y": 0, "z": 0}}
    return json.dumps({"width": w, "height": h, "maxDepth": 1,
                       "camera": {"position": {"x": 0, "y": 0, "z": -3}, "target": {"x": 0, "y": 0, "z": 0}, "up": {"x": 0, "y": 1, "z": 0},
                                  "projection": {"type": "perspective", "fovy": 0.9, "width": 2, "height": 2, "near": 0.1}},
                       "scene": {"objects": [{"geometry": {"type": "mesh", "fileName": fname, "transform": t},
                                              "material": {"type": "mirror", "ior": 1.5}}], "lights": []}})


@pytest.mark.parametrize("mesh", sorted(SURVEY_TREES))
def test_tree_build_matches_survey_table(mesh):
    rd = RenderDesc(text=_mesh_scene(f"data/{mesh}.obj"))
    a = rd.arrays()
    tris, inner, leaves, empties, depth = SURVEY_TREES[mesh]
    leaf = a["node_tri_count"] >= 0
    assert len(a["tri_pos"]) == tris and int(leaf.sum()) == leaves and int((~leaf).sum()) == inner
    kids = np.concatenate([a["node_left"][~leaf], a["node_right"][~leaf]])
    assert int((kids == -1).sum()) == empties and rd.tree_depth == depth
    # round-robin axis from the root, boxes are tight (KDTree.hs:79-90)
    assert a["node_axis"][a["mesh_root"][0]] == 0
    for k in np.nonzero(leaf)[0][:50]:
        pts = a["tri_pos"][a["node_first_tri"][k]: a["node_first_tri"][k] + a["node_tri_count"][k]].reshape(-1, 3)
        assert np.array_equal(pts.min(0), a["node_lo"][k]) and np.array_equal(pts.max(0), a["node_hi"][k])


def test_rotations_take_sin_and_cos_by_separate_calls(tmp_path):
    # gcc -O2 turns sin(a) and cos(a) of one angle into ONE call of glibc's sincos(), whose sine is one ulp off sin()'s for some angles
    # (this one: found by the round-6 soak over fresh fuzz seeds); GHC calls sin and cos, each by itself (src/Mat.hs:58-74).  Both the oracle
    # (oracle/Makefile: -fno-builtin-sin -fno-builtin-cos) and the host scene preparation must equal Python's math.sin / math.cos here.
    import math
    angle = 0.9261123132458575
    assert math.sin(angle).hex() == "0x1.993c7fce80301p-1"        # (the sincos() value ends in ...300)
    obj = tmp_path / "t.obj"
    obj.write_text("v 1 0 0\nv 0 1 0\nv 0 0 1\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    for kind, want in (("rotateY", [[math.cos(angle), 0.0, -math.sin(angle)], [0.0, 1.0, 0.0], [math.sin(angle), 0.0, math.cos(angle)]]),
                       ("rotateX", [[1.0, 0.0, 0.0], [0.0, math.cos(angle), math.sin(angle)], [0.0, -math.sin(angle), math.cos(angle)]]),
                       ("rotateZ", [[math.cos(angle), math.sin(angle), 0.0], [-math.sin(angle), math.cos(angle), 0.0], [0.0, 0.0, 1.0]])):
        t = {"type": kind, "angle": angle}
        host = RenderDesc(text=_mesh_scene(str(obj), t)).arrays()["tri_pos"][0]
        orc = O.Scene(json.loads(_mesh_scene(str(obj), t))["scene"]).mesh_triangles(0)[0, :, :3]
        assert np.array_equal(host, np.array(want)), kind      # (the images of the unit vectors are the matrix's columns)
        assert np.array_equal(orc, np.array(want)), kind


def test_rotate_about_axis_and_scale_follow_reference_matrices(tmp_path):
    obj = tmp_path / "t.obj"
    obj.write_text("v 1 0 0\nv 0 1 0\nv 0 0 1\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    t = {"type": "sequence", "transforms": [{"type": "scale", "vector": {"x": 2, "y": 3, "z": 4}},
                                            {"type": "rotate", "axis": {"x": 0.6, "y": 0.8, "z": 0}, "angle": 0.7},
                                            {"type": "rotateX", "angle": 0.3}, {"type": "rotateY", "angle": -1.1},
                                            {"type": "rotateZ", "angle": 2.0}]}
    a = RenderDesc(text=_mesh_scene(str(obj), t)).arrays()
    want = O.Scene(json.loads(_mesh_scene(str(obj), t))["scene"]).mesh_triangles(0)
    assert np.array_equal(a["tri_pos"][0], want[0, :, :3]) and np.array_equal(a["tri_nrm"][0], want[0, :, 3:6])
    assert np.allclose(np.linalg.norm(a["tri_pos"][0], axis=1), [2, 3, 4], rtol=1e-14)   # rotations keep lengths
    # an axis with x = y = 0 makes orthonormal normalise the zero vector (Vec.hs:151-158): NaN, as in the reference
    t = {"type": "rotate", "axis": {"x": 0, "y": 0, "z": 1}, "angle": 0.7}
    a = RenderDesc(text=_mesh_scene(str(obj), t)).arrays()
    assert np.isnan(a["tri_pos"]).all()


# ------------------------------------------------------------------- failures
def _fails_to_read(text):
    with pytest.raises(L.RayhsError) as e:
        RenderDesc(text=text)
    assert e.value.code == L.ERR_PARSE and e.value.message.startswith("Failed to read scene")


def test_missing_key_wrong_type_and_bad_json_fail_like_aeson():
    good = json.load(open("data/cornellBox.json"))
    for key in ("scene", "camera", "width", "height", "maxDepth"):
        bad = dict(good)
        del bad[key]
        _fails_to_read(json.dumps(bad))
    bad = json.loads(json.dumps(good))
    bad["width"] = 512.5                       # Int field (JSON.hs:139)
    _fails_to_read(json.dumps(bad))
    bad = json.loads(json.dumps(good))
    bad["scene"]["objects"][0]["material"]["type"] = "glossy"
    _fails_to_read(json.dumps(bad))
    bad = json.loads(json.dumps(good))
    del bad["scene"]["lights"][0]["radius"]
    _fails_to_read(json.dumps(bad))
    _fails_to_read("{ not json")
    _fails_to_read(json.dumps(good) + " trailing")


def test_example_dumps_snippets():
    # data/example_dumps.json:22-51 are one-line parser fixtures; the mesh ones use the stale key
    # "translation" and must FAIL on the live parser, which requires "transform" (JSON.hs:119-121)
    lines = [l for l in open("data/example_dumps.json").read().split("\n") if l.startswith("{ ")]
    base = json.load(open("data/cornellBox.json"))
    n_ok = n_bad = 0
    for ln in lines:
        v = json.loads(ln)
        doc = json.loads(json.dumps(base))
        t = v.get("type")
        if "geometry" in v and "material" in v:
            doc["scene"]["objects"] = [v]
        elif "objects" in v:
            doc["scene"] = v
        elif t in ("flat", "checker"):
            doc["scene"]["objects"][0]["material"] = {"type": "diffuse", "cd": v}
        elif t in ("perspective", "orthographic"):
            doc["camera"]["projection"] = v
        elif t in ("diffuse", "mirror", "plastic", "transparent", "emmit"):
            doc["scene"]["objects"][0]["material"] = v
        elif t in ("plane", "sphere", "mesh"):
            doc["scene"]["objects"][0]["geometry"] = v
        elif t in ("directional", "point"):
            doc["scene"]["lights"] = [v]
        else:
            raise AssertionError(ln)
        stale = "translation" in ln
        if stale:
            _fails_to_read(json.dumps(doc))
            n_bad += 1
        else:
            RenderDesc(text=json.dumps(doc)).close()
            n_ok += 1
    assert n_ok == 12 and n_bad == 3


def test_missing_mesh_file_and_bad_obj(tmp_path):
    with pytest.raises(L.RayhsError) as e:
        RenderDesc(text=_mesh_scene("data/nope.obj"))
    assert e.value.code == L.ERR_IO and "does not exist" in e.value.message
    with pytest.raises(L.RayhsError) as e:
        RenderDesc("data/nope.json")
    assert e.value.code == L.ERR_IO
    q = tmp_path / "quad.obj"
    q.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n")     # 4 indices do not regroup by 3
    with pytest.raises(L.RayhsError) as e:
        RenderDesc(text=_mesh_scene(str(q)))
    assert e.value.code == L.ERR_PARSE
    b = tmp_path / "badnum.obj"
    b.write_text("v 0 0 .5\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")               # ".5" is not Haskell-readable
    with pytest.raises(L.RayhsError, match="no parse"):
        RenderDesc(text=_mesh_scene(str(b)))
    o = tmp_path / "oob.obj"
    o.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n")
    with pytest.raises(L.RayhsError, match="index"):
        RenderDesc(text=_mesh_scene(str(o)))


def test_obj_quirks_match_oracle(tmp_path):
    p = tmp_path / "q.obj"
    p.write_text("# c\nv 0 0 0\r\nv 1 0 0\nv 1 1 0\nv 0 1 0\n v 9 9 9\nvt 0.5 0.25\nvn 0 0 1\nvx 1 2 3\n"
                 "f 1/1/1 2//1 3/1\nf 1 2 3 4 x 5/\nf 1 2\n")
    # ' v 9 9 9' (leading blank) is ignored; 'f 1 2' (<3 indices) is ignored; 'x' is dropped; '5/' parses as 5 -> out of range?
    with pytest.raises(L.RayhsError):
        RenderDesc(text=_mesh_scene(str(p)))                             # index 5 does not exist (only 4 'v ' lines)
    p.write_text("# c\nv 0 0 0\r\nv 1 0 0\nv 1 1 0\nv 0 1 0\n v 9 9 9\nvt 0.5 0.25\nvn 0 0 1\nvx 1 2 3\n"
                 "f 1/1/1 2//1 3/1\nf 1 2 3 4 x 4/ 1\nf 1 2\n")
    a = RenderDesc(text=_mesh_scene(str(p))).arrays()
    sc = O.Scene(json.loads(_mesh_scene(str(p)))["scene"])
    want = sc.mesh_triangles(0)
    assert len(a["tri_pos"]) == 3 == len(want)                           # 3 + 6 indices regrouped
    got = np.concatenate([a["tri_pos"], a["tri_nrm"], a["tri_uv"]], axis=2)
    assert np.array_equal(got, want)                                     # one leaf: file order is kept
    assert list(got[0, 0]) == [0, 0, 0, 0, 0, 1, 0.5, 0.25] and list(got[0, 1]) == [1, 0, 0, 0, 0, 1, 0, 0]
    assert list(got[0, 2]) == [1, 1, 0, 0, 0, 0, 0.5, 0.25]


# -------------------------------------------------------------- no CPU fallback
@pytest.mark.skipif(HAVE_GPU, reason="needs a machine without a GPU")
def test_compute_entry_points_refuse_without_gpu():
    rd = RenderDesc("scenes/cornell_96.json")
    h = C.c_void_p()
    rc = L.load().rayhs_scene_create(C.byref(rd.scene_desc), 0, C.byref(h))
    assert rc == L.ERR_NO_DEVICE and not h.value
    assert b"no CPU fallback" in L.load().rayhs_last_error()
    assert L.load().rayhs_device_prepare(0) == L.ERR_NO_DEVICE   # the optional warm-up call reports the same thing
    with pytest.raises(L.RayhsError):
        rayhs_amd.ray_trace_file("scenes/cornell_96.json")
    r = rayhs_amd.run_cli(["-o/tmp/_x.ppm", "scenes/cornell_96.json"], cwd=ROOT)
    assert r.returncode == 1 and "no HIP device" in r.stderr and not os.path.exists("/tmp/_x.ppm")


def test_multi_device_entry_validates_its_arguments_and_refuses_without_gpu():
    rd = RenderDesc("scenes/cornell_96.json")
    h = C.c_void_p()
    lib = L.load()
    assert lib.rayhs_multi_create(C.byref(rd.scene_desc), None, 2, C.byref(h)) == L.ERR_INVALID
    assert lib.rayhs_multi_create(C.byref(rd.scene_desc), (C.c_int * 1)(0), 0, C.byref(h)) == L.ERR_INVALID
    rc = lib.rayhs_multi_create(C.byref(rd.scene_desc), (C.c_int * 2)(0, 99), 2, C.byref(h))
    assert rc == (L.ERR_INVALID if HAVE_GPU else L.ERR_NO_DEVICE) and not h.value
    assert lib.rayhs_multi_device_count(None) == 0
    lib.rayhs_multi_destroy(None)                            # like free(NULL)
    if not HAVE_GPU:
        r = rayhs_amd.run_cli(["-o/tmp/_x.ppm", "--gpus=2", "scenes/cornell_96.json"], cwd=ROOT)
        assert r.returncode == 1 and "no HIP device" in r.stderr


def test_scene_desc_validation_rejects_broken_trees():
    rd = RenderDesc("scenes/cornell_96.json")
    d = rd.scene_desc
    h = C.c_void_p()
    left = np.ctypeslib.as_array(d.node_left, shape=(d.n_nodes,))
    inner = int(np.nonzero(np.ctypeslib.as_array(d.node_tri_count, shape=(d.n_nodes,)) < 0)[0][0])
    keep = int(left[inner])
    for bad in (d.n_nodes + 5, inner):          # out of range; a cycle
        left[inner] = bad
        assert L.load().rayhs_scene_create(C.byref(d), 0, C.byref(h)) == L.ERR_INVALID
    left[inner] = keep
    kinds = np.ctypeslib.as_array(d.obj_geom_kind, shape=(d.n_objects,))
    kinds[0] = 9
    assert L.load().rayhs_scene_create(C.byref(d), 0, C.byref(h)) == L.ERR_INVALID
    assert b"object 0" in L.load().rayhs_last_error()


def test_scene_desc_validation_rejects_null_arrays_with_a_positive_count():
    # every array whose count is > 0 must be there: checked before anything is dereferenced (and before the device is)
    import copy
    rd = RenderDesc("scenes/dragon_superlow_96.json")
    h = C.c_void_p()
    for field in ("mat_kind", "mat_ior", "cmap_kind", "cmap_c1", "plane_normal", "sphere_radius", "mesh_root", "node_lo",
                  "node_tri_count", "node_axis", "tri_pos", "tri_uv", "light_kind", "light_radius", "obj_geom_index"):
        d = L.SceneDesc.from_buffer_copy(rd.scene_desc)
        setattr(d, field, type(getattr(d, field))())      # NULL
        assert L.load().rayhs_scene_create(C.byref(d), 0, C.byref(h)) == L.ERR_INVALID, field
        assert field.encode() in L.load().rayhs_last_error(), field
    d = L.SceneDesc.from_buffer_copy(rd.scene_desc)       # a NULL array with a zero count is fine (here: no spheres, no lights)
    d.n_lights = 0
    d.light_kind = type(d.light_kind)()
    rc = L.load().rayhs_scene_create(C.byref(d), 0, C.byref(h))
    assert rc in (L.OK, L.ERR_NO_DEVICE)
    if rc == L.OK:
        L.load().rayhs_scene_destroy(h)


# ------------------------------------------------------------- params / output
def test_render_rows_follows_the_stripe_rule():
    from rayhs_amd import tiles
    lib = L.load()
    for world, height, stripe in ((1, 1080, 8), (2, 1080, 8), (8, 1080, 8), (8, 2160, 16), (3, 10, 4), (4, 3, 8)):
        for r in range(world):
            p = L.RenderParams(width=4, height=height, max_depth=1, spp=1, seed=0, row0=0, row1=height,
                               stripe_rows=stripe, stripe_rank=r, stripe_count=world)
            assert lib.rayhs_render_rows(C.byref(p)) == len(tiles.stripe_rows_of(r, world, height, stripe))
    p = L.RenderParams(width=4, height=10, row0=0, row1=10, stripe_rows=0, stripe_rank=0, stripe_count=1)
    assert lib.rayhs_render_rows(C.byref(p)) == L.ERR_INVALID


def test_ppm_writers_are_byte_exact(tmp_path):
    img = np.array([[[0, 0.5, 1], [2, -0.5, 0.25]], [[1, 1, 1], [float("nan"), 0, 0]]], dtype=np.float64)
    rayhs_amd.write_ppm(tmp_path / "a.ppm", img)
    assert (tmp_path / "a.ppm").read_bytes() == b"P3\n2 2\n255\n0 127 255  255 -127 63  \n255 255 255  255 0 0  "
    q = np.array([[[0, 127, 255], [255, -127, 63]]], dtype=np.int16)
    rayhs_amd.write_ppm(tmp_path / "b.ppm", q)
    assert (tmp_path / "b.ppm").read_bytes() == b"P3\n2 1\n255\n0 127 255  255 -127 63  "
    O.write_ppm(str(tmp_path / "c.ppm"), img)                             # oracle's writer agrees
    assert (tmp_path / "c.ppm").read_bytes() == (tmp_path / "a.ppm").read_bytes()


# ------------------------------------------------------------------------ CLI
def test_cli_argument_errors_mirror_the_reference():
    r = rayhs_amd.run_cli([])
    assert r.returncode == 1 and "No input." in r.stderr                          # RayHs.hs:233
    r = rayhs_amd.run_cli(["--bogus", "x.json"])
    assert r.returncode == 1 and "unrecognized option `--bogus'" in r.stderr and "Usage: main [OPTION...]" in r.stderr
    r = rayhs_amd.run_cli(["data/example_dumps.json"], cwd=ROOT)
    assert r.returncode == 1 and r.stdout == "Loading scene from data/example_dumps.json...\n"
    assert "Failed to read scene" in r.stderr                                       # RayHs.hs:226
    r = rayhs_amd.run_cli(["-o", "data/nope.json"], cwd=ROOT)                       # bare -o keeps out.ppm, arg is the scene
    assert r.returncode == 1 and r.stdout.startswith("Loading scene from data/nope.json...")


def test_committed_profiles_carry_the_two_roofs_bench_py_prices_with():
    # bench.py's `roofline` (the larger of the vector-pipe and the scalar-unit occupancy) takes the trace kernel's instruction
    # counts, by cost class, from profiles/rNN/<workload>/summary.json when that profile was taken on this very kernel source.
    # The derived fields must be there and consistent with the raw counters (tools/summarize_profiles.py), and every fraction
    # must be a fraction -- also when re-priced at bench.py's peak with the profile's own kernel time.
    import glob
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "dragon", "summary.json")))
    assert files, "no committed profile of the default workload"
    d = json.load(open(files[-1]))
    c = next(v for k, v in d["pmc_mean_per_launch"].items() if "count" not in k)
    classes = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")
    assert c["insts_per_launch"] == pytest.approx(sum(c[k] for k in classes))
    cls, cost = c["valu_class_counts"], d["valu_cost_simd_cycles"]
    assert cost == bench.VALU_COST and d["scalar_cost_cu_cycles"] == bench.SCALAR_COST_CU_CYCLES
    assert sum(cls.values()) == pytest.approx(c["SQ_INSTS_VALU"])
    assert cls["f64_arith"] == pytest.approx(c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"])
    assert c["scalar_unit_insts_per_launch"] == pytest.approx(c["SQ_INSTS_SALU"] + c["SQ_INSTS_SMEM"] + c["SQ_INSTS_BRANCH"])
    assert 0 < c["valu_pipe_frac"] <= c["valu_pipe_frac_measured_costs"] <= 1 and 0 < c["scalar_unit_frac"] <= 1
    assert c["roofline_frac"] == max(c["valu_pipe_frac"], c["scalar_unit_frac"]) and c["roofline_bound"] in ("valu", "scalar-issue")
    assert 0 < c["scalar_dcache_hit_rate"] <= 1
    sec = d["timed_launches"]["avg_ms"] * 1e-3
    base = sum(cost[k] * cls[k] for k in ("f32_arith", "f64_arith", "f64_trans", "f32_trans", "int_cvt"))
    assert 0 < (base + cost["other_measured"] * cls["other"]) / sec / 1e9 / (bench.N_SIMDS * bench.MAX_CLOCK_GHZ) <= 1
    assert 0 < bench.SCALAR_COST_CU_CYCLES * c["scalar_unit_insts_per_launch"] / sec / 1e9 / (bench.N_CUS * bench.MAX_CLOCK_GHZ) <= 1
    if d.get("kernel_source_sha256") != bench.kernel_source_sha256():
        pytest.skip("the committed profile was taken on another kernel source: bench.py will count instructions live (rocprofv3 --pmc child runs); "
                    "re-run tools/profile_round.sh + tools/summarize_profiles.py before the round ends")


def test_a_profiled_bench_run_never_starts_a_profiler_of_its_own():
    # tools/*.sh: every `rocprofv3 ... -- python3 bench.py` line must run with the live counter passes switched off (an exported
    # RAYHS_BENCH_NO_LIVE_PMC=1 that is still in force at that line, or --pmc-child on the line itself): bench.py otherwise starts
    # three `rocprofv3 --pmc` children from inside a process that is already being profiled whenever no committed profile matches the
    # kernel source -- the normal case while a round's profiles are being taken.
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tools = os.path.join(root, "tools")
    checked = 0
    for name in sorted(os.listdir(tools)):
        if not name.endswith(".sh"):
            continue
        guarded = False   # an `export RAYHS_BENCH_NO_LIVE_PMC=1` seen so far in this script
        for ln in open(os.path.join(tools, name)):
            code = ln.split("#")[0]
            if re.search(r"\bexport\s+RAYHS_BENCH_NO_LIVE_PMC=1\b", code):
                guarded = True
            if re.search(r"\bexport\s+RAYHS_BENCH_NO_LIVE_PMC=\s*$", code) or re.search(r"\bunset\s+RAYHS_BENCH_NO_LIVE_PMC\b", code):
                guarded = False
            if "rocprofv3" in code and re.search(r"(?<![A-Za-z_])bench\.py", code):   # (not tools/kbench.py: that one starts no profiler)
                checked += 1
                inline_off = re.search(r"RAYHS_BENCH_NO_LIVE_PMC=(\s|$)", code.split("rocprofv3")[0]) is not None
                assert ("--pmc-child" in code or (guarded and not inline_off)), f"tools/{name}: profiled bench.py without the guard: {ln.strip()}"
    assert checked >= 2   # the --kernel-trace run and the pmc() passes of tools/profile_round.sh


def test_kernel_source_hash_ignores_comments_but_not_code_behind_an_apostrophe():
    import bench
    a = "int a = 1'000'000; // c\nchar q = '\\''; char r = 'x'; /* y */ const char *s = \"it's // not\";\n#error don't do this // gone\nint z = 2;\n"
    got = bench._code_only(a)
    assert got == "int a = 1'000'000;\nchar q = '\\''; char r = 'x';  const char *s = \"it's // not\";\n#error don't do this\nint z = 2;\n"
    # code behind a digit separator or a stray apostrophe still counts: changing it changes what is hashed
    assert bench._code_only(a.replace("int z = 2;", "int z = 3;")) != got
    assert bench._code_only("x = 1'0; // a\ny = 2;\n") != bench._code_only("x = 1'0; // a\ny = 3;\n")


def test_selftest_arith_validates_its_arguments_and_refuses_without_gpu():
    lib = L.load()
    out = (C.c_uint64 * 5)()
    assert lib.rayhs_selftest_arith(0, 0, 1, out) == L.ERR_INVALID            # n = 0
    assert lib.rayhs_selftest_arith(0, 1 << 33, 1, out) == L.ERR_INVALID      # n beyond 2^32
    assert lib.rayhs_selftest_arith(0, 1000, 1, None) == L.ERR_INVALID        # nowhere to report
    if lib.rayhs_device_count() < 1:
        assert lib.rayhs_selftest_arith(0, 1000, 1, out) == L.ERR_NO_DEVICE


def test_basic_block_profiler_rewrite_counts_every_straight_line_segment_once(tmp_path):
    # tools/bbprof_rewrite.py on a small synthetic listing: one s_atomic_add at the function entry, behind every branch-target label and
    # behind every conditional branch; every original instruction in exactly one segment; the two scalar registers declared; other kernels untouched
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asm = tmp_path / "dev.s"
    asm.write_text("\n".join([
        "other_kernel:", "\ts_endpgm", ".Lfunc_end0:",
        "rayhs_trace_exact:", ".Lfunc_begin1:",
        "\t.loc\t2 10 3                          ; rayhs_amd/csrc/trace_kernel.hip:10:3",
        "\ts_load_dwordx2 s[4:5], s[0:1], 0x1c8",
        "\tv_mov_b32_e32 v1, 0",
        ".LBB1_1:                                ; a loop",
        "\t.loc\t2 0 0                           ; rayhs_amd/csrc/trace_kernel.hip:0:0",
        "\tv_add_u32_e32 v1, 1, v1",
        "\t.loc\t2 22 5                          ; rayhs_amd/csrc/trace_kernel.hip:22:5 @[ rayhs_amd/csrc/trace_kernel.hip:40:9 ]",
        "\ts_cmp_lt_u32 s2, s3",
        "\ts_cbranch_scc1 .LBB1_1",
        "\tv_mov_b32_e32 v2, v1",
        "\ts_endpgm",
        ".Lfunc_end1:",
        "\t.amdhsa_kernel rayhs_trace_exact", "\t\t.amdhsa_next_free_sgpr 100", "\t.end_amdhsa_kernel",
        "\t.set rayhs_trace_exact.numbered_sgpr, max(100, 3)", ""]))
    spec = importlib.util.spec_from_file_location("bbprof_rewrite", os.path.join(root, "tools", "bbprof_rewrite.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out, mapping = tmp_path / "mod.s", tmp_path / "map.json"
    argv = ["bbprof_rewrite.py", str(asm), str(out), str(mapping)]
    old = __import__("sys").argv
    try:
        __import__("sys").argv = argv
        mod.main()
    finally:
        __import__("sys").argv = old
    text = out.read_text()
    m = json.load(open(mapping))
    assert text.count("s_atomic_add s101, s[100:101]") == 3 == len(m["segments"])            # entry, the loop's label, behind the conditional branch
    assert [len(s) for s in m["segments"]] == [2, 3, 2]                                        # every original instruction once
    assert m["counters_offset"] == 0x1c8 - 16                                                  # KArgs.counters sits 16 bytes in front of KArgs.frames
    assert m["segments"][1][0][1] == [10]                                                      # a line-0 position keeps the one in force before it
    assert m["segments"][1][1][1] == [22, 40]                                                  # innermost line first, then its inline callers
    assert ".amdhsa_next_free_sgpr 102" in text and "numbered_sgpr, max(102," in text
    assert text.split("rayhs_trace_exact:")[0] == asm.read_text().split("rayhs_trace_exact:")[0]   # the other kernel is untouched
    assert "s_waitcnt lgkmcnt(0)\n\ts_dcache_wb\n\ts_endpgm" in text.split("rayhs_trace_exact:")[1]
