"""GPU parity tests: the HIP trace path, called through the C ABI
(librayhs_hip.so), against the CPU oracle and the committed fixtures.

Bar: the default kernels are built with -ffp-contract=off and perform the
reference's fp64 operations one for one, so images must equal the oracle's
BIT FOR BIT (max |diff| == 0.0), far inside the 1e-4-per-channel contract of
BASELINE.json.  The only device/host libm calls on the path are atan/acos in the
sphere UV (src/Geometry.hs:96); scenes using them are held to 1e-4 instead.
At BASELINE sizes the oracle is too slow, so full-size cases use size-independent
properties: stripe decomposition, exhaustive-vs-pruned traversal, ray identities.
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import rayhs_amd
from rayhs_amd import RenderDesc, Rendering, build_rendering, lib as L, tiles
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
META = json.load(open(os.path.join(GOLD, "golden.json")))
TOL = 1e-4  # BASELINE.json: per RGB channel


def oracle_image(path, **kw):
    return O.render_file(path, **kw)


@pytest.fixture(scope="module", autouse=True)
def _native_library_loaded():
    lib = L.load()
    assert lib.rayhs_device_count() >= 1, "no HIP device: GPU tests cannot run (and must not fall back)"
    assert any("librayhs_hip.so" in l for l in open("/proc/self/maps")), "native library not mapped"


# ------------------------------------------------ fixtures and oracle, bit-exact
# Every oracle / fixture comparison runs on the kernels that SHIP (flags 0: rayhs_trace_exact, or rayhs_trace_wide_exact
# for scenes with >= 65536 nodes) as well as on the instrumented variant (FLAG_COUNT: rayhs_trace_count_exact, whose
# work counters are compared too).  The reference-path replay and the shadow-query shortcut live in both.
PRODUCTION, COUNTING = 0, L.FLAG_COUNT
BOTH_KERNELS = pytest.mark.parametrize("flags", [PRODUCTION, COUNTING], ids=["production", "counting"])


@BOTH_KERNELS
@pytest.mark.parametrize("name", ["cornell_96", "cornell_256", "dragon_superlow_96", "dragon_superlow_96_spp4", "transform_64"])
def test_bit_exact_against_golden_and_oracle(name, flags):
    m = META[name]
    img, st = rayhs_amd.ray_trace_file(m["scene"], spp=m["spp"], seed=m["seed"], flags=flags)
    gold = np.load(os.path.join(GOLD, name + ".npy"))
    assert img.dtype == np.float64 and np.array_equal(img, gold)
    if flags == PRODUCTION:
        return
    assert st["pixels"] == m["width"] * m["height"]
    for k in ("rays_primary", "rays_reflect", "rays_refract", "rays_refract_exit", "rays_shadow", "shaded_hits"):
        assert st[k] == m["stats"][k], k          # same rays, whatever the traversal order
    # the fast BVH has its own (smaller, fp32) boxes, so box-test counts are not comparable; its leaves
    # hold at most 2 triangles against the reference's 19, so it never tests more triangles
    assert st["tris_tested"] <= m["stats"]["tris_tested"]


@BOTH_KERNELS
def test_the_other_reading_of_the_fifth_power_is_one_flag_away(flags):
    # fresnel's x ^ (5 :: Int): flat product by default, square-and-multiply with RAYHS_FLAG_POW5_SQMUL / orc_set_pow5_variant
    # (kept selectable until a value from a real GHC -O2 build settles it).  Each kernel reading is
    # bit-identical to the oracle's same reading, and the two readings do differ on a reflective scene.
    scene = "scenes/dragon_superlow_96.json"
    flat_gpu, _ = rayhs_amd.ray_trace_file(scene, spp=2, flags=flags)
    sq_gpu, _ = rayhs_amd.ray_trace_file(scene, spp=2, flags=flags | L.FLAG_POW5_SQMUL)
    flat_cpu, _ = O.render_file(scene, spp=2)
    with O.pow5_variant(True):
        sq_cpu, _ = O.render_file(scene, spp=2)
    assert np.array_equal(flat_gpu, flat_cpu) and np.array_equal(sq_gpu, sq_cpu)
    assert not np.array_equal(flat_cpu, sq_cpu) and np.abs(flat_cpu - sq_cpu).max() < 1e-12


@pytest.mark.parametrize("name", ["outscene_128x72", "texture_earth_96"])
def test_scenes_with_sphere_uv_within_tolerance(name):
    # checker / texture on spheres go through atan and acos (device libm vs glibc)
    m = META[name]
    img, _ = rayhs_amd.ray_trace_file(m["scene"], spp=m["spp"], seed=m["seed"])
    gold = np.load(os.path.join(GOLD, name + ".npy"))
    d = np.abs(img - gold)
    assert d.max() <= TOL, f"max |d| = {d.max()} at {np.unravel_index(d.argmax(), d.shape)}"


@pytest.mark.parametrize("scene,w,h,spp", [("data/cornellBox.json", 200, 120, 1), ("data/dragon.json", 160, 160, 1),
                                           ("scenes/dragon_low_1080p.json", 120, 68, 4), ("data/transform.json", 77, 53, 2)])
def test_bit_exact_against_oracle_at_other_sizes(scene, w, h, spp):
    want, ost = oracle_image(scene, width=w, height=h, spp=spp)
    got, st = rayhs_amd.ray_trace_file(scene, width=w, height=h, spp=spp, flags=COUNTING)
    assert np.array_equal(got, want)
    assert st["rays"] == ost["rays"]
    got, _ = rayhs_amd.ray_trace_file(scene, width=w, height=h, spp=spp, flags=PRODUCTION)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("spp", [2, 3, 5, 8, 16, 32, 64, 96])
def test_sample_fold_is_in_sample_order_for_every_spp(spp, library_env):
    # spp dividing 64 is folded inside the wave, anything else through the sample buffer + resolve pass;
    # both must equal the reference's foldl over samples (src/RayHs.hs:169-171), with and without counters
    scene = "data/transform.json"
    want, _ = oracle_image(scene, width=41, height=29, spp=spp)
    for flags in (0, L.FLAG_COUNT):
        got, _ = rayhs_amd.ray_trace_file(scene, width=41, height=29, spp=spp, flags=flags)
        assert np.array_equal(got, want), flags
    library_env("RAYHS_NO_WAVE_RESOLVE")  # policy switch: force the buffer path
    got, _ = rayhs_amd.ray_trace_file(scene, width=41, height=29, spp=spp)
    assert np.array_equal(got, want)


def test_exhaustive_mode_visits_exactly_the_reference_node_set():
    # under exhaustive traversal the work counters equal the oracle's
    O.lib().orc_set_shadow_exhaustive(1)
    try:
        for scene in ("scenes/cornell_96.json", "scenes/dragon_superlow_96.json", "scenes/transform_64.json"):
            want, ost = oracle_image(scene)
            got, st = rayhs_amd.ray_trace_file(scene, flags=L.FLAG_COUNT | L.FLAG_EXHAUSTIVE)
            assert np.array_equal(got, want)
            for k in ost:
                assert st[k] == ost[k], (scene, k)
    finally:
        O.lib().orc_set_shadow_exhaustive(0)


def test_fma_build_stays_within_tolerance_away_from_exact_ties():
    # the opt-in FMA kernels differ in the last bits; pixels may flip only at discontinuities
    for scene in ("scenes/dragon_superlow_96.json", "scenes/cornell_96.json"):
        want, _ = rayhs_amd.ray_trace_file(scene)
        got, _ = rayhs_amd.ray_trace_file(scene, flags=L.FLAG_FMA)
        bad = (np.abs(got - want) > TOL).any(axis=2)
        assert bad.sum() <= 2, int(bad.sum())
        assert np.median(np.abs(got - want)) < 1e-12


# ------------------------------------------------------------------ edge cases
def _scene_json(objects, lights, w=16, h=12, depth=2, cam_z=-2.0, proj="perspective"):
    return json.dumps({"width": w, "height": h, "maxDepth": depth,
                       "camera": {"position": {"x": 0, "y": 0, "z": cam_z}, "target": {"x": 0, "y": 0, "z": 0}, "up": {"x": 0, "y": 1, "z": 0},
                                  "projection": {"type": proj, "fovy": 0.9272952180016123, "width": 2, "height": 2, "near": 0.1}},
                       "scene": {"objects": objects, "lights": lights}})


def _same(a, b):
    return a.shape == b.shape and bool(((a == b) | (np.isnan(a) & np.isnan(b))).all())


def _both(text, tmp_path, **kw):
    """Oracle image + the image of the PRODUCTION kernel; the counting kernel supplies the work counters and must
    produce the same image bit for bit (so every caller's comparison with the oracle covers both kernels)."""
    p = tmp_path / "s.json"
    p.write_text(text)
    want, ost = oracle_image(str(p), **kw)
    counted, st = rayhs_amd.ray_trace_file(str(p), flags=COUNTING, **kw)
    got, _ = rayhs_amd.ray_trace_file(str(p), flags=PRODUCTION, **kw)
    assert _same(got, counted), ("production and counting kernels differ", int((got != counted).any(axis=2).sum()))
    return got, want, st, ost


SPH = {"geometry": {"type": "sphere", "center": {"x": 0.1, "y": 0, "z": 1}, "radius": 0.6},
       "material": {"type": "plastic", "ior": 1.7, "cd": {"type": "flat", "color": {"r": 1, "g": 0.5, "b": 0.2}}}}
FLOOR = {"geometry": {"type": "plane", "point": {"x": 0, "y": -1, "z": 0}, "normal": {"x": 0, "y": 1, "z": 0}, "tangent": {"x": 1, "y": 0, "z": 0}},
         "material": {"type": "plastic", "ior": 2, "cd": {"type": "checker", "color1": {"r": 0, "g": 0, "b": 0}, "color2": {"r": 2, "g": 2, "b": 2}, "size": 0.5}}}
GLASS = {"geometry": {"type": "sphere", "center": {"x": -0.5, "y": -0.4, "z": 0.2}, "radius": 0.3}, "material": {"type": "transparent", "ior": 1.5}}
PT = {"type": "point", "position": {"x": 0, "y": 0.9, "z": 0.2}, "color": {"r": 90, "g": 90, "b": 90}, "radius": 0.1}
DIR = {"type": "directional", "direction": {"x": 0.3, "y": 1, "z": -0.5}, "color": {"r": 1, "g": 1, "b": 1}}


def test_empty_scene_is_black_and_casts_only_primary_rays(tmp_path):
    got, want, st, _ = _both(_scene_json([], []), tmp_path)
    assert np.array_equal(got, want) and not got.any()
    assert st["rays"] == st["rays_primary"] == 16 * 12 and st["nodes_tested"] == 0


def test_no_lights_max_depth_zero_and_one_pixel(tmp_path):
    got, want, st, ost = _both(_scene_json([SPH, FLOOR, GLASS], []), tmp_path)
    assert np.array_equal(got, want) and st["rays_shadow"] == 0
    got, want, st, ost = _both(_scene_json([SPH, FLOOR, GLASS], [PT, DIR], depth=0), tmp_path)
    assert np.array_equal(got, want) and st["rays_reflect"] == st["rays_refract"] == st["rays_refract_exit"] == 0
    got, want, *_ = _both(_scene_json([SPH, FLOOR], [PT], w=1, h=1), tmp_path)
    assert got.shape == (1, 1, 3) and np.array_equal(got, want)


@pytest.mark.parametrize("w,h", [(13, 9), (64, 1), (1, 40), (9, 17)])
def test_ragged_sizes_not_multiples_of_the_8x8_tile(tmp_path, w, h):
    got, want, st, _ = _both(_scene_json([SPH, FLOOR, GLASS], [PT, DIR], w=w, h=h), tmp_path)
    assert np.array_equal(got, want) and st["pixels"] == w * h


@pytest.mark.parametrize("fovy", [0.9272952180016123, 1e-95, -0.7])
@pytest.mark.parametrize("proj", ["perspective", "orthographic"])
def test_primary_rays_with_and_without_the_short_divisions(tmp_path, fovy, proj):
    # ray_from_pixel divides and normalises in the short forms of short_div.h when the launcher finds the camera constants in range
    # (DCamera::short_ok) and in the compiler's full sequences otherwise: an ordinary field of view; one whose f = 0.5 h / (tan 0.5 * fovy) lies
    # beyond 2^300 (a telephoto of all but parallel rays: the full forms); a negative one (the image mirrored: in range).  Even sizes put a pixel
    # boundary, odd ones a pixel centre on the optical axis: with one sample per pixel the numerator of the centre column / row is exactly +0.
    big = {"geometry": {"type": "sphere", "center": {"x": 0.5, "y": -0.5, "z": 6}, "radius": 5}, "material": SPH["material"]}   # (an orthographic camera's rays are a pixel apart)
    for w, h in ((16, 12), (15, 11)):
        jd = json.loads(_scene_json([SPH, FLOOR, GLASS, big], [PT, DIR], w=w, h=h, proj=proj))
        jd["camera"]["projection"]["fovy"] = fovy
        got, want, st, ost = _both(json.dumps(jd), tmp_path)
        assert _same(got, want) and st["rays"] == ost["rays"], (w, h)
        assert np.isfinite(want).all() and (want > 0).any()


def test_deep_recursion_orthographic_camera_and_directional_light(tmp_path):
    mirror = {"geometry": {"type": "plane", "point": {"x": 0, "y": 0, "z": 2}, "normal": {"x": 0, "y": 0, "z": -1}, "tangent": {"x": 1, "y": 0, "z": 0}},
              "material": {"type": "mirror", "ior": 0.1}}
    back = {"geometry": {"type": "plane", "point": {"x": 0, "y": 0, "z": -3}, "normal": {"x": 0, "y": 0, "z": 1}, "tangent": {"x": 1, "y": 0, "z": 0}},
            "material": {"type": "mirror", "ior": 0.1}}
    got, want, st, ost = _both(_scene_json([mirror, back, SPH, GLASS], [DIR, PT], depth=7, w=24, h=16), tmp_path)
    assert np.array_equal(got, want) and st["rays"] == ost["rays"] and st["rays_reflect"] > 24 * 16 * 3
    got, want, *_ = _both(_scene_json([SPH, FLOOR, GLASS], [DIR], proj="orthographic", w=24, h=16), tmp_path)
    assert np.array_equal(got, want)


def test_mesh_with_empty_subtrees_and_emitter_inside_shadow_path(tmp_path):
    cone = {"geometry": {"type": "mesh", "fileName": "data/cone.obj", "transform": {"type": "sequence", "transforms": [
        {"type": "scale", "vector": {"x": 0.4, "y": 0.4, "z": 0.4}}, {"type": "rotateX", "angle": 0.4}, {"type": "translate", "vector": {"x": 0, "y": -0.2, "z": 0.8}}]}},
        "material": {"type": "diffuse", "cd": {"type": "flat", "color": {"r": 1, "g": 1, "b": 1}}}}
    bulb = {"geometry": {"type": "sphere", "center": {"x": 0, "y": 0.9, "z": 0.2}, "radius": 0.15}, "material": {"type": "emmit", "ce": {"r": 1, "g": 1, "b": 1}}}
    got, want, st, ost = _both(_scene_json([cone, FLOOR, bulb], [PT], w=40, h=30), tmp_path)
    assert np.array_equal(got, want) and st["rays"] == ost["rays"]


def test_two_identical_meshes_tie_goes_to_the_first_object(tmp_path):
    def cube(rgb):
        return {"geometry": {"type": "mesh", "fileName": "data/cube.obj", "transform": {"type": "translate", "vector": {"x": 0, "y": 0, "z": 1}}},
                "material": {"type": "diffuse", "cd": {"type": "flat", "color": dict(zip("rgb", rgb))}}}
    got, want, *_ = _both(_scene_json([cube((1, 0, 0)), cube((0, 1, 0))], [DIR], w=24, h=24), tmp_path)
    assert np.array_equal(got, want)
    hit = got[12, 12]
    assert hit[0] > 0 and hit[1] == 0           # red (object 0) wins every exact tie (RayHs.hs:67-71)


# --------------------------------------------------------------- output formats
def test_output_formats_f32_and_toIntC_i16():
    r = build_rendering("scenes/cornell_96.json")
    for spp in (1, 2, 16):   # stored by the lane that traced it; folded by one lane per pixel; folded by one lane per pixel and CHANNEL (spp >= 4)
        f64, _ = r.ray_trace(spp=spp, out_format=L.OUT_F64)
        f32, _ = r.ray_trace(spp=spp, out_format=L.OUT_F32)
        i16, _ = r.ray_trace(spp=spp, out_format=L.OUT_I16)
        assert np.array_equal(f32, f64.astype(np.float32)), spp
        want = np.array([O.lib().orc_to_int_c(float(v)) for v in f64.ravel()], dtype=np.int16).reshape(f64.shape)
        assert np.array_equal(i16, want), spp
    r.close()


def test_cli_writes_the_same_ppm_bytes_as_the_reference_format(tmp_path):
    out = tmp_path / "c.ppm"
    r = rayhs_amd.run_cli([f"-o{out}", "scenes/cornell_96.json"], cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert r.stdout == f"Loading scene from scenes/cornell_96.json...\nRendering...\nDone! Output written to {out}\n"
    ref = tmp_path / "ref.ppm"
    O.write_ppm(str(ref), np.load(os.path.join(GOLD, "cornell_96.npy")))
    assert out.read_bytes() == ref.read_bytes()
    # default output name and the `-o FILE SCENE` spelling of BASELINE.json
    r = rayhs_amd.run_cli(["-o", str(tmp_path / "d.ppm"), os.path.join(ROOT, "scenes/transform_64.json")], cwd=ROOT)
    assert r.returncode == 0 and (tmp_path / "d.ppm").read_bytes().startswith(b"P3\n64 64\n255\n")
    r = subprocess.run([L.CLI_PATH, os.path.join(ROOT, "scenes/transform_64.json")], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode != 0 or os.path.exists(tmp_path / "out.ppm")   # meshes resolve against cwd (Mesh.hs:218-221)


# ------------------------------------------------ stripes / device entry point
def test_stripes_reassemble_to_the_full_frame():
    r = build_rendering("scenes/dragon_superlow_96.json")
    full, _ = r.ray_trace()
    for world, stripe in ((2, 8), (3, 5), (8, 8)):
        frame = np.zeros_like(full)
        for rank in range(world):
            part, _ = r.ray_trace(stripe_rows=stripe, stripe_rank=rank, stripe_count=world)
            rows = tiles.stripe_rows_of(rank, world, 96, stripe)
            assert part.shape[0] == len(rows)
            frame[rows] = part
        assert np.array_equal(frame, full)
    part, _ = r.ray_trace(row0=30, row1=50)
    assert np.array_equal(part, full[30:50])
    r.close()


def test_device_entry_point_renders_into_a_torch_tensor_on_its_stream():
    import torch
    r = build_rendering("scenes/cornell_96.json")
    want, _ = r.ray_trace(out_format=L.OUT_F32)
    out = torch.zeros((96, 96, 3), dtype=torch.float32, device="cuda:0")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):  # back-to-back enqueues, no host sync in between
            r.ray_trace_device(out.data_ptr(), s.cuda_stream, out_format=L.OUT_F32)
    s.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    r.close()


def test_one_handle_alternating_between_two_streams_and_the_host_entry_point():
    # A handle has ONE launch workspace and ONE tile-cost / order buffer.  Launches enqueued on different
    # non-blocking streams (and through rayhs_render's NULL stream) would share them if they overlapped; the
    # library chains them with an event (include/rayhs.h, "Streams").  Every frame must be the single-stream image.
    import torch
    r = build_rendering("scenes/dragon_low_1080p.json")
    w, h = r.width, r.height
    want, _ = r.ray_trace(spp=2, out_format=L.OUT_F32)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = [torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0") for _ in range(6)]
    torch.cuda.synchronize()
    for i, o in enumerate(outs):                 # no host sync between the enqueues
        st = (s1, s2)[i % 2]
        r.ray_trace_device(o.data_ptr(), st.cuda_stream, spp=2, out_format=L.OUT_F32)
    mixed, _ = r.ray_trace(spp=2, out_format=L.OUT_F32)      # NULL stream right behind them
    torch.cuda.synchronize()
    assert np.array_equal(mixed, want)
    for i, o in enumerate(outs):
        assert np.array_equal(o.cpu().numpy(), want), i
    r.close()


def test_frame_wider_than_the_tile_limit_is_refused_not_wrapped():
    # tiles_x * tile rows is formed in 64 bits: 2^27 x 16 tiles would wrap a 32-bit product to a negative count
    r = build_rendering("scenes/cornell_96.json", width=1 << 30, height=128)
    with pytest.raises(L.RayhsError) as e:
        r.ray_trace_device(0x1000, 0, out_format=L.OUT_F32)
    assert e.value.code == L.ERR_UNSUPPORTED and "frame too large" in e.value.message
    r.close()


def test_empty_row_selection_records_the_callers_events():
    import torch
    r = build_rendering("scenes/cornell_96.json")
    out = torch.zeros(16, dtype=torch.float32, device="cuda:0")
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    e1.record(s)
    torch.cuda.synchronize()
    r.ray_trace_device(out.data_ptr(), s.cuda_stream, events=(e0.cuda_event, e1.cuda_event), row0=10, row1=10, out_format=L.OUT_F32)
    torch.cuda.synchronize()
    assert 0.0 <= e0.elapsed_time(e1) < 100.0
    img, st = r.ray_trace(row0=10, row1=10)
    assert img.shape == (0, 96, 3)
    r.close()


# --------------------------------------- several GPUs from one process (rayhs_multi)
@pytest.mark.parametrize("policy", ["default", "RAYHS_LPT_STRIPES", "RAYHS_ROUND_ROBIN_STRIPES"])
def test_multi_device_frame_equals_the_single_device_frame_one_gpu_rehearsal(policy, library_env):
    if policy != "default":
        library_env(policy)                  # policy switches: force the cost-sorted / the interleaved ownership
    # the N-GPU path of the library (stripe lists balanced by the static estimate, per-device streams, peer push into
    # the frame on the first device) with every "device" being GPU 0 -- what the driver's 8-GPU node runs with 8 GPUs
    rd = RenderDesc("scenes/dragon_low_1080p.json")
    single = Rendering(rd)
    for fmt in (L.OUT_F32, L.OUT_I16):
        want, _ = single.ray_trace(spp=4, out_format=fmt)
        for devices in ([0, 0, 0], [0] * 8):
            m = rayhs_amd.MultiRendering(rd, devices)
            for _ in range(2):                               # first frame (static tile order), then the measured order
                got, st = m.ray_trace(spp=4, out_format=fmt)
                assert np.array_equal(got, want), (fmt, devices)
            shares, rows = m.shares()
            assert sum(rows) == rd.height and abs(sum(shares) - 1) < 1e-9
            assert max(shares) / (1.0 / len(devices)) < 1.08, shares      # cost-balanced: no device above 108 % of its fair share
            m.close()
    _assert_frame_matches_digest(single.ray_trace(spp=4, out_format=L.OUT_F32)[0], "dragon_low")
    single.close()


@pytest.mark.skipif(L.load().rayhs_device_count() < 2, reason="needs two physical GPUs (peer access, cross-device events)")
@pytest.mark.parametrize("create", ["serial", "one thread per device"])
def test_multi_device_on_two_physical_gpus_equals_the_single_device_frame(create, library_env):
    # what the one-GPU rehearsals above cannot execute: hipDeviceEnablePeerAccess, the peer writes of rayhs_scatter_stripes
    # into a frame that lives on devices[0] (the library's own, and a caller-owned d_out), and the cross-device
    # hipStreamWaitEvent on every device's `done` event.  Runs wherever the box shows >= 2 GPUs -- with the devices brought
    # up one after the other (the default) and on one thread each (RAYHS_MULTI_PARALLEL_CREATE, opt-in until this has passed).
    import torch
    if create != "serial":
        library_env("RAYHS_MULTI_PARALLEL_CREATE")
    n = min(L.load().rayhs_device_count(), 8)
    rd = RenderDesc("scenes/dragon_low_1080p.json")
    single = Rendering(rd)
    want, st1 = single.ray_trace(spp=4, out_format=L.OUT_F32, flags=COUNTING)
    for devices in ([0, 1], [1, 0], list(range(n))):
        m = rayhs_amd.MultiRendering(rd, devices)
        for _ in range(2):
            got, st = m.ray_trace(spp=4, out_format=L.OUT_F32)          # rayhs_multi_render: host buffer
            assert np.array_equal(got, want), devices
        got, st = m.ray_trace(spp=4, out_format=L.OUT_F32, flags=COUNTING)
        assert np.array_equal(got, want) and st["rays"] == st1["rays"] and st["pixels"] == st1["pixels"]
        # rayhs_multi_render_device: the frame in a caller-owned buffer on devices[0]
        with torch.cuda.device(devices[0]):
            frame = torch.zeros((rd.height, rd.width, 3), dtype=torch.float32, device=f"cuda:{devices[0]}")
            torch.cuda.synchronize()
            m.ray_trace_into(frame.data_ptr(), spp=4, out_format=L.OUT_F32)
            assert np.array_equal(frame.cpu().numpy(), want), devices
        m.close()
    _assert_frame_matches_digest(want, "dragon_low")
    single.close()


@pytest.mark.parametrize("w,h,stripe,devices", [(77, 53, 5, [0, 0]), (64, 20, 8, [0] * 5), (33, 7, 8, [0, 0, 0]), (96, 96, 16, [0])])
def test_multi_device_ragged_frames_row_ranges_and_counters(w, h, stripe, devices):
    rd = RenderDesc("scenes/dragon_superlow_96.json")
    single = Rendering(rd, width=w, height=h)
    m = rayhs_amd.MultiRendering(rd, devices, width=w, height=h)
    want, st1 = single.ray_trace(spp=2, flags=COUNTING)
    got, st = m.ray_trace(spp=2, stripe_rows=stripe, flags=COUNTING)
    assert np.array_equal(got, want)
    for k in ("rays", "rays_shadow", "shaded_hits", "pixels", "shadow_skipped"):
        assert st[k] == st1[k], k
    got, _ = m.ray_trace(spp=2, stripe_rows=stripe)              # the production kernels
    assert np.array_equal(got, want)
    if h > 12:
        part, _ = m.ray_trace(spp=2, stripe_rows=stripe, row0=3, row1=h - 4)
        assert np.array_equal(part, want[3:h - 4])
    m.close()
    single.close()


def test_device_prepare_is_optional_and_idempotent():
    # rayhs_device_prepare only moves start-up cost (runtime, context, code object) to where the caller wants it
    lib = L.load()
    assert lib.rayhs_device_prepare(0) == L.OK and lib.rayhs_device_prepare(0) == L.OK
    assert lib.rayhs_device_prepare(lib.rayhs_device_count()) == L.ERR_INVALID
    img, _ = rayhs_amd.ray_trace_file("scenes/cornell_96.json")
    assert np.array_equal(img, np.load(os.path.join(GOLD, "cornell_96.npy")))


def test_cli_gpus_option_writes_the_same_file(tmp_path):
    a, b = tmp_path / "one.ppm", tmp_path / "multi.ppm"
    r = rayhs_amd.run_cli([f"-o{a}", "--spp=2", "scenes/dragon_superlow_96.json"], cwd=ROOT)
    assert r.returncode == 0, r.stderr
    r = rayhs_amd.run_cli([f"-o{b}", "--spp=2", "--devices=0,0,0", "scenes/dragon_superlow_96.json"], cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert a.read_bytes() == b.read_bytes()
    r = rayhs_amd.run_cli([f"-o{b}", "--gpus=1", "scenes/cornell_96.json"], cwd=ROOT)
    assert r.returncode == 0 and b.read_bytes().startswith(b"P3\n96 96\n255\n")
    r = rayhs_amd.run_cli([f"-o{b}", "--gpus=99", "scenes/cornell_96.json"], cwd=ROOT)
    assert r.returncode == 1 and "--gpus" in r.stderr


def _ppm_values(path):
    tok = open(path).read().split()
    assert tok[0] == "P3" and tok[3] == "255"
    w, h = int(tok[1]), int(tok[2])
    return np.array(tok[4:], dtype=np.int64).reshape(h, w, 3)


def test_the_commands_of_the_reference_test_scripts(tmp_path):
    """The reference's whole test strategy is two shell scripts that render its shipped scenes at their own sizes
    (testAll.sh:4-11: texture, outScene, cornellBox, dragon; test.sh:4: outScene to the default out.ppm).  The same
    command lines on the drop-in executable, each output file held against the oracle's frame written by the oracle's
    own P3 writer."""
    for out, scene, exact in (("outScene.ppm", "data/outScene.json", False), ("cornell.ppm", "data/cornellBox.json", True),
                              ("dragon.ppm", "data/dragon.json", True)):
        dst = tmp_path / out
        r = rayhs_amd.run_cli([f"-o{dst}", scene], cwd=ROOT)
        assert r.returncode == 0, r.stderr
        assert r.stdout == f"Loading scene from {scene}...\nRendering...\nDone! Output written to {dst}\n"
        img, _ = O.render_file(os.path.join(ROOT, scene), base_dir=ROOT)
        ref = tmp_path / ("ref_" + out)
        O.write_ppm(str(ref), img)
        if exact:
            assert dst.read_bytes() == ref.read_bytes(), scene
        else:   # checker spheres: atan / acos of the device against glibc's; a channel may round the other way
            a, b = _ppm_values(dst), _ppm_values(ref)
            assert a.shape == b.shape and np.abs(a - b).max() <= 1 and (a != b).mean() < 1e-5, scene
    # data/texture.json:59,87 names two images the reference repository does not hold: the reference dies on the
    # first readFile (Bitmap.hs:31-37) after its two progress lines, and so does the drop-in
    r = rayhs_amd.run_cli([f"-o{tmp_path / 'texture.ppm'}", "data/texture.json"], cwd=ROOT)
    assert r.returncode == 1 and r.stdout == "Loading scene from data/texture.json...\nRendering...\n"
    assert "data/checkerboard.ppm: openFile: does not exist" in r.stderr and not (tmp_path / "texture.ppm").exists()
    # test.sh: no -o, the frame goes to ./out.ppm; mesh paths resolve against the working directory
    os.symlink(os.path.join(ROOT, "data"), tmp_path / "data")
    r = subprocess.run([L.CLI_PATH, "data/outScene.json"], cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.endswith("Done! Output written to out.ppm\n")
    assert (tmp_path / "out.ppm").read_bytes() == (tmp_path / "outScene.ppm").read_bytes()


# ------------------------------------------------- BASELINE sizes: properties
@pytest.mark.parametrize("scene,spp", [("scenes/cornell_1080p_d0.json", 1), ("scenes/dragon_low_1080p.json", 4)])
def test_full_size_properties(scene, spp):
    r = build_rendering(scene)
    w, h = r.width, r.height
    full, st = r.ray_trace(spp=spp, flags=L.FLAG_COUNT)
    assert st["pixels"] == w * h and st["rays_primary"] == w * h * spp
    assert st["shaded_hits"] <= st["rays_primary"] + st["rays_reflect"] + st["rays_refract"]
    assert st["rays_shadow"] % 1 == 0 and np.isfinite(full).all()
    # (a) exhaustive traversal (the reference's visit set) gives the identical image
    exh, st2 = r.ray_trace(spp=spp, flags=L.FLAG_COUNT | L.FLAG_EXHAUSTIVE)
    assert np.array_equal(exh, full)
    assert st2["tris_tested"] >= st["tris_tested"] and st2["rays"] == st["rays"]
    # (b) two interleaved stripe sets reassemble the frame bit for bit
    frame = np.empty_like(full)
    for rank in range(2):
        part, _ = r.ray_trace(spp=spp, stripe_rows=8, stripe_rank=rank, stripe_count=2)
        frame[tiles.stripe_rows_of(rank, 2, h, 8)] = part
    assert np.array_equal(frame, full)
    # (c) a band of rows checked against the oracle itself (bit-exact)
    y0 = h // 2 - 2
    want, _ = oracle_image(scene, spp=spp, row0=y0, row1=y0 + 4)
    assert np.array_equal(full[y0:y0 + 4], want)
    r.close()


def test_4k_frame_every_tile_lands_where_it_should():
    # 3840x2160 = 480 x 270 tiles: the refill maps a tile index to its row and column with a float reciprocal and
    # one correction step, and frames 2.. are worked in measured-cost order; bands of the frame against the oracle
    # and an 8-rank stripe decomposition against the whole frame
    scene = "scenes/dragon_4k.json"
    r = build_rendering(scene)
    h = r.height
    frames = [r.ray_trace(spp=1)[0] for _ in range(2)]
    assert np.array_equal(frames[0], frames[1])
    for y0 in (0, 1079, 2152):
        want, _ = oracle_image(scene, spp=1, row0=y0, row1=y0 + 8)
        assert np.array_equal(frames[1][y0:y0 + 8], want), y0
    whole = np.empty_like(frames[0])
    for rank in range(8):
        part, _ = r.ray_trace(spp=1, stripe_rows=8, stripe_rank=rank, stripe_count=8)
        whole[tiles.stripe_rows_of(rank, 8, h, 8)] = part
    assert np.array_equal(whole, frames[0])
    r.close()


def test_headline_scene_band_against_oracle():
    scene = "scenes/dragon_1080p.json"
    r = build_rendering(scene)
    part, _ = r.ray_trace(spp=16, row0=600, row1=602)
    want, _ = oracle_image(scene, spp=16, row0=600, row1=602)
    assert np.array_equal(part, want)
    r.close()


SECOND = json.load(open(os.path.join(GOLD, "second.json")))


@pytest.mark.parametrize("name", [n for n, m in SECOND.items() if m["rows"] is not None])
def test_production_kernel_equals_the_second_restatement_on_baseline_windows(name):
    # vectors of oracle/second/ (plain Python written from the Haskell alone; they did not come from the C oracle):
    # pixel windows of configs[1]..[4] at their own size and sample count
    m = SECOND[name]
    want = np.load(os.path.join(GOLD, f"second_{name}.npy"))
    r = build_rendering(m["scene"])
    for i, y in enumerate(m["rows"]):
        row, _ = r.ray_trace(spp=m["spp"], seed=m["seed"], row0=y, row1=y + 1)
        assert np.array_equal(row[0][m["cols"]], want[i]), (name, y)
    r.close()


# ---------------------------------------- BASELINE configs: WHOLE frames, production kernels
# tests/golden/frame_digests.json holds sha256 + exact per-row sums of the oracle's full frames of configs[1]..[4]
# (tests/golden/make_frame_digests.py, run once in the build container: the 4K / 64 spp frame is 2.5 G rays).
DIGESTS = json.load(open(os.path.join(GOLD, "frame_digests.json")))


def _assert_frame_matches_digest(img, name):
    import hashlib
    import math
    d = DIGESTS[name]
    assert img.shape == (d["height"], d["width"], 3)
    key = "sha256_f64" if img.dtype == np.float64 else "sha256_f32"
    if hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == d[key]:
        return
    bad = [y for y in range(img.shape[0]) if math.fsum(img[y].astype(np.float64).ravel().tolist()).hex() != d["row_fsum"][y]] \
        if img.dtype == np.float64 else "(fp32 frame: rows not located)"
    raise AssertionError(f"{name}: frame differs from the oracle's digest; rows {bad if isinstance(bad, str) else bad[:20]}")


@pytest.mark.parametrize("name", ["cornell", "dragon_low", "dragon"])
def test_whole_baseline_frames_equal_the_oracle_digests(name):
    # configs[1]..[3] at their own size and sample count, every pixel: the production kernel's first frame (scan
    # order), its second (tiles re-ordered by measured cost), the fp32 frame bench.py renders, and the ray counts
    d = DIGESTS[name]
    r = build_rendering(d["scene"])
    for _ in range(2):
        img, _ = r.ray_trace(spp=d["spp"], seed=d["seed"])
        _assert_frame_matches_digest(img, name)
    f32, _ = r.ray_trace(spp=d["spp"], seed=d["seed"], out_format=L.OUT_F32)
    _assert_frame_matches_digest(f32, name)
    _, st = r.ray_trace(spp=d["spp"], seed=d["seed"], flags=COUNTING)
    for k in ("rays_primary", "rays_reflect", "rays_refract", "rays_refract_exit", "rays_shadow", "shaded_hits", "rays"):
        assert st[k] == d["stats"][k], k
    r.close()


def test_headline_frame_whole_against_the_live_oracle_and_the_exhaustive_kernel():
    # configs[3] (the bench default): all 1920 x 1080 pixels at 16 spp against the oracle run here and now
    # (seconds on the box's host cores), and against the exhaustive kernel that walks the reference's own tree
    scene = "scenes/dragon_1080p.json"
    want, ost = oracle_image(scene, spp=16)
    _assert_frame_matches_digest(want, "dragon")           # the committed digest is this oracle's
    r = build_rendering(scene)
    got, _ = r.ray_trace(spp=16)
    assert np.array_equal(got, want), int((got != want).any(axis=2).sum())
    exh, st = r.ray_trace(spp=16, flags=L.FLAG_COUNT | L.FLAG_EXHAUSTIVE)
    assert np.array_equal(exh, want)
    assert st["rays"] == ost["rays"]
    r.close()


def test_4k_64spp_whole_frame_bands_and_eight_rank_stripes():
    # configs[4] at its own sample count: the whole 3840 x 2160 / 64 spp frame against the committed oracle digest,
    # 16 two-row bands against the live oracle, and the 8-rank stripe decomposition reassembled
    d = DIGESTS["dragon4k"]
    scene, spp = d["scene"], d["spp"]
    assert spp == 64 and (d["width"], d["height"]) == (3840, 2160)
    r = build_rendering(scene)
    h = r.height
    full, _ = r.ray_trace(spp=spp)
    _assert_frame_matches_digest(full, "dragon4k")
    for i in range(16):
        y0 = min(h - 2, int((i + 0.5) * h / 16))
        want, _ = oracle_image(scene, spp=spp, row0=y0, row1=y0 + 2)
        assert np.array_equal(full[y0:y0 + 2], want), y0
    whole = np.empty_like(full)
    for rank in range(8):
        part, _ = r.ray_trace(spp=spp, stripe_rows=8, stripe_rank=rank, stripe_count=8)
        whole[tiles.stripe_rows_of(rank, 8, h, 8)] = part
    assert np.array_equal(whole, full)
    r.close()


def test_bench_two_ranks_rehearsal_on_one_gpu():
    # the N > 1 path of bench.py (stripes, gather, assembly, counter reduction) with two ranks sharing
    # the one GPU of this box over gloo; the driver runs the real thing over RCCL on 8 GPUs
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RAYHS_BENCH_ONE_DEVICE="1", RAYHS_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--workload", "dragon_low", "--cpu-seconds", "0", "--verify"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["verified_against_single_rank"] is True
    # the line says what the collective layer saw: backend, ranks of the process group, every rank's kernel time, the gather by itself
    d = out["distributed"]
    assert d["backend"] == "gloo" and d["ranks_seen"] == 2 and len(d["kernel_ms_per_rank"]) == 2
    assert all(t > 0 for t in d["kernel_ms_per_rank"]) and d["gather_ms"] > 0
    assert abs(d["kernel_ms_per_rank"][0] - out["kernel_ms_rank0"]) < 1e-9
    assert out["metric"] == "Mrays/s" and out["value"] > 0 and out["scaling"] == "strong"
    assert out["rays_per_frame"] > 1920 * 1080 * 4
    # the extra leg on the 4K / 64 spp frame (BASELINE configs[4]): two ranks' stripes gathered, whole frame == oracle digest
    k4 = out["north_star_4k"]
    assert k4["whole_frame_sha256_equals_oracle_digest"] is True and k4["ms_per_step"] > 0 and k4["rays_per_frame"] > 3840 * 2160 * 64


def test_timed_entry_point_records_events_around_the_trace_kernel():
    import torch
    r = build_rendering("scenes/dragon_superlow_96.json")
    want, _ = r.ray_trace(spp=4, out_format=L.OUT_F32)
    out = torch.zeros((96, 96, 3), dtype=torch.float32, device="cuda:0")
    s = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    e1.record(s)   # torch creates the hipEvent on first record
    r.ray_trace_device(out.data_ptr(), s.cuda_stream, events=(e0.cuda_event, e1.cuda_event), spp=4, out_format=L.OUT_F32)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)
    assert 0.0 < e0.elapsed_time(e1) < 1000.0
    r.close()


def test_large_mesh_uses_the_wide_stack_kernel(tmp_path):
    # > 65535 BVH nodes: 16-bit stack entries no longer fit, the 32-bit-stack kernel is selected
    n = 280                                        # 2 * 279^2 = 155,682 triangles
    xs = np.linspace(-1.0, 1.0, n)
    zz = 0.15 * np.sin(3.1 * xs)[:, None] * np.cos(2.3 * xs)[None, :]
    lines = []
    for j in range(n):
        for i in range(n):
            lines.append(f"v {xs[i]:.6f} {xs[j]:.6f} {zz[j, i] + 1.5:.6f}")
    lines.append("vn 0 0 -1")
    for j in range(n - 1):
        for i in range(n - 1):
            a, b, c, d = j * n + i + 1, j * n + i + 2, (j + 1) * n + i + 1, (j + 1) * n + i + 2
            lines.append(f"f {a}//1 {b}//1 {d}//1")
            lines.append(f"f {a}//1 {d}//1 {c}//1")
    obj = tmp_path / "sheet.obj"
    obj.write_text("\n".join(lines) + "\n")
    mesh = {"geometry": {"type": "mesh", "fileName": str(obj), "transform": {"type": "rotateY", "angle": 0.3}},
            "material": {"type": "plastic", "ior": 1.6, "cd": {"type": "flat", "color": {"r": 1, "g": 0.7, "b": 0.4}}}}
    got, want, st, ost = _both(_scene_json([mesh, GLASS], [PT, DIR], w=40, h=24, depth=2), tmp_path)
    assert np.array_equal(got, want) and st["rays"] == ost["rays"]
    p = tmp_path / "s.json"
    fast, _ = rayhs_amd.ray_trace_file(str(p))       # non-counting run -> the wide kernel itself
    assert np.array_equal(fast, want)


def _wall(point, normal, colour):
    return {"geometry": {"type": "plane", "point": dict(zip("xyz", point)), "normal": dict(zip("xyz", normal)), "tangent": {"x": 0.3, "y": 0.2, "z": 1}},
            "material": {"type": "diffuse", "cd": {"type": "flat", "color": dict(zip("rgb", colour))}}}


@pytest.mark.parametrize("offset", [0.0, 1e-12, -1e-12, 1e-7, -1e-7, 3e-5, -3e-5, 1e-3, -1e-3])
def test_point_light_on_or_next_to_a_wall(tmp_path, offset):
    # Shadow queries settle "is the wall hit in front of the light?" from an approximate quotient when the answer is
    # clear and divide exactly otherwise.  Lights sitting on a wall, an ulp in front of it, or just behind it put every
    # shadow ray of the frame into the ambiguous band; the image must still be the oracle's, bit for bit.
    walls = [_wall((0, -1, 0), (0, 1, 0), (1, 1, 1)), _wall((0, 1, 0), (0, -1, 0), (1, 1, 1)), _wall((0, 0, 1.5), (0, 0, -1), (0.9, 0.9, 0.9)),
             _wall((-1, 0, 0), (1, 0.05, 0.02), (1, 0.2, 0.2)), _wall((1, 0, 0), (-1, 0, 0), (0.2, 1, 0.2))]
    lights = [{"type": "point", "position": {"x": 0.2, "y": 1.0 + offset, "z": 0.4}, "color": {"r": 20, "g": 20, "b": 20}, "radius": 0.1},
              {"type": "point", "position": {"x": 1.0 + offset, "y": -0.3, "z": 0.9}, "color": {"r": 5, "g": 9, "b": 20}, "radius": 0.1},
              {"type": "point", "position": {"x": -0.2, "y": 0.1, "z": 1.5 + offset}, "color": {"r": 20, "g": 9, "b": 5}, "radius": 0.2}]
    got, want, st, ost = _both(_scene_json(walls + [SPH], lights, w=48, h=32, depth=2), tmp_path)
    assert np.array_equal(got, want)
    assert st["rays"] == ost["rays"]


def test_feedback_tile_order_never_changes_a_pixel(library_env):
    # frame 1 runs in scan order and measures chunk times, frames 2 and 3 are worked slowest-tile-first
    # (rayhs_order_tiles); a frame with the ordering switched off must be the same image, bit for bit
    r = build_rendering("scenes/dragon_low_1080p.json")
    frames = [r.ray_trace(spp=4, out_format=L.OUT_F32)[0] for _ in range(3)]
    assert np.array_equal(frames[0], frames[1]) and np.array_equal(frames[0], frames[2])
    band = [r.ray_trace(spp=4, out_format=L.OUT_F32, row0=400, row1=720, stripe_rows=8, stripe_rank=1, stripe_count=2)[0] for _ in range(2)]
    assert np.array_equal(band[0], band[1])          # a new geometry starts over in scan order
    rows = tiles.stripe_rows_of(1, 2, 1080, 8, row0=400, row1=720)
    assert np.array_equal(band[0], frames[0][rows])
    library_env("RAYHS_SCAN_ORDER")
    plain, _ = r.ray_trace(spp=4, out_format=L.OUT_F32)
    assert np.array_equal(plain, frames[0])
    r.close()


def test_first_frame_static_order_is_the_same_image_and_close_to_steady_state(library_env):
    # the first frame of a handle is ordered by rayhs_static_costs (no measured chunk times yet): same pixels as the
    # scan-order frame and as the feedback-ordered frames; its kernel time stays near the steady state (measured:
    # within 6 % on configs[1]..[4], profiles/r02/first_frame.jsonl -- asserted loosely, it is a timing)
    from rayhs_amd import build_rendering as br
    firsts, steady = [], []
    for _ in range(3):
        r = br("scenes/dragon_low_1080p.json")
        img0, st0 = r.ray_trace(spp=4, out_format=L.OUT_F32)
        firsts.append(st0["kernel_ms"])
        for i in range(4):
            img, st = r.ray_trace(spp=4, out_format=L.OUT_F32)
            assert np.array_equal(img, img0)
            if i >= 1:
                steady.append(st["kernel_ms"])
        r.close()
    library_env("RAYHS_NO_STATIC_ORDER")
    r = br("scenes/dragon_low_1080p.json")
    scan, _ = r.ray_trace(spp=4, out_format=L.OUT_F32)
    r.close()
    assert np.array_equal(scan, img0)
    # (a timing on a possibly shared box: only a gross regression -- scan order costs 1.2-1.65x, an unordered or broken
    # first frame far more -- fails the test; the measured ratio, 1.00-1.13, is evidence: tools/first_frame.py, profiles/)
    assert sorted(firsts)[1] < 2.0 * sorted(steady)[len(steady) // 2], (firsts, steady)
    _assert_frame_matches_digest(img0, "dragon_low")


def _leading_tiles(order, share):
    """The tiles at the head of a work order: it is dealt in groups of 8 over 8 bands (rayhs_order_tiles), so the
    slowest `share` of the frame is the first `share` of every band."""
    n = len(order)
    cycles, rest = divmod(n, 64)
    lead, begin = [], 0
    for b in range(8):
        size = cycles * 8 + min(max(rest - 8 * b, 0), 8)
        lead.extend(order[begin:begin + int(size * share)])
        begin += size
    assert begin == n
    return set(int(t) for t in lead)


def test_first_frame_static_order_is_a_permutation_that_leads_with_the_slow_tiles():
    # not a timing: the order the first frame was GIVEN (static estimate -> rayhs_order_hist / rayhs_order_tiles) is read
    # back (rayhs_scene_capture_tile_order / rayhs_scene_tile_order) and must (a) hold every tile exactly once and (b) start
    # with the tiles the second frame's MEASURED order starts with -- a broken estimate or ordering kernel degrades to scan
    # order, where that overlap is the share itself (5 %)
    r = build_rendering("scenes/dragon_low_1080p.json")
    r.capture_tile_order(True)
    recs = []
    for _ in range(2):
        r.ray_trace(spp=4, out_format=L.OUT_F32)
        recs.append(r.tile_order())
    assert r.ray_trace(spp=4, flags=COUNTING, row0=0, row1=8)[1]["pixels"] == 1920 * 8 and len(r.tile_order()[0]) == 0  # (a counting run: scan order)
    r.ray_trace(spp=4, out_format=L.OUT_F32)
    small = np.zeros(16, dtype=np.int32)   # a buffer that cannot hold the order is refused, not overrun
    assert L.load().rayhs_scene_tile_order(r._h, small.ctypes.data_as(C.POINTER(C.c_int32)), 16, None) == L.ERR_INVALID and not small.any()
    r.close()
    n = 240 * 135
    assert recs[0][1] is True and recs[1][1] is False        # static estimate first, measured chunk times second
    for order, _ in recs:
        assert len(order) == n and np.array_equal(np.sort(order), np.arange(n))   # a permutation of the tiles
    static_lead, measured_lead = _leading_tiles(recs[0][0], 0.05), _leading_tiles(recs[1][0], 0.15)
    overlap = len(static_lead & measured_lead) / len(static_lead)
    assert overlap > 0.5, overlap


def test_environment_switches_are_read_once_until_the_host_asks_again(monkeypatch):
    # nothing on the launch path reads the environment (include/rayhs.h, rayhs_config_reload): a switch set after the library was loaded
    # is not seen until the host asks for the environment to be read again
    r = build_rendering("scenes/dragon_low_1080p.json")
    r.capture_tile_order(True)
    try:
        img0, _ = r.ray_trace(spp=1, out_format=L.OUT_F32)
        assert len(r.tile_order()[0]) == 240 * 135                 # ordered (static estimate: first frame of the handle)
        monkeypatch.setenv("RAYHS_SCAN_ORDER", "1")
        img1, _ = r.ray_trace(spp=1, out_format=L.OUT_F32)
        assert len(r.tile_order()[0]) == 240 * 135                 # still ordered: the switch has not been read
        L.load().rayhs_config_reload()
        img2, _ = r.ray_trace(spp=1, out_format=L.OUT_F32)
        assert len(r.tile_order()[0]) == 0                         # scan order now
        assert np.array_equal(img0, img1) and np.array_equal(img0, img2)
    finally:
        monkeypatch.delenv("RAYHS_SCAN_ORDER", raising=False)
        L.load().rayhs_config_reload()
        r.close()


def test_unknown_flag_bits_are_refused():
    r = build_rendering("scenes/cornell_96.json")
    with pytest.raises(L.RayhsError) as e:
        r.ray_trace(flags=0x100)
    assert e.value.code == L.ERR_INVALID and "unknown flag" in e.value.message
    r.close()


def test_very_deep_recursion_limit_runs_on_a_smaller_grid(tmp_path):
    # two facing mirrors: rays bounce until maxDepth; 14 doubles of continuation frame per level and lane make the
    # workspace outgrow 32-bit offsets unless the persistent grid shrinks (abi.cpp prepare_launch)
    def mirror(z, nz):
        return {"geometry": {"type": "plane", "point": {"x": 0, "y": 0, "z": z}, "normal": {"x": 0, "y": 0, "z": nz}, "tangent": {"x": 1, "y": 0, "z": 0}},
                "material": {"type": "mirror", "ior": 1.8}}
    ball = {"geometry": {"type": "sphere", "center": {"x": 0.3, "y": -0.2, "z": 0.5}, "radius": 0.25},
            "material": {"type": "plastic", "ior": 1.5, "cd": {"type": "flat", "color": {"r": 0.9, "g": 0.4, "b": 0.2}}}}
    text = _scene_json([mirror(2.0, -1), mirror(-3.0, 1), ball], [PT], w=64, h=64, depth=1500)
    got, want, st, ost = _both(text, tmp_path)
    assert np.array_equal(got, want) and st["rays"] == ost["rays"]
    fast, _ = rayhs_amd.ray_trace_file(str(tmp_path / "s.json"))
    assert np.array_equal(fast, want)


def test_nan_vertex_meshes_the_documented_deviation_is_pinned(tmp_path):
    # `rotate` about an axis with x = y = 0 makes the reference normalise the zero vector (orthonormal, src/Vec.hs:151-158):
    # every vertex of the mesh becomes NaN (the host loader reproduces that bit for bit, tests/test_host.py).
    # * >= 20 triangles: the reference's tree is Node NaN-box Empty Empty (no barycentre compares with a NaN split,
    #   src/KDTree.hs:79-90), never hit -- kernel, oracle and reference agree: the scene renders as if the mesh were absent.
    # * < 20 triangles: the tree is one Leaf, and triangleIntersection accepts EVERY ray with t = NaN (all its
    #   rejection tests are false for NaN, src/Mesh.hs:59-72); minimumBy then lets that hit replace the hits of earlier
    #   objects and lose to later ones (src/Geometry.hs:54-57).  The kernels treat a NaN triangle as never hit.
    #   This is the ONE documented deviation: pinned here as "renders as if the mesh were absent".
    def mesh(file):
        return {"geometry": {"type": "mesh", "fileName": file, "transform": {"type": "rotate", "axis": {"x": 0, "y": 0, "z": 1}, "angle": 0.7}},
                "material": {"type": "diffuse", "cd": {"type": "flat", "color": {"r": 1, "g": 1, "b": 1}}}}
    others = [SPH, FLOOR, GLASS]
    without, _ = oracle_image_text(_scene_json(others, [PT, DIR], w=24, h=16), tmp_path)
    for file, reference_agrees in (("data/torus.obj", True), ("data/cube.obj", False)):
        text = _scene_json([others[0], mesh(file)] + others[1:], [PT, DIR], w=24, h=16)
        (tmp_path / "n.json").write_text(text)
        for flags in (PRODUCTION, COUNTING):
            got, _ = rayhs_amd.ray_trace_file(str(tmp_path / "n.json"), flags=flags)
            assert _same(got, without), (file, flags)
        want, _ = oracle_image(str(tmp_path / "n.json"))
        assert _same(want, without) == reference_agrees, file      # the oracle follows the reference: NaN hits for the 12-triangle cube


def oracle_image_text(text, tmp_path, **kw):
    p = tmp_path / "o.json"
    p.write_text(text)
    return oracle_image(str(p), **kw)


# ---------------------------------------------------------------------- fuzz
from scene_fuzz import rand_scene as _rand_scene  # noqa: E402  (tests/scene_fuzz.py; shared with the CPU differential test of the two restatements)


_FUZZ_BASE = int(os.environ.get("RAYHS_FUZZ_SEED_BASE", "0"))   # (a soak over scenes no earlier soak has seen: RAYHS_FUZZ_SEED_BASE=100000)


@pytest.mark.parametrize("seed", range(int(os.environ.get("RAYHS_FUZZ_SEEDS", "24"))))  # RAYHS_FUZZ_SEEDS=400 for a long soak
def test_random_scenes_bit_exact(tmp_path, seed):
    rng = np.random.default_rng(1000 + _FUZZ_BASE + seed)
    jd = _rand_scene(rng)
    spp = int(rng.choice([1, 1, 2, 3]))
    p = tmp_path / "f.json"
    p.write_text(json.dumps(jd))
    want, ost = oracle_image(str(p), spp=spp)
    got, st = rayhs_amd.ray_trace_file(str(p), spp=spp, flags=L.FLAG_COUNT)
    same = (got == want) | (np.isnan(got) & np.isnan(want))     # NaN colours are legal (e.g. 0/0 in a degenerate scene)
    assert same.all(), (seed, np.abs(np.nan_to_num(got) - np.nan_to_num(want)).max(), int((~same).any(axis=2).sum()))
    assert st["rays"] == ost["rays"], seed
    r = build_rendering(str(p))                                  # the production kernels: first frame in scan order, second re-ordered
    for _ in range(2):
        img, _ = r.ray_trace(spp=spp)
        assert ((img == want) | (np.isnan(img) & np.isnan(want))).all(), seed
    r.close()


@pytest.mark.parametrize("seed", range(int(os.environ.get("RAYHS_FUZZ_SEEDS_SPP", "12"))))
def test_random_scenes_many_samples_per_pixel_bit_exact(tmp_path, seed):
    # the other end of the work layout: a wave holds the 64 samples of ONE pixel (spp 64), of four (16), or a count
    # that does not divide 64 -- nearly parallel rays, the wave-uniform packet walk and the in-wave sample fold --
    # on random scenes that may hold a 1,700-triangle mesh under a random transform
    rng = np.random.default_rng(5000 + _FUZZ_BASE + seed)
    jd = _rand_scene(rng, more_meshes=("data/dragon_superlow.obj", "data/dragon_superlow.obj"))
    spp = int(rng.choice([16, 64, 7, 96]))
    p = tmp_path / "f.json"
    p.write_text(json.dumps(jd))
    want, ost = oracle_image(str(p), spp=spp)
    for flags in (PRODUCTION, COUNTING):
        got, st = rayhs_amd.ray_trace_file(str(p), spp=spp, flags=flags)
        same = (got == want) | (np.isnan(got) & np.isnan(want))
        assert same.all(), (seed, flags, np.abs(np.nan_to_num(got) - np.nan_to_num(want)).max(), int((~same).any(axis=2).sum()))
        if flags == COUNTING:
            assert st["rays"] == ost["rays"], seed


@pytest.mark.parametrize("n_tris", [1, 2, 3, 4])
def test_meshes_of_one_leaf(tmp_path, n_tris):
    # a mesh of at most three triangles is ONE leaf of the fast BVH: its root node's second slot is the empty leaf (no
    # triangles, the first slot's box -- bvh_build.cpp); four triangles make the smallest tree with an inner node
    verts = [(-0.8, -0.6, 0.4), (0.7, -0.7, 0.9), (0.1, 0.8, 0.6), (0.9, 0.6, 0.2), (-0.6, 0.5, 1.1), (0.0, -0.1, -0.2)]
    faces = [(1, 2, 3), (2, 4, 3), (1, 3, 5), (6, 2, 1)][:n_tris]
    obj = tmp_path / "few.obj"
    obj.write_text("\n".join(["v %r %r %r" % v for v in verts] + ["vn 0 0 -1"] + ["f %d//1 %d//1 %d//1" % f for f in faces]) + "\n")
    mesh = {"geometry": {"type": "mesh", "fileName": str(obj), "transform": {"type": "translate", "vector": {"x": 0, "y": 0, "z": 0}}},
            "material": {"type": "plastic", "ior": 1.4, "cd": {"type": "flat", "color": {"r": 0.3, "g": 0.8, "b": 0.5}}}}
    text = _scene_json([mesh, FLOOR, SPH], [PT, DIR], w=40, h=32, depth=2)
    got, want, st, ost = _both(text, tmp_path)
    assert np.array_equal(got, want) and st["rays"] == ost["rays"]
    prod, _ = rayhs_amd.ray_trace_file(str(tmp_path / "s.json"))
    assert np.array_equal(prod, want)
    assert st["tris_tested"] > 0


def test_far_away_scene_walks_the_reference_tree(tmp_path):
    # coordinates beyond 1e30: the fp32 boxes of the fast BVH cannot hold them, every ray takes the reference tree
    # (trace_kernel.hip, begin(): `magnitude`); the answer must still be the oracle's, bit for bit
    big = 3.0e31
    def at(x, y, z): return {"x": big + x, "y": y, "z": z}
    mesh = {"geometry": {"type": "mesh", "fileName": "data/cube.obj", "transform": {"type": "translate", "vector": at(0.1, -0.2, 1.0)}},
            "material": {"type": "diffuse", "cd": {"type": "flat", "color": {"r": 0.9, "g": 0.4, "b": 0.2}}}}
    wall = {"geometry": {"type": "plane", "point": at(0, 0, 3.0), "normal": {"x": 0, "y": 0, "z": -1}, "tangent": {"x": 1, "y": 0, "z": 0}},
            "material": {"type": "diffuse", "cd": {"type": "flat", "color": {"r": 0.5, "g": 0.5, "b": 0.9}}}}
    scene = {"width": 24, "height": 16, "maxDepth": 1,
             "camera": {"position": at(0, 0, -2.0), "target": at(0, 0, 0), "up": {"x": 0, "y": 1, "z": 0},
                        "projection": {"type": "perspective", "fovy": 0.9, "width": 2, "height": 2, "near": 1}},
             "scene": {"objects": [mesh, wall], "lights": [dict(DIR), {"type": "point", "position": at(0.3, 0.8, -0.5), "color": {"r": 60, "g": 60, "b": 60}, "radius": 0.2}]}}
    p = tmp_path / "s.json"
    p.write_text(json.dumps(scene))
    want, ost = oracle_image(str(p))
    for flags in (PRODUCTION, COUNTING):
        got, st = rayhs_amd.ray_trace_file(str(p), flags=flags)
        assert _same(got, want), (flags, int((~((got == want) | (np.isnan(got) & np.isnan(want)))).any(axis=2).sum()))
    assert st["rays"] == ost["rays"]


def test_rays_through_mesh_vertices_and_edges(tmp_path):
    # Adversarial for the culling structure and the tie rule: an orthographic camera whose pixel rays pass
    # through the vertices of a triangulated height field to within an ulp (vertices were placed ON the rays),
    # so every primary ray hits up to six triangles at (almost) the same t, right on the faces of their leaf
    # boxes.  The fast BVH, the reference tree and the oracle must still agree bit for bit.
    n = 48
    pos, target = (0.3, -0.2, -5.0), (1.7, 0.4, -3.0)
    fwd = np.array(target) - np.array(pos)
    fwd = fwd * (1 / np.sqrt(np.dot(fwd, fwd)))
    lines = []
    for j in range(n + 1):          # one vertex per pixel-corner ray of an (n x n) image, plus the last row/column
        for i in range(n + 1):
            o = np.array([float(i - n // 2) + pos[0], float(n // 2 - j) + pos[1], pos[2]])
            s = 6.0 + 0.9 * np.sin(0.37 * i) + 0.7 * np.cos(0.23 * j) + 0.011 * ((i * 7 + j * 13) % 5)
            v = o + s * fwd
            lines.append("v %s %s %s" % tuple(repr(float(c)) for c in v))
    lines.append("vn 0.1 0.2 -1")
    for j in range(n):
        for i in range(n):
            a, b, c, d = j * (n + 1) + i + 1, j * (n + 1) + i + 2, (j + 1) * (n + 1) + i + 1, (j + 1) * (n + 1) + i + 2
            lines.append(f"f {a}//1 {b}//1 {d}//1")
            lines.append(f"f {a}//1 {d}//1 {c}//1")
    obj = tmp_path / "field.obj"
    obj.write_text("\n".join(lines) + "\n")
    mesh = {"geometry": {"type": "mesh", "fileName": str(obj), "transform": {"type": "translate", "vector": {"x": 0, "y": 0, "z": 0}}},
            "material": {"type": "plastic", "ior": 1.5, "cd": {"type": "flat", "color": {"r": 1, "g": 0.8, "b": 0.6}}}}
    scene = {"width": n, "height": n, "maxDepth": 2,
             "camera": {"position": dict(zip("xyz", pos)), "target": dict(zip("xyz", target)), "up": {"x": 0, "y": 1, "z": 0},
                        "projection": {"type": "orthographic", "fovy": 1.0, "width": 2, "height": 2, "near": 0.1}},
             "scene": {"objects": [mesh], "lights": [{"type": "directional", "direction": {"x": -0.2, "y": 0.5, "z": -1.0}, "color": {"r": 1, "g": 1, "b": 1}},
                                                     {"type": "point", "position": {"x": 0.5, "y": 0.5, "z": -4.0}, "color": {"r": 40, "g": 40, "b": 40}, "radius": 0.2}]}}
    p = tmp_path / "s.json"
    p.write_text(json.dumps(scene))
    want, ost = oracle_image(str(p))
    got, st = rayhs_amd.ray_trace_file(str(p), flags=COUNTING)
    exh, _ = rayhs_amd.ray_trace_file(str(p), flags=L.FLAG_COUNT | L.FLAG_EXHAUSTIVE)
    assert np.array_equal(exh, want)
    assert np.array_equal(got, want), (np.abs(got - want).max(), int((got != want).any(axis=2).sum()))
    r = build_rendering(str(p))                      # the production kernel: scan order, then (were the frame large enough) re-ordered
    for _ in range(2):
        prod, _ = r.ray_trace()
        assert np.array_equal(prod, want), (np.abs(prod - want).max(), int((prod != want).any(axis=2).sum()))
    r.close()
    assert st["rays"] == ost["rays"] and (want > 0).any(axis=2).mean() > 0.9     # the field really is hit everywhere


@pytest.mark.parametrize("dist", [3.0, 300.0, 30000.0])
def test_camera_far_from_a_small_mesh(tmp_path, dist):
    # the fp32 culling structure works in mesh-centred coordinates with an explicit margin for the rounding of a
    # far-away origin; results must not depend on it
    torus = {"geometry": {"type": "mesh", "fileName": "data/torus.obj", "transform": {"type": "sequence", "transforms": [
        {"type": "scale", "vector": {"x": 0.5, "y": 0.5, "z": 0.5}}, {"type": "rotateX", "angle": 0.9},
        {"type": "translate", "vector": {"x": 0.2, "y": -0.1, "z": 0.3}}]}},
        "material": {"type": "plastic", "ior": 1.6, "cd": {"type": "flat", "color": {"r": 1, "g": 0.6, "b": 0.3}}}}
    jd = json.loads(_scene_json([torus, FLOOR], [DIR, PT], w=40, h=30, depth=2))
    jd["camera"]["position"] = {"x": 0.3 * dist, "y": 0.2 * dist, "z": -dist}
    jd["camera"]["projection"]["fovy"] = 0.9272952180016123 * dist / 3.0    # keep the object in frame (f ~ h / fovy)
    got, want, st, ost = _both(json.dumps(jd), tmp_path)
    assert np.array_equal(got, want) and st["rays"] == ost["rays"]
    assert (want > 0).any(axis=2).sum() > 50


def test_short_divisions_and_roots_equal_the_compilers_ieee_sequences_bit_for_bit():
    # rayhs_amd/csrc/short_div.h: 1/x, n/d and sqrt x without the range fix-ups of the compiler's correctly rounded sequences, used
    # by the trace kernels where the operands' range makes those fix-ups identities.  2^22 pseudo-random operands per case and seed,
    # exponents spread over the whole stated range, every 64th a power of two or all-ones mantissa: not one may differ.
    import ctypes as C
    lib = L.load()
    for seed in (1, 0x9e3779b97f4a7c15):
        out = (C.c_uint64 * 5)()
        assert lib.rayhs_selftest_arith(0, 1 << 22, seed, out) == 0, lib.rayhs_last_error()
        assert list(out) == [0, 0, 0, 0, 0], list(out)


def test_mesh_with_vertex_coordinates_beyond_1e90_is_rendered_by_the_exhaustive_kernel(tmp_path):
    # The fast walks take 1 / det without the division's range fix-ups, which is the same double only while |det| < 2^1022 -- certain for
    # vertex coordinates below 1e90 and rays below 1e30, not beyond.  The launcher therefore renders a scene with larger (finite) vertex
    # coordinates with the exhaustive kernel whatever was asked for (abi.cpp, prepare_launch): same pixels as the oracle, bit for bit,
    # with the production flags as well as the counting ones.
    cube = {"geometry": {"type": "mesh", "fileName": "data/cube.obj", "transform": {"type": "scale", "vector": {"x": 1e95, "y": 1e95, "z": 1e95}}},
            "material": {"type": "diffuse", "cd": {"type": "flat", "color": {"r": 0.9, "g": 0.4, "b": 0.2}}}}
    jd = json.loads(_scene_json([cube], [DIR, PT], w=24, h=16, depth=1))
    jd["camera"]["position"] = {"x": 0.1, "y": 0.2, "z": -0.3}
    p = tmp_path / "s.json"
    p.write_text(json.dumps(jd))
    want, ost = oracle_image(str(p))
    assert (want > 0).any(axis=2).sum() > 50          # the camera sits inside the cube: every ray meets a face ~1e95 away
    for flags in (PRODUCTION, COUNTING):
        got, st = rayhs_amd.ray_trace_file(str(p), flags=flags)
        assert _same(got, want), flags
    assert st["rays"] == ost["rays"]
