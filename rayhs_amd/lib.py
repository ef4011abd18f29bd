"""ctypes binding of librayhs_hip.so -- the C ABI declared in include/rayhs.h.

The library is the product: HIP kernels for gfx950 behind plain C entry points.
This module only loads it and mirrors its structs; there is no Python or CPU
implementation of the trace loop here, and loading fails loudly if the shared
library has not been built (run `python -c "import __graft_entry__ as g; g.build()"`).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RAYHS_LIB: development override used by tools/kbench.py to time experimental builds
LIB_PATH = os.environ.get("RAYHS_LIB") or os.path.join(_HERE, "bin", "librayhs_hip.so")
CLI_PATH = os.path.join(_HERE, "bin", "rayhs")

OK, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_IO, ERR_PARSE, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6
GEOM_PLANE, GEOM_SPHERE, GEOM_MESH = 0, 1, 2
OUT_F32, OUT_F64, OUT_I16 = 0, 1, 2
FLAG_FMA, FLAG_EXHAUSTIVE, FLAG_COUNT, FLAG_POW5_SQMUL = 1, 2, 4, 8

_pi32 = C.POINTER(C.c_int32)
_pi64 = C.POINTER(C.c_int64)
_pf64 = C.POINTER(C.c_double)


class SceneDesc(C.Structure):
    """rayhs_scene_desc (include/rayhs.h)."""
    _fields_ = [
        ("n_objects", C.c_int32), ("obj_geom_kind", _pi32), ("obj_geom_index", _pi32),
        ("mat_kind", _pi32), ("mat_ior", _pf64), ("mat_ce", _pf64), ("cmap_kind", _pi32),
        ("cmap_c1", _pf64), ("cmap_c2", _pf64), ("cmap_size", _pf64),
        ("tex_w", _pi32), ("tex_h", _pi32), ("tex_offset", _pi64), ("n_texels", C.c_int64), ("texels", _pf64),
        ("n_planes", C.c_int32), ("plane_point", _pf64), ("plane_normal", _pf64), ("plane_tangent", _pf64),
        ("n_spheres", C.c_int32), ("sphere_center", _pf64), ("sphere_radius", _pf64),
        ("n_meshes", C.c_int32), ("mesh_root", _pi32),
        ("n_nodes", C.c_int32), ("node_lo", _pf64), ("node_hi", _pf64),
        ("node_left", _pi32), ("node_right", _pi32), ("node_first_tri", _pi32), ("node_tri_count", _pi32), ("node_axis", _pi32),
        ("n_tris", C.c_int32), ("tri_pos", _pf64), ("tri_nrm", _pf64), ("tri_uv", _pf64),
        ("n_lights", C.c_int32), ("light_kind", _pi32), ("light_vec", _pf64), ("light_color", _pf64), ("light_radius", _pf64),
    ]


class Camera(C.Structure):
    """rayhs_camera."""
    _fields_ = [("position", C.c_double * 3), ("target", C.c_double * 3), ("up", C.c_double * 3),
                ("proj_kind", C.c_int32), ("fovy", C.c_double), ("proj_width", C.c_double),
                ("proj_height", C.c_double), ("near_", C.c_double)]


class RenderParams(C.Structure):
    """rayhs_render_params."""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("max_depth", C.c_int32), ("spp", C.c_int32),
                ("seed", C.c_uint64), ("row0", C.c_int32), ("row1", C.c_int32),
                ("stripe_rows", C.c_int32), ("stripe_rank", C.c_int32), ("stripe_count", C.c_int32),
                ("out_format", C.c_int32), ("flags", C.c_uint32)]


class Stats(C.Structure):
    """rayhs_stats."""
    _fields_ = [(n, C.c_uint64) for n in
                ("rays_primary", "rays_reflect", "rays_refract", "rays_refract_exit", "rays_shadow",
                 "nodes_tested", "tris_tested", "prims_tested", "shaded_hits", "pixels")] + [("kernel_ms", C.c_double),
                                                                                            ("shadow_skipped", C.c_uint64)]

    def as_dict(self):
        d = {n: (float(getattr(self, n)) if n == "kernel_ms" else int(getattr(self, n))) for n, _ in self._fields_}
        d["rays"] = sum(v for k, v in d.items() if k.startswith("rays_"))   # the reference's scene queries
        d["rays_traced"] = d["rays"] - d["shadow_skipped"]                   # ... of which these were really traced
        return d


# every symbol include/rayhs.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("rayhs_last_error", C.c_char_p, []),
    ("rayhs_abi_version", C.c_int, []),
    ("rayhs_sizeof", C.c_uint32, [C.c_int]),
    ("rayhs_device_count", C.c_int, []),
    ("rayhs_config_reload", None, []),
    ("rayhs_device_prepare", C.c_int, [C.c_int]),
    ("rayhs_selftest_arith", C.c_int, [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
    ("rayhs_scene_create", C.c_int, [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]),
    ("rayhs_scene_destroy", None, [C.c_void_p]),
    ("rayhs_render_rows", C.c_int, [C.POINTER(RenderParams)]),
    ("rayhs_render", C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]),
    ("rayhs_render_device", C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rayhs_render_device_events", C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_void_p]),
    ("rayhs_scene_capture_tile_order", C.c_int, [C.c_void_p, C.c_int]),
    ("rayhs_scene_tile_order", C.c_int, [C.c_void_p, _pi32, C.c_int, C.POINTER(C.c_int)]),
    ("rayhs_multi_create", C.c_int, [C.POINTER(SceneDesc), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    ("rayhs_multi_destroy", None, [C.c_void_p]),
    ("rayhs_multi_device_count", C.c_int, [C.c_void_p]),
    ("rayhs_multi_render", C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]),
    ("rayhs_multi_render_device", C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]),
    ("rayhs_multi_shares", C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int]),
    ("rayhs_render_desc_load", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    ("rayhs_render_desc_parse", C.c_int, [C.c_char_p, C.POINTER(C.c_void_p)]),
    ("rayhs_render_desc_free", None, [C.c_void_p]),
    ("rayhs_render_desc_scene", C.POINTER(SceneDesc), [C.c_void_p]),
    ("rayhs_render_desc_camera", C.POINTER(Camera), [C.c_void_p]),
    ("rayhs_render_desc_width", C.c_int, [C.c_void_p]),
    ("rayhs_render_desc_height", C.c_int, [C.c_void_p]),
    ("rayhs_render_desc_max_depth", C.c_int, [C.c_void_p]),
    ("rayhs_render_desc_tree_depth", C.c_int, [C.c_void_p]),
    ("rayhs_write_ppm_i16", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
    ("rayhs_write_ppm_f64", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
    ("rayhs_sample01", C.c_double, [C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]),
]

ABI_VERSION = 5   # RAYHS_ABI_VERSION this binding was written against
_LIB = None


class RayhsError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"librayhs_hip error {code}: {message}")
        self.code = code
        self.message = message


def load():
    """dlopen librayhs_hip.so; raises if it has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                              "(hipcc --offload-arch=gfx950); rayhs_amd has no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        # the structs below are hand-written mirrors: refuse a library of another ABI version or another layout
        # instead of letting it write past them (version 2 grew rayhs_stats, versions 3 and 4 added entry points;
        # include/rayhs.h, "ABI history").  The version is asked FIRST: an older library lacks symbols bound below.
        try:
            lib.rayhs_abi_version.restype = C.c_int
            found = lib.rayhs_abi_version()
        except AttributeError:
            found = "none (no rayhs_abi_version)"
        if found != ABI_VERSION:
            raise ImportError(f"{LIB_PATH} implements ABI version {found}, this binding needs {ABI_VERSION}: rebuild it")
        for name, res, args in SYMBOLS:
            try:
                f = getattr(lib, name)
            except AttributeError:
                raise ImportError(f"{LIB_PATH} does not export {name} (declared in include/rayhs.h): rebuild it") from None
            f.restype = res
            f.argtypes = args
        for which, (name, cls) in enumerate((("rayhs_scene_desc", SceneDesc), ("rayhs_camera", Camera),
                                             ("rayhs_render_params", RenderParams), ("rayhs_stats", Stats))):
            if lib.rayhs_sizeof(which) != C.sizeof(cls):
                raise ImportError(f"{LIB_PATH}: sizeof({name}) is {lib.rayhs_sizeof(which)} in the library, {C.sizeof(cls)} in rayhs_amd/lib.py")
        _LIB = lib
    return _LIB


def check(rc):
    if rc != OK:
        raise RayhsError(rc, load().rayhs_last_error().decode(errors="replace"))
    return rc
