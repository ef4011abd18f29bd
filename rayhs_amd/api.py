"""Host-side mirror of the reference's driver for the trace path.

Names follow src/RayHs.hs: a `Rendering` (scene, camera, width, height,
maxDepth; src/RayHs.hs:41-45) is built from a RenderDesc JSON file
(`build_rendering`, src/RayHs.hs:47-50) and `ray_trace` (src/RayHs.hs:161-166)
returns the image.  All computation happens in librayhs_hip.so on the GPU.
"""
import ctypes as C
import subprocess

import numpy as np

from . import lib as L

_NP = {L.OUT_F32: np.float32, L.OUT_F64: np.float64, L.OUT_I16: np.int16}


class RenderDesc:
    """Decoded + flattened RenderDesc (src/Descriptors.hs:19-23), host memory only."""

    def __init__(self, path=None, text=None):
        lib = L.load()
        h = C.c_void_p()
        if path is not None:
            L.check(lib.rayhs_render_desc_load(str(path).encode(), C.byref(h)))
        else:
            L.check(lib.rayhs_render_desc_parse(text.encode(), C.byref(h)))
        self._h = h
        self.width = lib.rayhs_render_desc_width(h)
        self.height = lib.rayhs_render_desc_height(h)
        self.max_depth = lib.rayhs_render_desc_max_depth(h)
        self.tree_depth = lib.rayhs_render_desc_tree_depth(h)

    @property
    def scene_desc(self):
        return L.load().rayhs_render_desc_scene(self._h).contents

    @property
    def camera(self):
        return L.load().rayhs_render_desc_camera(self._h).contents

    def arrays(self):
        """The SoA arrays as numpy copies (for host-logic tests)."""
        d = self.scene_desc

        def arr(ptr, n, dt):
            if n == 0:
                return np.zeros(0, dtype=dt)
            return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dt, copy=True)

        no, npl, ns, nm, nn, nt, nl = (d.n_objects, d.n_planes, d.n_spheres, d.n_meshes, d.n_nodes, d.n_tris, d.n_lights)
        return {
            "obj_geom_kind": arr(d.obj_geom_kind, no, np.int32), "obj_geom_index": arr(d.obj_geom_index, no, np.int32),
            "mat_kind": arr(d.mat_kind, no, np.int32), "mat_ior": arr(d.mat_ior, no, np.float64),
            "mat_ce": arr(d.mat_ce, 3 * no, np.float64).reshape(-1, 3), "cmap_kind": arr(d.cmap_kind, no, np.int32),
            "cmap_c1": arr(d.cmap_c1, 3 * no, np.float64).reshape(-1, 3), "cmap_c2": arr(d.cmap_c2, 3 * no, np.float64).reshape(-1, 3),
            "cmap_size": arr(d.cmap_size, no, np.float64),
            "tex_w": arr(d.tex_w, no, np.int32), "tex_h": arr(d.tex_h, no, np.int32), "tex_offset": arr(d.tex_offset, no, np.int64),
            "texels": arr(d.texels, 3 * d.n_texels, np.float64).reshape(-1, 3),
            "plane_point": arr(d.plane_point, 3 * npl, np.float64).reshape(-1, 3),
            "plane_normal": arr(d.plane_normal, 3 * npl, np.float64).reshape(-1, 3),
            "plane_tangent": arr(d.plane_tangent, 3 * npl, np.float64).reshape(-1, 3),
            "sphere_center": arr(d.sphere_center, 3 * ns, np.float64).reshape(-1, 3), "sphere_radius": arr(d.sphere_radius, ns, np.float64),
            "mesh_root": arr(d.mesh_root, nm, np.int32),
            "node_lo": arr(d.node_lo, 3 * nn, np.float64).reshape(-1, 3), "node_hi": arr(d.node_hi, 3 * nn, np.float64).reshape(-1, 3),
            "node_left": arr(d.node_left, nn, np.int32), "node_right": arr(d.node_right, nn, np.int32),
            "node_first_tri": arr(d.node_first_tri, nn, np.int32), "node_tri_count": arr(d.node_tri_count, nn, np.int32),
            "node_axis": arr(d.node_axis, nn, np.int32),
            "tri_pos": arr(d.tri_pos, 9 * nt, np.float64).reshape(-1, 3, 3), "tri_nrm": arr(d.tri_nrm, 9 * nt, np.float64).reshape(-1, 3, 3),
            "tri_uv": arr(d.tri_uv, 6 * nt, np.float64).reshape(-1, 3, 2),
            "light_kind": arr(d.light_kind, nl, np.int32), "light_vec": arr(d.light_vec, 3 * nl, np.float64).reshape(-1, 3),
            "light_color": arr(d.light_color, 3 * nl, np.float64).reshape(-1, 3), "light_radius": arr(d.light_radius, nl, np.float64),
        }

    def close(self):
        if getattr(self, "_h", None):
            L.load().rayhs_render_desc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the module globals may already be gone
            pass


class Rendering:
    """src/RayHs.hs:41-45 with the Scene resident in one GPU's HBM."""

    def __init__(self, desc, device=0, width=None, height=None, max_depth=None):
        lib = L.load()
        self.desc = desc
        self.width = desc.width if width is None else width
        self.height = desc.height if height is None else height
        self.max_depth = desc.max_depth if max_depth is None else max_depth
        self.device = device
        h = C.c_void_p()
        L.check(lib.rayhs_scene_create(C.byref(desc.scene_desc), device, C.byref(h)))
        self._h = h

    def params(self, spp=1, seed=24, row0=0, row1=None, stripe_rows=None, stripe_rank=0, stripe_count=1,
               out_format=L.OUT_F64, flags=0):
        p = L.RenderParams()
        p.width, p.height, p.max_depth = self.width, self.height, self.max_depth
        p.spp, p.seed = spp, seed
        p.row0 = row0
        p.row1 = self.height if row1 is None else row1
        p.stripe_rows = max(1, (p.row1 - p.row0) if stripe_rows is None else stripe_rows)
        p.stripe_rank, p.stripe_count = stripe_rank, stripe_count
        p.out_format, p.flags = out_format, flags
        return p

    def rows(self, p):
        n = L.load().rayhs_render_rows(C.byref(p))
        if n < 0:
            L.check(n)
        return n

    def ray_trace(self, **kw):
        """rayTrace (src/RayHs.hs:161-166): returns (image[rows, width, 3], stats dict)."""
        p = self.params(**kw)
        n = self.rows(p)
        out = np.empty((n, self.width, 3), dtype=_NP[p.out_format])
        st = L.Stats()
        L.check(L.load().rayhs_render(self._h, C.byref(self.desc.camera), C.byref(p), out.ctypes.data, C.byref(st)))
        return out, st.as_dict()

    def ray_trace_device(self, d_out_ptr, stream_ptr=0, d_counters_ptr=0, events=None, **kw):
        """Enqueue the trace into a caller-owned device buffer (no sync, no copy).  `events` = (before, after)
        hipEvent_t handles recorded around the trace kernel alone."""
        p = self.params(**kw)
        if events is None:
            L.check(L.load().rayhs_render_device(self._h, C.byref(self.desc.camera), C.byref(p), C.c_void_p(d_out_ptr),
                                                 C.c_void_p(stream_ptr), C.c_void_p(d_counters_ptr)))
        else:
            L.check(L.load().rayhs_render_device_events(self._h, C.byref(self.desc.camera), C.byref(p), C.c_void_p(d_out_ptr),
                                                        C.c_void_p(stream_ptr), C.c_void_p(d_counters_ptr),
                                                        C.c_void_p(events[0]), C.c_void_p(events[1])))
        return p

    def capture_tile_order(self, enable=True):
        """Diagnostics: have every launch of this handle keep the work order of the 8x8 tiles it was given."""
        L.check(L.load().rayhs_scene_capture_tile_order(self._h, 1 if enable else 0))

    def tile_order(self):
        """(tile indices in the order the last launch handed them out, came-from-the-static-estimate); empty in scan order."""
        lib = L.load()
        static = C.c_int(0)
        n = lib.rayhs_scene_tile_order(self._h, None, 0, C.byref(static))
        if n < 0:
            L.check(n)
        order = np.empty(n, dtype=np.int32)
        if n:
            got = lib.rayhs_scene_tile_order(self._h, order.ctypes.data_as(C.POINTER(C.c_int32)), n, C.byref(static))
            if got < 0:
                L.check(got)
        return order, bool(static.value)

    def close(self):
        if getattr(self, "_h", None):
            L.load().rayhs_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the module globals may already be gone
            pass


class MultiRendering:
    """One frame over several GPUs of the node from one process (rayhs_multi, include/rayhs.h): the replacement for
    the reference's parallel pixel map (src/Image.hs:34-36).  `devices` may repeat a device (rehearsal)."""

    def __init__(self, desc, devices, width=None, height=None, max_depth=None):
        lib = L.load()
        self.desc = desc
        self.width = desc.width if width is None else width
        self.height = desc.height if height is None else height
        self.max_depth = desc.max_depth if max_depth is None else max_depth
        self.devices = list(devices)
        h = C.c_void_p()
        arr = (C.c_int * len(self.devices))(*self.devices)
        L.check(lib.rayhs_multi_create(C.byref(desc.scene_desc), arr, len(self.devices), C.byref(h)))
        self._h = h

    def ray_trace(self, spp=1, seed=24, row0=0, row1=None, stripe_rows=8, out_format=L.OUT_F64, flags=0):
        p = L.RenderParams()
        p.width, p.height, p.max_depth = self.width, self.height, self.max_depth
        p.spp, p.seed, p.row0 = spp, seed, row0
        p.row1 = self.height if row1 is None else row1
        p.stripe_rows, p.stripe_rank, p.stripe_count = stripe_rows, 0, 1
        p.out_format, p.flags = out_format, flags
        out = np.empty((p.row1 - p.row0, self.width, 3), dtype=_NP[out_format])
        st = L.Stats()
        L.check(L.load().rayhs_multi_render(self._h, C.byref(self.desc.camera), C.byref(p), out.ctypes.data, C.byref(st)))
        return out, st.as_dict()

    def ray_trace_into(self, d_out, spp=1, seed=24, row0=0, row1=None, stripe_rows=8, out_format=L.OUT_F64, flags=0):
        """Same frame into a caller-owned DEVICE buffer on devices[0] (rayhs_multi_render_device); returns the stats."""
        p = L.RenderParams()
        p.width, p.height, p.max_depth = self.width, self.height, self.max_depth
        p.spp, p.seed, p.row0 = spp, seed, row0
        p.row1 = self.height if row1 is None else row1
        p.stripe_rows, p.stripe_rank, p.stripe_count = stripe_rows, 0, 1
        p.out_format, p.flags = out_format, flags
        st = L.Stats()
        L.check(L.load().rayhs_multi_render_device(self._h, C.byref(self.desc.camera), C.byref(p), C.c_void_p(d_out), C.byref(st)))
        return st.as_dict()

    def shares(self):
        n = len(self.devices)
        sh, rows = (C.c_double * n)(), (C.c_int * n)()
        L.check(L.load().rayhs_multi_shares(self._h, sh, rows, None, n))
        return list(sh), list(rows)

    def kernel_ms(self):
        """Trace-kernel time of every device in the last frame (rendered through ray_trace, which asks for stats)."""
        n = len(self.devices)
        ms = (C.c_double * n)()
        L.check(L.load().rayhs_multi_shares(self._h, None, None, ms, n))
        return list(ms)

    def close(self):
        if getattr(self, "_h", None):
            L.load().rayhs_multi_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: the module globals may already be gone
            pass


def build_rendering(path, device=0, **overrides):
    """buildRendering (src/RayHs.hs:47-50)."""
    return Rendering(RenderDesc(path), device, **overrides)


def write_ppm(path, img):
    """writePPM (src/Image.hs:70-75) from fp64 colours or toIntC values."""
    lib = L.load()
    h, w, _ = img.shape
    if img.dtype == np.int16:
        a = np.ascontiguousarray(img)
        L.check(lib.rayhs_write_ppm_i16(str(path).encode(), w, h, a.ctypes.data))
    else:
        a = np.ascontiguousarray(img, dtype=np.float64)
        L.check(lib.rayhs_write_ppm_f64(str(path).encode(), w, h, a.ctypes.data))


def run_cli(args, cwd=None):
    """Run the drop-in `rayhs` executable; returns CompletedProcess."""
    return subprocess.run([L.CLI_PATH] + list(args), cwd=cwd, capture_output=True, text=True)


def ray_trace_file(path, device=0, width=None, height=None, max_depth=None, **kw):
    """Decode, build, upload and trace one scene file; returns (image, stats)."""
    r = build_rendering(path, device, width=width, height=height, max_depth=max_depth)
    try:
        return r.ray_trace(**kw)
    finally:
        r.close()
