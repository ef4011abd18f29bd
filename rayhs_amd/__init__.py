"""rayhs_amd -- MI355X (gfx950) replacement for the per-pixel trace loop of
oilandrust/rayhs, behind the C ABI of include/rayhs.h.

    csrc/   HIP kernels, the C ABI, the C++ host preparation and the `rayhs` CLI
    lib.py  ctypes binding of librayhs_hip.so (fails loudly if it is not built)
    api.py  RenderDesc / Rendering / ray_trace mirroring src/RayHs.hs
"""
from . import lib  # noqa: F401
from .api import MultiRendering, RenderDesc, Rendering, build_rendering, ray_trace_file, run_cli, write_ppm  # noqa: F401
