"""Where the wall-clock of a one-shot `rayhs -oX scene.json` goes besides the trace kernel: HIP runtime start-up,
scene preparation (decode, reference tree, SAH BVH), upload, first launch (code-object load), P3 writer.
PYTHONPATH=. python tools/cli_phases.py [scene] (GPU box)"""
import ctypes as C
import sys
import time

t00 = time.perf_counter()
hip = C.CDLL("libamdhip64.so")
t_dl = time.perf_counter()
hip.hipInit(0)
t_init = time.perf_counter()
hip.hipSetDevice(0)
hip.hipFree(None)
t_ctx = time.perf_counter()
print(f"dlopen libamdhip64 {1e3 * (t_dl - t00):.1f} ms, hipInit {1e3 * (t_init - t_dl):.1f} ms, hipSetDevice+hipFree(0) {1e3 * (t_ctx - t_init):.1f} ms")

from rayhs_amd import RenderDesc, Rendering, lib as L  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/dragon_1080p.json"
t = time.perf_counter(); L.load(); t1 = time.perf_counter()
print(f"dlopen librayhs_hip {1e3 * (t1 - t):.1f} ms")
for rep in range(2):
    t = time.perf_counter(); rd = RenderDesc(scene); t1 = time.perf_counter()
    r = Rendering(rd, device=0); t2 = time.perf_counter()
    img, _ = r.ray_trace(spp=16, out_format=L.OUT_I16); t3 = time.perf_counter()
    img, _ = r.ray_trace(spp=16, out_format=L.OUT_I16); t4 = time.perf_counter()
    print(f"rep {rep}: decode+trees (RenderDesc) {1e3 * (t1 - t):.1f} ms, scene create (BVH + upload) {1e3 * (t2 - t1):.1f} ms, "
          f"first render {1e3 * (t3 - t2):.1f} ms, second render {1e3 * (t4 - t3):.1f} ms")
    r.close()
