"""Development aid: kernel time of N consecutive steady-state frames (RAYHS_LIB selects the build).  tools/frame_times.py [workload] [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayhs_amd import build_rendering, lib as L
W = {"dragon": ("scenes/dragon_1080p.json", 16), "dragon_low": ("scenes/dragon_low_1080p.json", 4), "cornell": ("scenes/cornell_1080p_d0.json", 1)}
scene, spp = W[sys.argv[1] if len(sys.argv) > 1 else "dragon"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
r = build_rendering(scene)
ms = []
for i in range(n):
    _, st = r.ray_trace(spp=spp, out_format=L.OUT_F32)
    ms.append(st["kernel_ms"])
print(os.environ.get("RAYHS_LIB", "default").split("/")[-2], " ".join(f"{x:.3f}" for x in ms))
