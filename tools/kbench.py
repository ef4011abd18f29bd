"""Kernel timing harness for experimental builds (development aid).

    RAYHS_LIB=/path/to/librayhs_hip.so python tools/kbench.py [workload] [reps]
Prints HIP-event kernel time of the trace kernel, Mrays/s and a parity check
against tests/golden (bit-exact expected for the default exact kernels).
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayhs_amd import build_rendering, lib as L

W = {"dragon": ("scenes/dragon_1080p.json", 16), "dragon_low": ("scenes/dragon_low_1080p.json", 4),
     "cornell": ("scenes/cornell_1080p_d0.json", 1), "dragon1": ("scenes/dragon_1080p.json", 1),
     "dragon4k": ("scenes/dragon_4k.json", 64), "dragon64": ("scenes/dragon_1080p.json", 64)}
name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
flags = L.FLAG_FMA if os.environ.get("RAYHS_FMA") else 0
# parity first
ok = True
meta = json.load(open("tests/golden/golden.json"))
for g in ("cornell_96", "dragon_superlow_96_spp4", "transform_64"):
    m = meta[g]
    r = build_rendering(m["scene"])
    img, _ = r.ray_trace(spp=m["spp"], seed=m["seed"], flags=flags)
    same = np.array_equal(img, np.load(f"tests/golden/{g}.npy"))
    ok &= same
    if not same:
        print("PARITY FAIL", g, np.abs(img - np.load(f"tests/golden/{g}.npy")).max())
    r.close()
scene, spp = W[name]
r = build_rendering(scene)
_, st = r.ray_trace(spp=spp, flags=flags | L.FLAG_COUNT, out_format=L.OUT_F32)
rays = st["rays"]
ms = []
for i in range(reps):
    _, s2 = r.ray_trace(spp=spp, flags=flags, out_format=L.OUT_F32)
    ms.append(s2["kernel_ms"])
best = min(ms)
alg = 32 * st["nodes_tested"] + 36 * st["tris_tested"] + 40 * st["prims_tested"] + 60 * st["shaded_hits"] + 12 * st["pixels"]
med = sorted(ms)[len(ms) // 2]
print(f"{os.environ.get('RAYHS_LIB','default')} {name}: parity={'OK' if ok else 'FAIL'} kernel_ms: n={len(ms)} median={med:.3f} best={best:.3f} "
      f"Mrays/s={rays/best/1e3:.0f} alg_GB/s={alg/best/1e6:.0f} frac={alg/best/1e6/8000:.3f} nodes={st['nodes_tested']} tris={st['tris_tested']}")
