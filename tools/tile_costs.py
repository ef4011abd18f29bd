"""Measured chunk time per 8x8 tile against the static first-frame estimate (GPU box; development aid).

    RAYHS_LIB=build_exp/cur/librayhs_hip.so python tools/tile_costs.py [raw] > gpurun_out/static_costs.md      (a -DRAYHS_DEV_KNOBS build)
Renders three frames per workload with RAYHS_DUMP_TILE_COSTS (the third frame's measured chunk times and the
estimate of rayhs_static_costs are appended to a file by the library) and tabulates one against the other.
`raw`: tabulate the estimate's raw feature instead (boxes of the upper BVH levels met by the tile's five rays)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayhs_amd import build_rendering, lib as L

raw = len(sys.argv) > 1 and sys.argv[1] == "raw"
if raw:
    os.environ["RAYHS_DUMP_RAW_FEATURE"] = "1"
print("# Static tile estimate vs measured chunk times (100 MHz ticks, longest chunk of the tile)\n")
for name, scene, spp in (("dragon", "scenes/dragon_1080p.json", 16), ("dragon_low", "scenes/dragon_low_1080p.json", 4),
                         ("cornell", "scenes/cornell_1080p_d0.json", 1)):
    path = f"/tmp/costs_{name}.bin"
    if os.path.exists(path):
        os.unlink(path)
    r = build_rendering(scene)
    for i in range(3):
        if i == 2:
            os.environ["RAYHS_DUMP_TILE_COSTS"] = path
            L.load().rayhs_config_reload()   # (the library reads its environment once)
        r.ray_trace(spp=spp, out_format=L.OUT_F32)
    os.environ.pop("RAYHS_DUMP_TILE_COSTS")
    L.load().rayhs_config_reload()
    r.close()
    buf = open(path, "rb").read()
    n = int(np.frombuffer(buf[:4], dtype=np.int32)[0])
    m = np.frombuffer(buf[16:16 + 4 * n], dtype=np.float32)
    st = np.frombuffer(buf[16 + 4 * n:16 + 8 * n], dtype=np.float32)
    rk = lambda v: np.argsort(np.argsort(v)).astype(np.float64)
    rho = np.corrcoef(rk(m), rk(st))[0, 1]
    print(f"## {name} ({n} tiles, {spp} spp): rank correlation {rho:.3f}\n")
    print("| estimate | tiles | measured mean | p50 | p95 |\n|---|---|---|---|---|")
    edges = [0, 1, 5, 10, 20, 40, 60, 80, 100, 150, 200, 300, 1000] if raw else [0, 1.2, 1.6, 2.5, 4, 6, 8, 11, 14, 17, 25]
    for a, b in zip(edges[:-1], edges[1:]):
        sel = (st >= a) & (st < b)
        if sel.sum():
            print(f"| [{a}, {b}) | {int(sel.sum())} | {m[sel].mean():.0f} | {np.median(m[sel]):.0f} | {np.percentile(m[sel], 95):.0f} |")
    print()
