"""Diagnostic: is the work order a frame was given really slowest-first by the chunk times THAT frame measured?
    RAYHS_LIB=build_exp/cur/librayhs_hip.so python tools/order_check.py [workload]
(production kernels of a -DRAYHS_DEV_KNOBS build: the chunk times come from its RAYHS_DUMP_TILE_COSTS, the order from rayhs_scene_tile_order)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayhs_amd import build_rendering, lib as L
W = {"dragon": ("scenes/dragon_1080p.json", 16), "dragon_low": ("scenes/dragon_low_1080p.json", 4), "cornell": ("scenes/cornell_1080p_d0.json", 1)}
scene, spp = W[sys.argv[1] if len(sys.argv) > 1 else "dragon"]
r = build_rendering(scene)
pc = "/tmp/tile_costs.bin"
if os.path.exists(pc): os.remove(pc)
r.capture_tile_order(True)
for i in range(5):
    if i == 4:
        os.environ["RAYHS_DUMP_TILE_COSTS"] = pc
        L.load().rayhs_config_reload()
    _, st = r.ray_trace(spp=spp, out_format=L.OUT_F32)
order_given, _ = r.tile_order()
raw = np.fromfile(pc, dtype=np.float32)
hdr = raw[:4].view(np.int32); n = int(hdr[0])
cost = raw[4:4 + n]            # 100 MHz ticks: longest chunk of the tile in THIS frame
tile = order_given.astype(np.int64)
# (the issue-priority class of a tile -- quarters of the slowest tile's time -- is not part of what rayhs_scene_tile_order returns: recomputed)
top = cost[tile].max() if n else 1.0
prio = np.where(cost[tile] > 0.75 * top, 3, np.where(cost[tile] > 0.5 * top, 2, np.where(cost[tile] > 0.25 * top, 1, 0)))
print(f"{n} tiles, kernel {st['kernel_ms']:.3f} ms; cost (us) min {cost.min()/100:.0f} median {np.median(cost)/100:.0f} p90 {np.percentile(cost,90)/100:.0f} max {cost.max()/100:.0f}")
per_band = -(-n // 8)
# the trace kernel's bands: n_tiles*64*spp items cut into 8 ranges of whole chunks
total = n * 64 * spp; per_range = -(-total // 512) * 64; tiles_per_band = per_range / (64 * spp)
print(f"tiles per band in the kernel: {tiles_per_band:.2f}")
c = cost[tile]
for b in range(8):
    lo, hi = int(round(b * tiles_per_band)), min(n, int(round((b + 1) * tiles_per_band)))
    seg = c[lo:hi]
    q = [np.median(seg[int(k * len(seg) / 8):int((k + 1) * len(seg) / 8)]) / 100 for k in range(8)]
    last = seg[-40:] / 100
    print(f"band {b}: median cost by eighth of the band (us): " + " ".join(f"{x:5.0f}" for x in q) + f" | last 40 tiles: median {np.median(last):.0f} max {last.max():.0f}  prio of last: {np.bincount(prio[lo:hi][-40:], minlength=4)}")
from scipy.stats import spearmanr
print("rank correlation (position in the order vs this frame's cost):", spearmanr(np.arange(n) % int(tiles_per_band), c).correlation)
