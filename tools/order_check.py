"""Diagnostic: is the work order a frame was given really slowest-first by the chunk times THAT frame measured?
    python tools/order_check.py [workload]   (production kernels; uses RAYHS_DUMP_TILE_COSTS / RAYHS_DUMP_TILE_ORDER)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayhs_amd import build_rendering, lib as L
W = {"dragon": ("scenes/dragon_1080p.json", 16), "dragon_low": ("scenes/dragon_low_1080p.json", 4), "cornell": ("scenes/cornell_1080p_d0.json", 1)}
scene, spp = W[sys.argv[1] if len(sys.argv) > 1 else "dragon"]
r = build_rendering(scene)
pc, po = "/tmp/tile_costs.bin", "/tmp/tile_order.bin"
for f in (pc, po):
    if os.path.exists(f): os.remove(f)
for i in range(5):
    if i == 4: os.environ["RAYHS_DUMP_TILE_COSTS"] = pc; os.environ["RAYHS_DUMP_TILE_ORDER"] = po
    _, st = r.ray_trace(spp=spp, out_format=L.OUT_F32)
raw = np.fromfile(pc, dtype=np.float32)
hdr = raw[:4].view(np.int32); n = int(hdr[0])
cost = raw[4:4 + n]            # 100 MHz ticks: longest chunk of the tile in THIS frame
order = np.fromfile(po, dtype=np.int32)[4:4 + n]
tile = order & ((1 << 28) - 1); prio = (order >> 28) & 3
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
