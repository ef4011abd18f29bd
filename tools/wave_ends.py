"""Diagnostic (-DRAYHS_PROFILE_TIME build via RAYHS_LIB, or a -DRAYHS_WAVE_ENDS build with RAYHS_WAVE_ENDS_PRODUCTION=1: the production kernel): when the waves of a steady-state frame end, and how long their last chunk took.
    RAYHS_LIB=build_exp/binpt/librayhs_hip.so python tools/wave_ends.py [workload]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rayhs_amd import build_rendering, lib as L
W = {"dragon": ("scenes/dragon_1080p.json", 16), "dragon_low": ("scenes/dragon_low_1080p.json", 4), "cornell": ("scenes/cornell_1080p_d0.json", 1)}
scene, spp = W[sys.argv[1] if len(sys.argv) > 1 else "dragon"]
r = build_rendering(scene)
path = "/tmp/wave_ends.bin"
for i in range(4):
    if i == 3:
        os.environ["RAYHS_DUMP_WAVE_ENDS"] = path
        L.load().rayhs_config_reload()   # (the library reads its environment once)
    _, st = r.ray_trace(spp=spp, flags=0 if os.environ.get("RAYHS_WAVE_ENDS_PRODUCTION") else L.FLAG_COUNT, out_format=L.OUT_F32)
w = np.fromfile(path, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
w = w[(w[:, 3] > 0) & (w[:, 3] < 10**7) & (w[:, 1] > w[:, 0]) & (w[:, 1] - w[:, 0] < 10**9)]  # (the dump is sized for the largest grid: keep the rows waves wrote)
t0 = w[:, 0].min()
start, end, last, chunks = (w[:, 0] - t0) / 100.0, (w[:, 1] - t0) / 100.0, (w[:, 2] - t0) / 100.0, w[:, 3]
print(f"waves {len(w)}  kernel_ms {st['kernel_ms']:.3f}  span {end.max():.0f} us  starts: max {start.max():.1f} us")
print(f"end times (us): min {end.min():.0f}  p10 {np.percentile(end, 10):.0f}  median {np.median(end):.0f}  p90 {np.percentile(end, 90):.0f}  max {end.max():.0f}   mean idle {end.max() - end.mean():.0f} us")
d = end - last
print(f"last chunk of a wave (us): median {np.median(d):.0f}  p90 {np.percentile(d, 90):.0f}  max {d.max():.0f};  drawn at: median {np.median(last):.0f}  min {last.min():.0f}")
late = np.argsort(end)[-10:]
print("the ten last waves: " + "  ".join(f"[end {end[i]:.0f} last chunk {d[i]:.0f} us, {chunks[i]} chunks]" for i in late))
print(f"chunks per wave: min {chunks.min()} median {np.median(chunks):.0f} max {chunks.max()}")
h, edges = np.histogram(end, bins=12)
print("histogram of end times: " + "  ".join(f"{edges[i]:.0f}:{h[i]}" for i in range(12)))
