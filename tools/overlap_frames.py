"""Throughput of back-to-back frames when two scene handles on two streams alternate, against one handle on one stream:
the idle tail of frame i (a persistent kernel ends with its slowest wave) then overlaps the start of frame i + 1.

    python tools/overlap_frames.py [scene] [spp] [world] [frames in flight]   (world > 1: rank 0's share of a striped frame)
"""
import sys, time, torch
sys.path.insert(0, ".")
from rayhs_amd import build_rendering, lib as L
scene = sys.argv[1] if len(sys.argv) > 1 else "scenes/dragon_1080p.json"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 2
rs = [build_rendering(scene) for _ in range(depth)]
w, h = rs[0].width, rs[0].height
outs = [torch.empty((h, w, 3), dtype=torch.float32, device="cuda") for _ in range(depth)]
streams = [torch.cuda.Stream() for _ in range(depth)]
kw = dict(spp=spp, out_format=L.OUT_F32, stripe_rows=8, stripe_rank=0, stripe_count=world)
def run(n, two):
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(n):
        j = i % depth if two else 0
        rs[j].ray_trace_device(outs[j].data_ptr(), streams[j].cuda_stream, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
for two in (False, True):
    run(6, two)
    print((f"{depth} handles / {depth} streams" if two else "one handle / one stream  "), " ".join(f"{run(40, two):.3f}" for _ in range(3)), "ms per frame")
