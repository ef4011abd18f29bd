"""Kernel time of every rank's share of a frame when it is cut into interleaved 8-row stripes (one GPU, ranks in turn).

    python tools/stripe_balance.py [world] [scene] [spp]
Shows the load balance of the sharding rule and how well a 1/world frame still fills the chip.
"""
import sys
sys.path.insert(0, ".")
from rayhs_amd import build_rendering, lib as L
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scene = sys.argv[2] if len(sys.argv) > 2 else "scenes/dragon_1080p.json"
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
r = build_rendering(scene)
full = sorted(r.ray_trace(spp=spp, out_format=L.OUT_F32)[1]["kernel_ms"] for _ in range(6))[2]
parts = []
for rank in range(world):
    kw = dict(spp=spp, out_format=L.OUT_F32, stripe_rows=8, stripe_rank=rank, stripe_count=world)
    parts.append(sorted(r.ray_trace(**kw)[1]["kernel_ms"] for _ in range(6))[2])
print(f"whole frame {full:.3f} ms; per rank of {world}: " + " ".join(f"{p:.3f}" for p in parts))
print(f"slowest rank {max(parts):.3f} ms -> kernel-level speed-up {full / max(parts):.2f}x of {world}")
