"""Kernel time of every GPU's share of a frame cut into 8-row stripes, on ONE GPU (shares timed one after the other).

    python tools/stripe_balance.py [workload ...]      (JSON lines; GPU box)
For 2, 4 and 8 shares: the round-robin rule of rayhs_render_params (what bench.py's ranks use) and the cost-balanced
stripe lists of rayhs_multi (RAYHS_MULTI_SERIAL=1 makes the library run its entries in turn, so that entries sharing
this GPU are timed as if each had a GPU of its own).  slowest share -> the kernel-level speed-up N GPUs could reach.
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RAYHS_MULTI_SERIAL"] = "1"
from rayhs_amd import MultiRendering, RenderDesc, Rendering, lib as L

W = {"cornell": ("scenes/cornell_1080p_d0.json", 1), "dragon_low": ("scenes/dragon_low_1080p.json", 4),
     "dragon": ("scenes/dragon_1080p.json", 16), "dragon4k": ("scenes/dragon_4k.json", 64)}
med = lambda v: sorted(v)[len(v) // 2]
for name in (sys.argv[1:] or ["dragon", "dragon4k"]):
    scene, spp = W[name]
    rd = RenderDesc(scene)
    r = Rendering(rd)
    reps = 3 if name == "dragon4k" else 5
    full = med([r.ray_trace(spp=spp, out_format=L.OUT_F32)[1]["kernel_ms"] for _ in range(reps + 1)][1:])
    for world in (2, 4, 8):
        rr = []
        for rank in range(world):
            kw = dict(spp=spp, out_format=L.OUT_F32, stripe_rows=8, stripe_rank=rank, stripe_count=world)
            rr.append(med([r.ray_trace(**kw)[1]["kernel_ms"] for _ in range(reps + 1)][1:]))
        m = MultiRendering(rd, [0] * world)
        per = []
        for _ in range(reps + 1):
            m.ray_trace(spp=spp, out_format=L.OUT_F32)
            per.append(m.kernel_ms())
        bal = [med([p[i] for p in per[1:]]) for i in range(world)]
        shares, rows = m.shares()
        m.close()
        print(json.dumps({"workload": name, "shares": world, "whole_frame_ms": round(full, 3),
                          "round_robin_ms": [round(v, 3) for v in rr], "round_robin_speedup": round(full / max(rr), 2),
                          "cost_balanced_ms": [round(v, 3) for v in bal], "cost_balanced_speedup": round(full / max(bal), 2),
                          "estimated_shares": [round(v, 4) for v in shares], "rows": rows}), flush=True)
    r.close()
