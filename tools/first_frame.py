"""First (cold) frame of a scene handle against its steady state, per BASELINE workload (GPU box).

    python tools/first_frame.py [workload ...]  > profiles/rNN/first_frame.jsonl
The first frame of a geometry has no measured chunk times: it is ordered by the static tile estimate
(rayhs_static_costs); RAYHS_NO_STATIC_ORDER=1 shows the scan-order first frame it replaces.  Every frame of a
handle must be the same image.
"""
import json, os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

W = {"cornell": ("scenes/cornell_1080p_d0.json", 1), "dragon_low": ("scenes/dragon_low_1080p.json", 4),
     "dragon": ("scenes/dragon_1080p.json", 16), "dragon4k": ("scenes/dragon_4k.json", 64)}


def one(name):
    from rayhs_amd import build_rendering, lib as L
    scene, spp = W[name]
    firsts, steady, warm = [], [], []
    for rep in range(3):                      # three fresh handles: three first frames
        r = build_rendering(scene)
        img0, st = r.ray_trace(spp=spp, out_format=L.OUT_F32)
        firsts.append(st["kernel_ms"])
        for i in range(5):
            img, st = r.ray_trace(spp=spp, out_format=L.OUT_F32)
            if i >= 2:
                steady.append(st["kernel_ms"])
        assert np.array_equal(img, img0)
        # the same static order once more on the WARM handle (another geometry in between resets the measured order): what is
        # left of first-over-steady here is the order alone, the rest of it is first touch of fresh buffers (scene, workspace)
        r.ray_trace(spp=spp, out_format=L.OUT_F32, row1=r.height - 8)
        warm.append(r.ray_trace(spp=spp, out_format=L.OUT_F32)[1]["kernel_ms"])
        r.close()
    return {"workload": name, "first_frame_ms": sorted(firsts)[1], "steady_ms": sorted(steady)[len(steady) // 2],
            "first_over_steady": sorted(firsts)[1] / sorted(steady)[len(steady) // 2],
            "first_order_on_warm_handle_ms": sorted(warm)[1], "warm_first_over_steady": sorted(warm)[1] / sorted(steady)[len(steady) // 2],
            "order": "scan" if os.environ.get("RAYHS_NO_STATIC_ORDER") else "static estimate"}


if __name__ == "__main__":
    if os.environ.get("_FF_CHILD"):
        print(json.dumps(one(sys.argv[1])), flush=True)
    else:
        for name in (sys.argv[1:] or list(W)):
            for env in ({}, {"RAYHS_NO_STATIC_ORDER": "1"}):
                subprocess.run([sys.executable, __file__, name], env=dict(os.environ, _FF_CHILD="1", **env), check=True)
