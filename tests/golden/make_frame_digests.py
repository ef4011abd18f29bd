"""Digests of the FULL oracle frames of the BASELINE configurations (configs[1]..[4]).

RESTATEMENT-DERIVED, like everything under tests/golden/ (the reference has no expected outputs and
no GHC exists here): the CPU oracle renders each whole frame once, in the build container, and this
script stores
  * sha256 of the fp64 image bytes and of its fp32 cast (the 12 B/pixel frame bench.py renders),
  * one exactly-rounded sum (math.fsum, order-independent) per image row, as a hex float, so that a
    mismatch can be located without the oracle,
  * the oracle's ray / work counters of the frame.
The GPU tests and bench.py compare whole production frames against these without running the oracle
at full size on the GPU box.  Run from the repo root (minutes; the 4K / 64 spp frame is ~2.5 G rays):

    python tests/golden/make_frame_digests.py [name ...]
"""
import hashlib
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "frame_digests.json")

# name: (scene, spp)  -- the workloads of bench.py
FRAMES = {
    "cornell": ("scenes/cornell_1080p_d0.json", 1),
    "dragon_low": ("scenes/dragon_low_1080p.json", 4),
    "dragon": ("scenes/dragon_1080p.json", 16),
    "dragon4k": ("scenes/dragon_4k.json", 64),
}


def digest(img):
    """img: float64 [h, w, 3]"""
    img = np.ascontiguousarray(img, dtype=np.float64)
    return {"sha256_f64": hashlib.sha256(img.tobytes()).hexdigest(),
            "sha256_f32": hashlib.sha256(img.astype(np.float32).tobytes()).hexdigest(),
            "row_fsum": [math.fsum(row.ravel().tolist()).hex() for row in img]}


if __name__ == "__main__":
    names = sys.argv[1:] or list(FRAMES)
    meta = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in names:
        scene, spp = FRAMES[name]
        jd = O.load_render_desc(scene)
        t = time.perf_counter()
        img, st = O.render_file(scene, spp=spp, seed=24)
        dt = time.perf_counter() - t
        d = digest(img)
        meta[name] = {"scene": scene, "spp": spp, "seed": 24, "width": jd["width"], "height": jd["height"],
                      "max_depth": jd["maxDepth"], "stats": st, "mean": float(img.mean()), **d}
        print(f"{name}: {img.shape} {st['rays']} rays in {dt:.1f} s  sha256_f64 {d['sha256_f64'][:16]}", flush=True)
        with open(OUT, "w") as f:
            json.dump(meta, f, indent=0)
            f.write("\n")
