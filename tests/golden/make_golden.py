"""Generate the golden fixtures under tests/golden/ from the CPU oracle.

RESTATEMENT-DERIVED, not reference-derived: the reference ships no expected
outputs and cannot be run here (no GHC), so these vectors pin the oracle and
the HIP path to each other and guard both against regressions.  Run from the
repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

# (name, scene file, spp)
CASES = [
    ("cornell_96", "scenes/cornell_96.json", 1),
    ("cornell_256", "scenes/cornell_256.json", 1),   # BASELINE configs[0]: data/cornellBox.json at 256x256, maxDepth 3
    ("dragon_superlow_96", "scenes/dragon_superlow_96.json", 1),
    ("dragon_superlow_96_spp4", "scenes/dragon_superlow_96.json", 4),
    ("outscene_128x72", "scenes/outscene_128x72.json", 1),
    ("transform_64", "scenes/transform_64.json", 1),
    ("texture_earth_96", "scenes/texture_earth_96.json", 1),
]

if __name__ == "__main__":
    meta = {}
    for name, path, spp in CASES:
        jd = O.load_render_desc(path)
        img, st = O.render_file(path, spp=spp, seed=24)
        np.save(os.path.join(OUT, name + ".npy"), img)
        meta[name] = {"scene": path, "spp": spp, "seed": 24, "width": jd["width"], "height": jd["height"],
                      "max_depth": jd["maxDepth"], "stats": st, "mean": float(img.mean())}
        print(name, img.shape, st["rays"])
    # sampler known answers
    L = O.lib()
    meta["sampler"] = [[seed, pixel, sample, dim, spp, L.orc_sample01(seed, pixel, sample, dim, spp).hex()]
                       for seed, pixel, sample, dim, spp in
                       [(24, 0, 0, 0, 4), (24, 0, 0, 1, 4), (24, 12345, 3, 0, 16), (24, 2073599, 15, 1, 16), (7, 99, 63, 1, 64)]]
    with open(os.path.join(OUT, "golden.json"), "w") as f:
        json.dump(meta, f, indent=1)
        f.write("\n")
