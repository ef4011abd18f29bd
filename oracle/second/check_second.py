"""Cross-check of the two restatements of the reference, run in the BUILD CONTAINER only.

    python oracle/second/check_second.py            # asserts + writes tests/golden/second_*.npy

(a) asserts that oracle/second/rayhs_second.py (plain Python, written from the Haskell alone) and
    oracle/rayhs_oracle.c (the C oracle every parity test uses) produce BIT-IDENTICAL images on the seven
    golden frames (whole) and on pixel windows of the four BASELINE configurations at their own size and
    sample count -- every material, both light kinds, both projections, all transform kinds, textures,
    the mesh tree, and the multi-sample path;
(b) asserts that the committed golden frames tests/golden/<name>.npy ARE the Python restatement's images,
    and writes its BASELINE windows to tests/golden/second_<name>.npy, so that the CPU test-suite
    (tests/test_oracle.py) re-checks the C oracle -- and the GPU test-suite the HIP kernels -- against
    vectors that did NOT come from the C oracle.
Both restatements remain restatement-derived: parity with GHC's output stays UNPINNED.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
sys.setrecursionlimit(10000)
from oracle import oracle as O  # noqa: E402
from oracle.second import rayhs_second as S  # noqa: E402

# name: (scene, width, height, spp, rows, cols)   -- rows / cols None = the whole frame
CASES = {
    # the seven golden frames of tests/golden/make_golden.py, whole
    "cornell_96": ("scenes/cornell_96.json", 96, 96, 1, None, None),
    "cornell_256": ("scenes/cornell_256.json", 256, 256, 1, None, None),   # BASELINE configs[0]
    "dragon_superlow_96": ("scenes/dragon_superlow_96.json", 96, 96, 1, None, None),
    "dragon_superlow_96_spp4": ("scenes/dragon_superlow_96.json", 96, 96, 4, None, None),
    "outscene_128x72": ("scenes/outscene_128x72.json", 128, 72, 1, None, None),
    "transform_64": ("scenes/transform_64.json", 64, 64, 1, None, None),
    "texture_earth_96": ("scenes/texture_earth_96.json", 96, 96, 1, None, None),
    # windows of the BASELINE configurations at their own size and sample count
    "cornell_1080p_window": ("scenes/cornell_1080p_d0.json", 1920, 1080, 1, [0, 270, 539, 540, 1079], range(0, 1920, 7)),
    "dragon_low_1080p_window": ("scenes/dragon_low_1080p.json", 1920, 1080, 4, [400, 640], range(760, 1160, 3)),
    "dragon_1080p_window": ("scenes/dragon_1080p.json", 1920, 1080, 16, [330, 600, 800], range(850, 1070, 5)),
    "dragon_4k_window": ("scenes/dragon_4k.json", 3840, 2160, 64, [1200], range(1800, 2040, 12)),
}
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    meta = {}
    for name, (scene, w, h, spp, rows, cols) in CASES.items():
        t = time.perf_counter()
        second = np.array(S.render_file(scene, width=w, height=h, spp=spp, seed=24, rows=rows, cols=cols), dtype=np.float64)
        dt = time.perf_counter() - t
        if rows is None:
            first, _ = O.render_file(scene, width=w, height=h, spp=spp, seed=24)
        else:
            first = np.stack([O.render_file(scene, width=w, height=h, spp=spp, seed=24, row0=y, row1=y + 1)[0][0][list(cols)] for y in rows])
        same = (first == second) | (np.isnan(first) & np.isnan(second))
        n_bad = int((~same).any(axis=2).sum())
        print(f"{name}: {w}x{h} spp {spp}: python {dt:.1f} s, pixels differing from the C oracle: {n_bad}"
              + ("" if not n_bad else f", max |d| {np.nanmax(np.abs(first - second)):.3e}"), flush=True)
        assert n_bad == 0, name
        if rows is None:     # a golden frame: the committed fixture (made by the C oracle) must be this very image
            gold = np.load(os.path.join(OUT, name + ".npy"))
            assert gold.shape == second.shape and bool(((gold == second) | (np.isnan(gold) & np.isnan(second))).all()), name
        else:                # a window of a BASELINE frame: kept as a fixture that did not come from the C oracle
            np.save(os.path.join(OUT, f"second_{name}.npy"), second)
        meta[name] = {"scene": scene, "width": w, "height": h, "spp": spp, "seed": 24,
                      "rows": None if rows is None else list(rows), "cols": None if cols is None else list(cols)}
    with open(os.path.join(OUT, "second.json"), "w") as f:
        json.dump(meta, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    main()
