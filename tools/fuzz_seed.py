#!/usr/bin/env python3
"""One fuzz scene of tests/test_gpu_parity.py::test_random_scenes_bit_exact against the oracle (development aid, GPU box):
    [RAYHS_LIB=build_exp/<name>/librayhs_hip.so] python3 tools/fuzz_seed.py <rng seed> [...]     (rng seed = 1000 + RAYHS_FUZZ_SEED_BASE + test seed)
Prints, per kernel (counting / production), how many pixels differ from the oracle's and where; --dump keeps the scene as gpurun_out/fuzz_<seed>.json."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.chdir(ROOT)
import numpy as np
import rayhs_amd
from rayhs_amd import lib as L
from scene_fuzz import rand_scene
from oracle import oracle as O


def oracle_image(path, **kw):
    return O.render_file(path, **kw)


os.makedirs("gpurun_out", exist_ok=True)
for a in sys.argv[1:]:
    if a.startswith("--"):
        continue
    seed = int(a)
    rng = np.random.default_rng(seed)
    jd = rand_scene(rng)
    spp = int(rng.choice([1, 1, 2, 3]))
    path = f"gpurun_out/fuzz_{seed}.json"
    json.dump(jd, open(path, "w"))
    want, ost = oracle_image(path, spp=spp)
    for name, flags in (("counting", L.FLAG_COUNT), ("production", 0)):
        got, st = rayhs_amd.ray_trace_file(path, spp=spp, flags=flags)
        same = (got == want) | (np.isnan(got) & np.isnan(want))
        bad = np.argwhere(~same.all(axis=2))
        print(f"seed {seed} spp {spp} {got.shape[1]}x{got.shape[0]} lib {os.environ.get('RAYHS_LIB', 'product')} {name}: {len(bad)} pixels differ"
              + (f", first {bad[:4].tolist()} got {got[tuple(bad[0])].tolist()} want {want[tuple(bad[0])].tolist()}" if len(bad) else ""), flush=True)
