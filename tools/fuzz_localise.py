#!/usr/bin/env python3
"""Localise a fuzz scene's mismatch against the oracle (development aid, GPU box):  python3 tools/fuzz_localise.py gpurun_out/fuzz_<seed>.json <spp>
Renders the scene at several maxDepth / spp, then with each object and each light removed in turn, and prints how many pixels differ each time."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.chdir(ROOT)
import numpy as np
import rayhs_amd
from oracle import oracle as O

path, spp0 = sys.argv[1], int(sys.argv[2])
jd0 = json.load(open(path))


def diff(jd, spp, depth, tag):
    p = "gpurun_out/_loc.json"
    json.dump(jd, open(p, "w"))
    want, _ = O.render_file(p, spp=spp, max_depth=depth)
    got, _ = rayhs_amd.ray_trace_file(p, spp=spp, max_depth=depth)
    same = (got == want) | (np.isnan(got) & np.isnan(want))
    bad = np.argwhere(~same.all(axis=2))
    msg = ""
    if len(bad):
        y, x = bad[0]
        msg = f"  first ({y},{x}) got {got[y, x].tolist()} want {want[y, x].tolist()}"
    print(f"{tag}: spp {spp} depth {depth}: {len(bad)} pixels differ{msg}", flush=True)
    return len(bad)


for depth in (0, 1, 2, 3):
    for spp in sorted({1, spp0}):
        diff(jd0, spp, depth, "whole scene")
n_obj = len(jd0["scene"]["objects"])
for k in range(n_obj):
    jd = copy.deepcopy(jd0)
    o = jd["scene"]["objects"].pop(k)
    diff(jd, spp0, jd0["maxDepth"], f"without object {k} ({o['geometry']['type']} {o['geometry'].get('fileName', '')} {o['material']['type']})")
for k in range(len(jd0["scene"]["lights"])):
    jd = copy.deepcopy(jd0)
    l = jd["scene"]["lights"].pop(k)
    diff(jd, spp0, jd0["maxDepth"], f"without light {k} ({l['type']})")
for k in range(n_obj):
    jd = copy.deepcopy(jd0)
    jd["scene"]["objects"] = [jd0["scene"]["objects"][k]]
    diff(jd, spp0, jd0["maxDepth"], f"only object {k}")
