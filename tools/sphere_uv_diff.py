"""How far the sphere-UV scenes (atan / acos of src/Geometry.hs:96, device libm against glibc) are from the oracle:
PYTHONPATH=. python tools/sphere_uv_diff.py   (GPU box; output kept as profiles/r02/sphere_uv_diff.txt)"""
import json, os, numpy as np, rayhs_amd
GOLD="tests/golden"; META=json.load(open(GOLD+"/golden.json"))
for name in ["outscene_128x72","texture_earth_96"]:
    m=META[name]
    img,_=rayhs_amd.ray_trace_file(m["scene"], spp=m["spp"], seed=m["seed"])
    gold=np.load(f"{GOLD}/{name}.npy")
    d=np.abs(img-gold)
    print(name, "pixels differing:", int((d>0).any(axis=2).sum()), "of", d.shape[0]*d.shape[1], "max", d.max())
from oracle import oracle as O
for scene in ["data/outScene.json"]:
    img,_=rayhs_amd.ray_trace_file(scene)
    ref,_=O.render_file(scene)
    d=np.abs(img-ref); print(scene, "pixels differing:", int((d>0).any(axis=2).sum()), "of", d.shape[0]*d.shape[1], "max", d.max())
