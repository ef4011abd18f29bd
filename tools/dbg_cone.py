import json, sys, os
import numpy as np
sys.path.insert(0, "."); 
import rayhs_amd
from rayhs_amd import lib as L
from oracle import oracle as O
def sj(objects, lights, w=40, h=30, depth=2):
    return json.dumps({"width": w, "height": h, "maxDepth": depth,
        "camera": {"position": {"x": 0, "y": 0, "z": -2.0}, "target": {"x": 0, "y": 0, "z": 0}, "up": {"x": 0, "y": 1, "z": 0},
                   "projection": {"type": "perspective", "fovy": 0.9272952180016123, "width": 2, "height": 2, "near": 0.1}},
        "scene": {"objects": objects, "lights": lights}})
FLOOR = {"geometry": {"type": "plane", "point": {"x": 0, "y": -1, "z": 0}, "normal": {"x": 0, "y": 1, "z": 0}, "tangent": {"x": 1, "y": 0, "z": 0}},
         "material": {"type": "plastic", "ior": 2, "cd": {"type": "checker", "color1": {"r": 0, "g": 0, "b": 0}, "color2": {"r": 2, "g": 2, "b": 2}, "size": 0.5}}}
PT = {"type": "point", "position": {"x": 0, "y": 0.9, "z": 0.2}, "color": {"r": 90, "g": 90, "b": 90}, "radius": 0.1}
cone = {"geometry": {"type": "mesh", "fileName": "data/cone.obj", "transform": {"type": "sequence", "transforms": [
    {"type": "scale", "vector": {"x": 0.4, "y": 0.4, "z": 0.4}}, {"type": "rotateX", "angle": 0.4}, {"type": "translate", "vector": {"x": 0, "y": -0.2, "z": 0.8}}]}},
    "material": {"type": "diffuse", "cd": {"type": "flat", "color": {"r": 1, "g": 1, "b": 1}}}}
bulb = {"geometry": {"type": "sphere", "center": {"x": 0, "y": 0.9, "z": 0.2}, "radius": 0.15}, "material": {"type": "emmit", "ce": {"r": 1, "g": 1, "b": 1}}}
open("/tmp/c.json","w").write(sj([cone, FLOOR, bulb], [PT]))
want, ost = O.render_file("/tmp/c.json")
for fl,tag in ((L.FLAG_COUNT,"fast"),(L.FLAG_COUNT|L.FLAG_EXHAUSTIVE,"exh")):
    got, st = rayhs_amd.ray_trace_file("/tmp/c.json", flags=fl)
    d = np.abs(got-want).max(axis=2)
    ys,xs = np.nonzero(d>0)
    print(tag, "ndiff", len(ys), "max", d.max(), list(zip(xs[:10],ys[:10])))
    for x,y in list(zip(xs,ys))[:4]: print("   ", x,y, got[y,x], want[y,x])
