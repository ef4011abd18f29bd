"""Random scene descriptors for the fuzz tests: every material, colour map, transform kind, both light kinds, both
projections, axis-aligned cameras (zero direction components), meshes from data/.  Used by the GPU parity fuzz
(tests/test_gpu_parity.py: HIP kernels against the C oracle) and by the CPU differential test of the two restatements
(tests/test_oracle.py: C oracle against the plain-Python one)."""
import os


def rand_scene(rng, more_meshes=()):
    def v(lo, hi): return {k: float(rng.uniform(lo, hi)) for k in "xyz"}
    def col(hi=1.5): return {k: float(rng.uniform(0, hi)) for k in "rgb"}
    def cmap(allow_checker):
        if allow_checker and rng.random() < 0.4:
            return {"type": "checker", "color1": col(), "color2": col(2), "size": float(rng.choice([0.25, 0.3, 0.5, 1.0]))}
        return {"type": "flat", "color": col(2)}
    def material(allow_checker=True):
        k = rng.choice(["diffuse", "plastic", "mirror", "transparent", "emmit"], p=[0.35, 0.3, 0.12, 0.13, 0.1])
        if k == "diffuse": return {"type": "diffuse", "cd": cmap(allow_checker)}
        if k == "plastic": return {"type": "plastic", "cd": cmap(allow_checker), "ior": float(rng.uniform(1.1, 2.5))}
        if k == "mirror": return {"type": "mirror", "ior": float(rng.uniform(0.1, 2.0))}
        if k == "transparent": return {"type": "transparent", "ior": float(rng.uniform(1.1, 2.0))}
        return {"type": "emmit", "ce": col(1)}
    def xform(depth=0):
        k = rng.choice(["translate", "scale", "rotateX", "rotateY", "rotateZ", "rotate", "sequence"] if depth < 2 else ["translate", "rotateY"])
        if k == "translate": return {"type": k, "vector": v(-0.6, 0.6)}
        if k == "scale": return {"type": k, "vector": {a: float(rng.uniform(0.3, 0.8)) for a in "xyz"}}
        if k == "rotate": return {"type": k, "axis": {"x": float(rng.uniform(0.2, 1)), "y": float(rng.uniform(0.2, 1)), "z": float(rng.uniform(-1, 1))}, "angle": float(rng.uniform(-3, 3))}
        if k == "sequence": return {"type": k, "transforms": [xform(depth + 1) for _ in range(int(rng.integers(1, 4)))]}
        return {"type": k, "angle": float(rng.uniform(-3, 3))}
    objs = []
    for _ in range(int(rng.integers(0, 4))):     # spheres never get a checker: their UV needs atan/acos
        objs.append({"geometry": {"type": "sphere", "center": v(-0.8, 0.8), "radius": float(rng.uniform(0.1, 0.5))}, "material": material(False)})
    for _ in range(int(rng.integers(0, 4))):
        n = v(-1, 1)
        objs.append({"geometry": {"type": "plane", "point": v(-1.2, 1.2), "normal": n, "tangent": v(-1, 1)}, "material": material()})
    for _ in range(int(rng.integers(0, 3))):
        f = str(rng.choice(["data/cube.obj", "data/cone.obj", "data/torus.obj", "data/uvcube.obj", "data/uvtorus.obj"] + list(more_meshes)))
        objs.append({"geometry": {"type": "mesh", "fileName": f, "transform": xform()}, "material": material(f.startswith("data/uv"))})
    rng.shuffle(objs)
    lights = []
    for _ in range(int(rng.integers(0, 4))):
        if rng.random() < 0.4: lights.append({"type": "directional", "direction": v(-1, 1), "color": col(1.2)})
        else: lights.append({"type": "point", "position": v(-1, 1), "color": col(60), "radius": float(rng.uniform(0.05, 0.3))})
    axis_aligned = rng.random() < 0.35           # camera on an axis looking down it: zero direction components
    pos = {"x": 0.0, "y": 0.0, "z": -2.5} if axis_aligned else v(-2.5, 2.5)
    cam = {"position": pos, "target": {"x": 0.0, "y": 0.0, "z": 0.0} if axis_aligned else v(-0.3, 0.3), "up": {"x": 0, "y": 1, "z": 0},
           "projection": {"type": str(rng.choice(["perspective", "perspective", "orthographic"])), "fovy": float(rng.uniform(0.5, 1.2)),
                          "width": 2, "height": 2, "near": 0.1}}
    scale = int(os.environ.get("RAYHS_FUZZ_SCALE", "1"))  # soak runs: larger frames reach the tile-ordering path
    w, h = (int(rng.integers(3, 9)) * 4 * scale, int(rng.integers(3, 9)) * 2 * scale)
    return {"width": w, "height": h, "maxDepth": int(rng.integers(0, 5)), "camera": cam, "scene": {"objects": objs, "lights": lights}}
