"""Derive the BASELINE.json configurations from the reference's sample scenes.

The reference has no CLI override for width/height/maxDepth/mesh (SURVEY.md D3,
D7), so each configuration is an edited copy of data/cornellBox.json or
data/dragon.json.  Run from the repo root; outputs are committed under scenes/.
"""
import json
import os


def derive(src, dst, width, height, max_depth=None, mesh=None):
    with open(src) as f:
        jd = json.load(f)
    jd["width"], jd["height"] = width, height
    if max_depth is not None:
        jd["maxDepth"] = max_depth
    if mesh is not None:
        for o in jd["scene"]["objects"]:
            if o["geometry"]["type"] == "mesh":
                o["geometry"]["fileName"] = mesh
    with open(dst, "w") as f:
        json.dump(jd, f, indent=1)
        f.write("\n")


if __name__ == "__main__":
    os.makedirs("scenes", exist_ok=True)
    derive("data/cornellBox.json", "scenes/cornell_256.json", 256, 256)                         # config 1
    derive("data/cornellBox.json", "scenes/cornell_1080p_d0.json", 1920, 1080, max_depth=0)     # config 2
    derive("data/dragon.json", "scenes/dragon_low_1080p.json", 1920, 1080, mesh="data/dragon_low.obj")  # config 3 (4 spp)
    derive("data/dragon.json", "scenes/dragon_1080p.json", 1920, 1080, mesh="data/dragon.obj")  # config 4 (16 spp)
    derive("data/dragon.json", "scenes/dragon_4k.json", 3840, 2160, mesh="data/dragon.obj")     # config 5 (64 spp)
    # small parity cases
    derive("data/dragon.json", "scenes/dragon_superlow_96.json", 96, 96)
    derive("data/cornellBox.json", "scenes/cornell_96.json", 96, 96)
    derive("data/outScene.json", "scenes/outscene_128x72.json", 128, 72)
    derive("data/transform.json", "scenes/transform_64.json", 64, 64)
    # the texture colour map on the one texture the reference ships (data/texture.json
    # names two PPM files that are not in the repository)
    with open("data/texture.json") as f:
        jd = json.load(f)
    jd["width"], jd["height"] = 96, 96
    jd["scene"]["objects"] = [o for o in jd["scene"]["objects"]
                              if o["material"].get("cd", {}).get("fileName", "data/earth.ppm") == "data/earth.ppm"]
    with open("scenes/texture_earth_96.json", "w") as f:
        json.dump(jd, f, indent=1)
        f.write("\n")
