"""Ad-hoc GPU probe: parity vs the oracle and a first timing (development aid, not a test)."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from rayhs_amd import build_rendering, lib as L
from oracle import oracle as O


def report(tag, g, o):
    d = np.abs(g - o)
    bad = (d > 1e-4).any(axis=2)
    print(f"{tag}: max|d|={d.max():.3e} exact={np.array_equal(g, o)} n_diff_px={int((d>0).any(axis=2).sum())} n_bad_px(>1e-4)={int(bad.sum())}", flush=True)
    if bad.any():
        ys, xs = np.nonzero(bad)
        for y, x in list(zip(ys, xs))[:5]:
            print("   bad", x, y, g[y, x], o[y, x])


def scene_case(path, w, h, spp=1, mesh=None, depth=None):
    jd = O.load_render_desc(path)
    if mesh:
        for ob in jd["scene"]["objects"]:
            if ob["geometry"]["type"] == "mesh":
                ob["geometry"]["fileName"] = mesh
    tmp = "/tmp/_scene.json"
    json.dump(jd, open(tmp, "w"))
    r = build_rendering(tmp, width=w, height=h, max_depth=depth)
    sc = O.Scene(jd["scene"]); cam = O.make_camera(jd["camera"])
    md = jd["maxDepth"] if depth is None else depth
    t = time.time(); oimg, ost = sc.render(cam, w, h, md, spp=spp); tcpu = time.time() - t
    name = f"{path.split('/')[-1]} {w}x{h} spp{spp} mesh={mesh}"
    for flags, tag in ((0, "exact"), (L.FLAG_FMA, "fma")):
        g, st = r.ray_trace(spp=spp, flags=flags)
        report(f"{name} [{tag}] kernel {st['kernel_ms']:.2f} ms (oracle {tcpu*1e3:.0f} ms, {len(__import__('os').sched_getaffinity(0))} thr)", g, oimg)
    g, st = r.ray_trace(spp=spp, flags=L.FLAG_COUNT)
    print("   counted (pruned)   :", {k: v for k, v in st.items() if k != "kernel_ms"}, flush=True)
    O.lib().orc_set_shadow_exhaustive(1)
    _, ost2 = sc.render(cam, w, h, md, spp=spp)
    O.lib().orc_set_shadow_exhaustive(0)
    g2, st2 = r.ray_trace(spp=spp, flags=L.FLAG_COUNT | L.FLAG_EXHAUSTIVE)
    print("   counted (exhaustive):", {k: v for k, v in st2.items() if k != "kernel_ms"})
    print("   oracle (exhaustive) :", ost2)
    print("   exhaustive image == pruned image:", np.array_equal(g, g2), " counters equal oracle:",
          all(st2[k] == ost2[k] for k in ost2 if k != "rays") , flush=True)
    r.close()


if __name__ == "__main__":
    print("devices", L.load().rayhs_device_count(), flush=True)
    scene_case("data/cornellBox.json", 256, 256)
    scene_case("data/dragon.json", 256, 256)
    scene_case("data/outScene.json", 320, 180)
    scene_case("data/transform.json", 128, 128)
    scene_case("data/dragon.json", 192, 108, spp=4, mesh="data/dragon_low.obj")
    # timing of the headline config
    jd = O.load_render_desc("data/dragon.json")
    jd["scene"]["objects"][0]["geometry"]["fileName"] = "data/dragon.obj"
    json.dump(jd, open("/tmp/_d.json", "w"))
    r = build_rendering("/tmp/_d.json", width=1920, height=1080)
    for spp in (1, 16):
        for flags, tag in ((L.FLAG_FMA, "fma"), (0, "exact")):
            for rep in range(2):
                g, st = r.ray_trace(spp=spp, flags=flags, out_format=L.OUT_F32)
                print(f"dragon.obj 1080p spp{spp} [{tag}] kernel {st['kernel_ms']:.2f} ms", flush=True)
    g, st = r.ray_trace(spp=16, flags=L.FLAG_COUNT, out_format=L.OUT_F32)
    print("counts spp16:", st, flush=True)
