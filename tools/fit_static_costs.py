"""Fit the weights of the static first-frame tile estimate (rayhs_static_costs) to measured chunk times (development aid).

    GPU box (a -DRAYHS_DEV_KNOBS build WITH profiles/r05/experiments/static_cost_features.patch applied: the shipped estimate computes
    four of the eight features only and has no feature dump):   RAYHS_LIB=build_exp/cur/librayhs_hip.so python tools/fit_static_costs.py collect
        -> profiles/r05/experiments/tile_features/tile_features_<workload>.npz: per 8x8 tile the longest chunk time a steady-state frame measured (100 MHz ticks) and
           the estimate's eight raw features (RAYHS_DUMP_TILE_COSTS + RAYHS_DUMP_RAW_FEATURE=8)
    anywhere:                              python tools/fit_static_costs.py fit
        -> least-squares weights (costs in units of the median wall tile), the rank correlation of old and new estimate, and the
           makespan of a slowest-first list schedule of the measured costs under either order (4096 waves, spp-many... chunks per tile)
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); os.chdir(ROOT)
W = {"dragon": ("scenes/dragon_1080p.json", 16), "dragon_low": ("scenes/dragon_low_1080p.json", 4), "cornell": ("scenes/cornell_1080p_d0.json", 1)}
NAMES = ["boxes_met", "box_rays", "shadow_cross", "shadow_boxes", "refl", "trans", "refl_mesh", "refl_boxes"]


def collect():
    from rayhs_amd import build_rendering, lib as L
    os.environ["RAYHS_DUMP_RAW_FEATURE"] = "8"
    for name, (scene, spp) in W.items():
        path = f"/tmp/costs_{name}.bin"
        if os.path.exists(path):
            os.unlink(path)
        r = build_rendering(scene)
        for i in range(4):
            if i == 3:
                os.environ["RAYHS_DUMP_TILE_COSTS"] = path
                L.load().rayhs_config_reload()
            r.ray_trace(spp=spp, out_format=L.OUT_F32)
        os.environ.pop("RAYHS_DUMP_TILE_COSTS")
        L.load().rayhs_config_reload()
        r.close()
        buf = open(path, "rb").read()
        n, tiles_x = (int(v) for v in np.frombuffer(buf[:8], dtype=np.int32))
        m = np.frombuffer(buf[16:16 + 4 * n], dtype=np.float32)
        f = np.frombuffer(buf[16 + 4 * n:16 + 4 * n + 32 * n], dtype=np.float32).reshape(n, 8)
        np.savez_compressed(f"gpurun_out/tile_features_{name}.npz", measured=m, feat=f, spp=spp, tiles_x=tiles_x)
        print(name, n, "tiles; measured median", float(np.median(m)), "max", float(m.max()), "feature means", f.mean(axis=0).round(2))


def estimate(f, w):
    box = np.where(f[:, 1] > 0, np.where(f[:, 0] < 40, w["box_lo"] * f[:, 0], np.minimum(w["box_cap"], w["box_lo"] * 40 + w["box_hi"] * (f[:, 0] - 40))), 0.0)
    return (1.0 + box + w["shadow_cross"] * f[:, 2] + w["shadow_box"] * np.minimum(f[:, 3], 60) + w["refl"] * f[:, 4] + w["trans"] * f[:, 5]
            + w["refl_mesh"] * f[:, 6] + w["refl_box"] * np.minimum(f[:, 7], 60))


def makespan(cost, order_key, chunks_per_tile, waves=4096):
    """slowest-first list schedule: every tile is `chunks_per_tile` chunks of cost[tile]; chunks are handed out in the order of order_key (descending)"""
    import heapq
    order = np.argsort(-order_key, kind="stable")
    free = [0.0] * waves
    heapq.heapify(free)
    end = 0.0
    for t in order:
        for _ in range(chunks_per_tile):
            s = heapq.heappop(free)
            e = s + float(cost[t])
            end = max(end, e)
            heapq.heappush(free, e)
    return end


def fit():
    old = dict(box_lo=0.045, box_hi=0.26, box_cap=16.0, shadow_cross=0.25, shadow_box=0.0, refl=2.5, trans=1.5, refl_mesh=3.0, refl_box=0.0)
    data = {}
    for name in W:
        p = f"profiles/r05/experiments/tile_features/tile_features_{name}.npz"
        if os.path.exists(p):
            d = np.load(p)
            data[name] = (d["measured"].astype(np.float64), d["feat"].astype(np.float64), int(d["spp"]))
    rows, rhs = [], []
    for name in ("dragon", "dragon_low"):
        if name not in data:
            continue
        m, f, spp = data[name]
        unit = np.median(m[(f[:, 1] == 0) & (f[:, 4] == 0)])          # a plain wall tile
        y = m / unit - 1.0
        lo = np.where((f[:, 1] > 0) & (f[:, 0] < 40), f[:, 0], np.where(f[:, 1] > 0, 40.0, 0.0))
        hi = np.where((f[:, 1] > 0) & (f[:, 0] >= 40), f[:, 0] - 40, 0.0)
        X = np.stack([lo, hi, f[:, 2], np.minimum(f[:, 3], 60), f[:, 4], f[:, 5], f[:, 6], np.minimum(f[:, 7], 60)], axis=1)
        rows.append(X); rhs.append(y)
    X, y = np.concatenate(rows), np.concatenate(rhs)
    from scipy.optimize import nnls
    coef, _ = nnls(X, y)
    new = dict(box_lo=coef[0], box_hi=coef[1], box_cap=1e9, shadow_cross=coef[2], shadow_box=coef[3], refl=coef[4], trans=coef[5], refl_mesh=coef[6], refl_box=coef[7])
    print("non-negative least squares on the dragon scenes (cost in wall-tile units - 1):")
    for k, v in new.items():
        print(f"  {k:13s} old {old[k]:8.3f}   fitted {v:8.3f}")
    rk = lambda v: np.argsort(np.argsort(v)).astype(np.float64)
    for name, (m, f, spp) in data.items():
        cpt = max(1, spp * 64 // 64)
        line = [f"{name:11s}"]
        for tag, w in (("old", old), ("fitted", new)):
            e = estimate(f, w)
            rho = np.corrcoef(rk(m), rk(e))[0, 1]
            line.append(f"{tag}: rank corr {rho:.3f}, makespan {makespan(m, e, cpt) / 100:.0f} us")
        line.append(f"measured order: {makespan(m, m, cpt) / 100:.0f} us, scan order: {makespan(m, -np.arange(len(m), dtype=np.float64), cpt) / 100:.0f} us")
        print("  ".join(line))
    return new


if __name__ == "__main__":
    {"collect": collect, "fit": fit}[sys.argv[1] if len(sys.argv) > 1 else "fit"]()
