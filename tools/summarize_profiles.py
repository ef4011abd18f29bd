"""Condense rocprofv3 output under gpurun_out/ into the small files kept in profiles/<round>/.

    python tools/summarize_profiles.py r01
Copies the --stats kernel summary, averages the PMC passes per kernel, and derives HBM traffic per
launch as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE are in KiB and come from separate
passes; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x, so the read side is given both
raw and doubled -- the kernel's reads are mostly narrow gathers, for which the factor is uncalibrated).
"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src, dst = "gpurun_out", os.path.join("profiles", tag)
os.makedirs(dst, exist_ok=True)
out = {"workload": sys.argv[2] if len(sys.argv) > 2 else "dragon"}
for f in glob.glob(f"{src}/{tag}_trace/*/*_kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f))]
    keep = [r for r in rows if r["Name"].startswith("rayhs_") or float(r["Percentage"]) > 0.5]
    with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
    out["kernel_stats"] = {r["Name"]: {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])} for r in keep}
# the same kernel over the TIMED launches only (the --stats average also covers the warm-up frames, the first of
# which run in scan order): the last `steps` dispatches of the trace kernel in the kernel trace of the bench run
for f in glob.glob(f"{src}/{tag}_trace/*/*_kernel_trace.csv"):
    try:
        steps = json.load(open(f"{src}/{tag}_trace_bench.json"))["steps"]
    except Exception:
        steps = 0
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("rayhs_trace") and "count" not in r["Kernel_Name"]]
    if steps and len(rows) >= steps:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows[-steps:]]
        out["timed_launches"] = {"kernel": rows[-1]["Kernel_Name"], "launches": steps, "avg_ms": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d),
                                 "bench_events_avg_ms": json.load(open(f"{src}/{tag}_trace_bench.json")).get("kernel_ms_rank0")}
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for f in glob.glob(f"{src}/{tag}_pmc_*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("rayhs_trace") and "count" not in r["Kernel_Name"]:
            pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[r["Kernel_Name"]] = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count")}
out["pmc_mean_per_launch"] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
out["dispatch"] = meta
for k, d in out["pmc_mean_per_launch"].items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        rd, wr = d["FETCH_SIZE"] * 1024, d["WRITE_SIZE"] * 1024
        d["hbm_bytes_per_launch_raw"] = rd + wr
        d["hbm_bytes_per_launch_read_doubled"] = 2 * rd + wr
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"])
    if "SQ_THREAD_CYCLES_VALU" in d and "SQ_ACTIVE_INST_VALU" in d:
        d["valu_lane_utilisation"] = d["SQ_THREAD_CYCLES_VALU"] / (64 * d["SQ_ACTIVE_INST_VALU"])
    if "SQ_WAVE_CYCLES" in d:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if c in d:
                d[c + "_per_wave_cycle"] = d[c] / d["SQ_WAVE_CYCLES"]
for name in (f"bench_{tag}.json", f"{tag}_trace_bench.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, name))
json.dump(out, open(os.path.join(dst, "summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
