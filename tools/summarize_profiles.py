"""Condense rocprofv3 output under gpurun_out/ into the small files kept in profiles/<round>/<workload>/.

    python tools/summarize_profiles.py r02 dragon
Copies the --stats kernel summary, averages the PMC passes per kernel, and derives HBM traffic per launch as
MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE are in KiB and come from separate passes; on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x, so the read side is given both raw and doubled -- the kernel's
reads are mostly narrow gathers, for which the factor is uncalibrated).  `dispatch` comes from the code object's
metadata (tools/kinfo.sh) and from the launch line the library prints, not from rocprof's columns.
"""
import collections, csv, glob, json, os, re, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
wl = sys.argv[2] if len(sys.argv) > 2 else "dragon"
src, pre = "gpurun_out", f"gpurun_out/{tag}_{wl}"
dst = os.path.join("profiles", tag, wl)
os.makedirs(dst, exist_ok=True)
out = {"workload": wl, "frames_in_flight": 1}
if os.path.exists(pre + "_kernel_source.sha256"):
    out["kernel_source_sha256"] = open(pre + "_kernel_source.sha256").read().strip()
def newest(pattern):
    """gpurun merges a run's files INTO the local gpurun_out/: files of earlier runs stay beside the new ones (they are
    named by pid).  Only the newest file of a kind belongs to the run being summarised."""
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    return files[-1:]


for f in newest(f"{pre}_trace/*/*_kernel_stats.csv"):
    rows = [r for r in csv.DictReader(open(f))]
    keep = [r for r in rows if r["Name"].startswith("rayhs_") or float(r["Percentage"]) > 0.5]
    with open(os.path.join(dst, "kernel_stats.csv"), "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader(); w.writerows(keep)
    out["kernel_stats"] = {r["Name"]: {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6, "pct": float(r["Percentage"])} for r in keep}
# the trace kernel over the TIMED launches only (the --stats average also covers the set-up and warm-up frames)
for f in newest(f"{pre}_trace/*/*_kernel_trace.csv"):
    try:
        tb = json.load(open(pre + "_trace_bench.json"))
    except Exception:
        tb = {}
    steps = tb.get("steps", 0)
    rows = [r for r in csv.DictReader(open(f)) if r["Kernel_Name"].startswith("rayhs_trace") and "count" not in r["Kernel_Name"]]
    after = tb.get("trace_launches_after_timed", 1 if tb.get("first_frame_ms") is not None else 0)  # the first-frame launch on a fresh handle (and --verify / the 4K leg) follow the timed loop
    if steps and len(rows) >= steps + after:
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows[len(rows) - after - steps:len(rows) - after]]
        out["timed_launches"] = {"kernel": rows[-1]["Kernel_Name"], "launches": steps, "avg_ms": sum(d) / len(d), "min_ms": min(d), "max_ms": max(d),
                                 "bench_events_avg_ms": tb.get("kernel_ms_rank0"), "first_frame_ms": tb.get("first_frame_ms")}
# What one wave64 instruction costs the unit that executes it (tools/micro/issue_rates.hip -> profiles/r05/micro/issue_rates.txt: SIMD busy span /
# instructions executed on it, residency measured, not assumed).  VALU, in cycles of its SIMD: fp32 add / mul / fma / min / max / mov 2.2 (the guide's
# nominal 2: a SIMD is 32 lanes wide); fp64 add / mul / fma, packed fp32, every integer, compare, select and convert instruction 4.1 (nominal 4);
# fp64 rcp / rsq / sqrt 16.  SCALAR: a CU has ONE scalar unit for its four SIMDs; it retires 0.95 instructions per cycle (4.2 SIMD-cycles per
# instruction with all four SIMDs asking), whatever the waves per SIMD.  Vector and scalar instructions of different waves issue side by side,
# so a launch is bounded by the busier of the two: roofline.frac = max(valu_pipe_frac, scalar_unit_frac).
VALU_COST = {"f32_arith": 2.0, "f64_arith": 4.0, "f64_trans": 16.0, "f32_trans": 8.0, "int_cvt": 4.0, "other_nominal": 2.0, "other_measured": 4.0}
SCALAR_COST_CU_CYCLES = 1.05   # per SALU / scalar-memory / branch instruction
pmc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in [g for d in glob.glob(f"{pre}_pmc_*") for g in newest(f"{d}/*/*_counter_collection.csv")]:
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("rayhs_trace") and "count" not in r["Kernel_Name"]:
            pmc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out["pmc_mean_per_launch"] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in pmc.items()}
for k, d in out["pmc_mean_per_launch"].items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        rd, wr = d["FETCH_SIZE"] * 1024, d["WRITE_SIZE"] * 1024
        d["hbm_read_bytes_raw"] = rd
        d["hbm_write_bytes"] = wr
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
    if "SQ_INSTS_SALU" in d and "SQ_INSTS_VALU" in d:
        d["salu_per_valu"] = d["SQ_INSTS_SALU"] / d["SQ_INSTS_VALU"]
    # Cycles the vector PIPE needs for the instructions issued (MI355X_MICROARCH.md: a SIMD is 32 lanes wide, so a
    # wave64 fp32 / integer instruction occupies it for 2 cycles, fp64 add / mul / fma for 4 (half rate), fp64 rcp / sqrt
    # for 16), against the SIMD-cycles of the launch (GRBM_GUI_ACTIVE is summed over the 8 XCDs).  SQ_ACTIVE_INST_VALU --
    # `valu_busy` -- counts 4 cycles per instruction whatever its kind: it is ISSUE time of the waves, not pipe occupancy.
    f64 = [d.get("SQ_INSTS_VALU_" + k) for k in ("ADD_F64", "MUL_F64", "FMA_F64")]
    if all(v is not None for v in f64) and "SQ_INSTS_VALU_TRANS_F64" in d and "GRBM_GUI_ACTIVE" in d:
        arith, trans = sum(f64), d["SQ_INSTS_VALU_TRANS_F64"]
        pipe_cycles = 4 * arith + 16 * trans + 2 * (d["SQ_INSTS_VALU"] - arith - trans)
        simd_cycles = d["GRBM_GUI_ACTIVE"] / 8 * 256 * 4
        d["valu_f64_share_of_insts"] = (arith + trans) / d["SQ_INSTS_VALU"]
        d["valu_pipe_frac"] = pipe_cycles / simd_cycles
        d["fp64_flops_per_launch"] = 64.0 * (f64[0] + f64[1] + 2 * f64[2])   # wave-level instructions x 64 lanes (an upper bound: idle lanes count)
    if "SQ_INSTS_BRANCH" in d and "SQ_INSTS_VALU" in d:
        d["branches_per_valu"] = d["SQ_INSTS_BRANCH"] / d["SQ_INSTS_VALU"]
    # The two roofs (see VALU_COST above).  valu_pipe_frac prices "other" vector instructions -- moves, compares, selects, lane reads: what no
    # class counter covers -- at the guide's nominal 2 cycles (a lower bound: measured, only v_mov is that cheap); valu_pipe_frac_measured_costs
    # at the 4 cycles compares and selects really take (an upper bound).  The fp32 / integer / convert split needs the round-5 `mix2` pass.
    classes = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")
    if all(c in d for c in classes) and all(v is not None for v in f64) and "SQ_INSTS_VALU_TRANS_F64" in d and "GRBM_GUI_ACTIVE" in d:
        insts = sum(d[c] for c in classes)     # (SQ_INSTS_FLAT counts the same global_* instructions as VMEM_RD + VMEM_WR)
        arith, trans = sum(f64), d["SQ_INSTS_VALU_TRANS_F64"]
        simd_cycles = d["GRBM_GUI_ACTIVE"] / 8 * 256 * 4
        cu_cycles = d["GRBM_GUI_ACTIVE"] / 8 * 256
        d["insts_per_launch"] = insts
        d["simd_cycles_per_launch"] = simd_cycles
        d["insts_per_simd_cycle"] = insts / simd_cycles
        f32 = [d.get("SQ_INSTS_VALU_" + k) for k in ("ADD_F32", "MUL_F32", "FMA_F32")]
        if all(v is not None for v in f32):
            f32a, f32t = sum(f32), d.get("SQ_INSTS_VALU_TRANS_F32", 0.0)
            intc = d.get("SQ_INSTS_VALU_INT32", 0.0) + d.get("SQ_INSTS_VALU_INT64", 0.0) + d.get("SQ_INSTS_VALU_CVT", 0.0)
            other = max(d["SQ_INSTS_VALU"] - arith - trans - f32a - f32t - intc, 0.0)
            base = VALU_COST["f32_arith"] * f32a + VALU_COST["f64_arith"] * arith + VALU_COST["f64_trans"] * trans + VALU_COST["f32_trans"] * f32t + VALU_COST["int_cvt"] * intc
            d["valu_class_counts"] = {"f32_arith": f32a, "f64_arith": arith, "f64_trans": trans, "f32_trans": f32t, "int_cvt": intc, "other": other}
            d["valu_pipe_frac"] = (base + VALU_COST["other_nominal"] * other) / simd_cycles
            d["valu_pipe_frac_measured_costs"] = (base + VALU_COST["other_measured"] * other) / simd_cycles
        scalar = d["SQ_INSTS_SALU"] + d["SQ_INSTS_SMEM"] + d["SQ_INSTS_BRANCH"]
        d["scalar_unit_insts_per_launch"] = scalar
        d["scalar_unit_frac"] = SCALAR_COST_CU_CYCLES * scalar / cu_cycles
        d["roofline_frac"] = max(d.get("valu_pipe_frac", 0.0), d["scalar_unit_frac"])
        d["roofline_bound"] = "valu" if d.get("valu_pipe_frac", 0.0) >= d["scalar_unit_frac"] else "scalar-issue"
    for cache in ("DCACHE", "ICACHE"):
        if f"SQC_{cache}_REQ" in d and f"SQC_{cache}_MISSES" in d and d[f"SQC_{cache}_REQ"] > 0:
            d[f"scalar_{cache.lower()}_hit_rate"] = 1.0 - (d[f"SQC_{cache}_MISSES"] + d.get(f"SQC_{cache}_MISSES_DUPLICATE", 0.0)) / d[f"SQC_{cache}_REQ"]
# dispatch: the code object's own metadata + the launch the library reports
disp = {}
try:
    txt = subprocess.run(["bash", "tools/kinfo.sh"], capture_output=True, text=True, timeout=600).stdout
    for line in txt.splitlines():
        p = line.split()
        if len(p) == 6 and p[0].startswith("rayhs_trace"):
            disp[p[0]] = {"scratch_bytes_per_lane": int(p[1]), "sgpr_count": int(p[2]), "sgpr_spill_count": int(p[3]), "vgpr_count": int(p[4]), "vgpr_spill_count": int(p[5])}
except Exception as e:
    disp["error"] = str(e)
if os.path.exists(pre + "_launch.txt"):
    m = re.search(r"variant (\d+), lds (\d+) B, (\d+) blocks/CU, stack cap (\d+)", open(pre + "_launch.txt").read())
    if m:
        disp["launch"] = {"variant": int(m.group(1)), "lds_bytes_per_workgroup": int(m.group(2)), "workgroups_per_cu": int(m.group(3)), "stack_cap": int(m.group(4)),
                          "workgroup_size": 256, "waves_per_simd": int(m.group(3))}
out["dispatch"] = disp
out["valu_cost_simd_cycles"] = VALU_COST
out["scalar_cost_cu_cycles"] = SCALAR_COST_CU_CYCLES
for name, to in ((f"{tag}_{wl}_bench.json", "bench.json"), (f"{tag}_{wl}_trace_bench.json", "trace_bench.json")):
    p = os.path.join(src, name)
    if os.path.exists(p):
        shutil.copy(p, os.path.join(dst, to))
json.dump(out, open(os.path.join(dst, "summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "pmc_mean_per_launch"}, indent=1)[:1500])
