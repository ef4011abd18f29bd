#!/usr/bin/env python3
"""The per-workload table of DESIGN.md section 4 from a round's committed profiles (development aid):   python3 tools/design_table.py r06
Reads profiles/<round>/<workload>/summary.json (tools/summarize_profiles.py) and prices the two roofs exactly as bench.py does."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
tag = sys.argv[1]
W = [("cornell", "[1] Cornell"), ("dragon_low", "[2] dragon_low"), ("dragon", "[3] dragon"), ("dragon4k", "[4] dragon 4K/64")]
rows = {}
for wl, _ in W:
    s = json.load(open(os.path.join(ROOT, "profiles", tag, wl, "summary.json")))
    b = json.load(open(os.path.join(ROOT, "profiles", tag, wl, "bench.json")))
    c = s["pmc_mean_per_launch"]["rayhs_trace_exact"]
    t = s["timed_launches"]
    ms = t["avg_ms"]; sec = ms * 1e-3
    vc = c["valu_class_counts"]; cost = s["valu_cost_simd_cycles"]
    nominal = (vc["f32_arith"] * cost["f32_arith"] + vc["f64_arith"] * cost["f64_arith"] + vc["f64_trans"] * cost["f64_trans"] + vc["f32_trans"] * cost["f32_trans"]
               + vc["int_cvt"] * cost["int_cvt"])
    valu_nom = (nominal + vc["other"] * cost["other_nominal"]) / (1024 * 2.4e9 * sec)
    valu_meas = (nominal + vc["other"] * cost["other_measured"]) / (1024 * 2.4e9 * sec)
    scal = c["scalar_unit_insts_per_launch"] * s["scalar_cost_cu_cycles"] / (256 * 2.4e9 * sec)
    rr = b["roofline"]
    alg = rr["hbm_contract"]["algorithmic_bytes_per_launch"]
    rows[wl] = {
        "hash": s["kernel_source_sha256"][:8],
        "ms": f"**{ms:.3f}** / {t['bench_events_avg_ms']:.3f}",
        "insts": f"{c['insts_per_launch'] / 1e9:.3f} G; {c['scalar_unit_insts_per_launch'] / 1e9:.3f}",
        "scalar": f"**{scal:.2f}**",
        "valu": f"{valu_nom:.2f}; {valu_meas:.2f}",
        "contract": f"{alg / sec / 1e9 / 8000:.2f}",
        "hbm": f"{c['hbm_bytes_per_launch_read_doubled'] / 1e9:.2f} GB; {c['hbm_bytes_per_launch_read_doubled'] / sec / 1e9 / 8000:.3f}; {2 * c['hbm_read_bytes_raw'] / sec / 1e9 / 8000:.3f}; {c['hbm_write_bytes'] / 1e6:.0f} MB",
        "fp64": f"{c['fp64_flops_per_launch'] / sec / 1e12:.1f} TFLOP/s; {c.get('valu_lane_utilisation', float('nan')):.2f}",
        "life": f"{c.get('SQ_ACTIVE_INST_ANY_per_wave_cycle', float('nan')):.2f} / {c.get('SQ_WAIT_INST_ANY_per_wave_cycle', float('nan')):.2f} / {c.get('SQ_WAIT_ANY_per_wave_cycle', float('nan')):.2f}",
        "dcache": f"{c.get('scalar_dcache_hit_rate', float('nan')):.2f}; {c['SQC_DCACHE_MISSES'] / c['SQC_DCACHE_REQ']:.3f}" if 'SQC_DCACHE_REQ' in c else "-",
        "valu_M": f"{c['SQ_INSTS_VALU'] / 1e6:.1f}", "bench_ms_per_step": f"{b['ms_per_step']:.3f}", "bench_value": f"{b['value']:.0f}",
    }
for key in ("hash", "ms", "insts", "scalar", "valu", "contract", "hbm", "fp64", "life", "dcache", "valu_M", "bench_ms_per_step", "bench_value"):
    print("| " + key + " | " + " | ".join(rows[wl][key] for wl, _ in W) + " |")
