"""Means per launch of the production trace kernel over the passes of tools/pmc_extra.sh:  python tools/pmc_extra_summary.py r02 dragon"""
import collections, csv, glob, os, sys
tag, wl = (sys.argv + ["r02", "dragon"])[1:3]
acc = collections.defaultdict(list)
for d in glob.glob(f"gpurun_out/{tag}_{wl}_pmcx_*"):
    files = sorted(glob.glob(f"{d}/*/*_counter_collection.csv"), key=os.path.getmtime)
    for r in csv.DictReader(open(files[-1])) if files else []:
        if r["Kernel_Name"].startswith("rayhs_trace") and "count" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(m):
    print(f"{k:32s} {m[k]:16.0f}")
def ratio(a, b, label):
    if a in m and b in m and m[b]:
        print(f"{label:60s} {m[a] / m[b]:10.3f}")
print()
ratio("SQ_INST_LEVEL_SMEM", "SQ_INSTS_SMEM", "scalar loads: mean latency (cycles in flight per instruction)")
ratio("SQ_INST_LEVEL_LDS", "SQ_INSTS_LDS", "LDS: mean latency")
ratio("SQ_INST_LEVEL_VMEM", "SQ_INSTS_VMEM_RD", "vector memory: mean latency (level / reads)")
ratio("SQ_IFETCH_LEVEL", "SQ_IFETCH", "instruction fetch: mean latency")
ratio("SQ_INST_LEVEL_SMEM", "SQ_WAVE_CYCLES", "scalar loads in flight per wave-cycle")
ratio("SQ_INST_LEVEL_LDS", "SQ_WAVE_CYCLES", "LDS in flight per wave-cycle")
ratio("SQ_INST_LEVEL_VMEM", "SQ_WAVE_CYCLES", "vector memory in flight per wave-cycle")
ratio("SQ_IFETCH_LEVEL", "SQ_WAVE_CYCLES", "instruction fetches in flight per wave-cycle")
ratio("SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES", "waiting on LDS per wave-cycle")
ratio("SQ_INST_CYCLES_SALU", "SQ_WAVE_CYCLES", "SALU cycles per wave-cycle")
ratio("SQ_ACTIVE_INST_SCA", "SQ_WAVE_CYCLES", "scalar unit active per wave-cycle")
ratio("SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "LDS active per wave-cycle")
ratio("SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES", "VMEM active per wave-cycle")
ratio("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "VALU active per wave-cycle")
ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "LDS bank-conflict cycles / active cycles")
