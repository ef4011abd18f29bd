#!/bin/bash
# Instruction counts of the trace kernel per build variant (development aid, GPU box) -- the DETERMINISTIC metric small diets are judged on:
#   tools/inst_counts.sh <tag> <spec> ...      spec = <build_exp dir>[:ENV=VALUE[,ENV=VALUE...]]     (WORKLOADS="dragon" by default)
# Two rocprofv3 --pmc passes per spec and workload (counters only, nothing else traced) around tools/kbench.py <workload> 3: what the scalar
# unit is asked for (SALU + SMEM + BRANCH), and the vector pipes by cost class -> priced cycles (2 per fp32 add / mul / fma and unclassified,
# 4 per fp64 add / mul / fma / integer / convert, 16 per fp64 rcp / rsq / sqrt: profiles/r05/micro/issue_rates.txt).
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/${tag}_insts.log
: > $out
for spec in "$@"; do
  dir=${spec%%:*}
  [ "$spec" != "$dir" ] && for kv in $(echo "${spec#*:}" | tr ',' ' '); do export "$kv"; done
  if [ "$dir" = default ]; then unset RAYHS_LIB; else export RAYHS_LIB=$PWD/build_exp/$dir/librayhs_hip.so; fi
  for wl in ${WORKLOADS:-dragon}; do
    d=gpurun_out/${tag}_insts_$(echo $spec | tr ':=,' '___')_$wl
    rm -rf ${d}_a ${d}_b
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d ${d}_a -- python3 tools/kbench.py $wl 3 > ${d}_a.txt 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_INT32 --output-format csv -d ${d}_b -- python3 tools/kbench.py $wl 3 > ${d}_b.txt 2>&1
    rocprofv3 --pmc SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY --output-format csv -d ${d}_c -- python3 tools/kbench.py $wl 3 > ${d}_c.txt 2>&1
    python3 - "$spec" $wl ${d} >> $out <<'PY'
import csv, glob, sys, collections
spec, wl, d = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(list)
for part in "abc":
    for f in glob.glob(f"{d}_{part}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("rayhs_trace") and "count" not in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v[-3:]) / len(v[-3:]) for k, v in acc.items()}   # the three timed frames of the workload (the parity frames come first)
if "SQ_INSTS_VALU" in m and "SQ_INSTS_VALU_ADD_F64" in m:
    g = lambda k: m.get(k, 0.0)
    scalar = g("SQ_INSTS_SALU") + g("SQ_INSTS_SMEM") + g("SQ_INSTS_BRANCH")
    f64 = g("SQ_INSTS_VALU_ADD_F64") + g("SQ_INSTS_VALU_MUL_F64") + g("SQ_INSTS_VALU_FMA_F64")
    f32 = g("SQ_INSTS_VALU_ADD_F32") + g("SQ_INSTS_VALU_MUL_F32") + g("SQ_INSTS_VALU_FMA_F32")
    intc = g("SQ_INSTS_VALU_INT32") + g("SQ_INSTS_VALU_INT64") + g("SQ_INSTS_VALU_CVT")
    other = g("SQ_INSTS_VALU") - f64 - f32 - intc - g("SQ_INSTS_VALU_TRANS_F64") - g("SQ_INSTS_VALU_TRANS_F32")
    cyc = 4 * f64 + 2 * f32 + 4 * intc + 16 * g("SQ_INSTS_VALU_TRANS_F64") + 8 * g("SQ_INSTS_VALU_TRANS_F32")
    wait = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else float("nan")
    print(f"[{spec}] {wl}: VALU {g('SQ_INSTS_VALU')/1e6:.2f} M (f64 {f64/1e6:.2f} f32 {f32/1e6:.2f} int/cvt {intc/1e6:.2f} trans64 {g('SQ_INSTS_VALU_TRANS_F64')/1e6:.2f} other {other/1e6:.2f})  "
          f"VALU pipe cycles {(cyc + 2 * other)/1e9:.4f} G (other at 2) / {(cyc + 4 * other)/1e9:.4f} G (other at 4)  scalar-unit {scalar/1e6:.2f} M (SALU {g('SQ_INSTS_SALU')/1e6:.2f} SMEM {g('SQ_INSTS_SMEM')/1e6:.2f} BRANCH {g('SQ_INSTS_BRANCH')/1e6:.2f})  "
          f"LDS {g('SQ_INSTS_LDS')/1e6:.2f} M  VMEM {(g('SQ_INSTS_VMEM_RD') + g('SQ_INSTS_VMEM_WR'))/1e6:.2f} M  s_waitcnt share of wave life {wait:.3f}")
else:
    print(f"[{spec}] {wl}: no counter rows ({sorted(m)})")
PY
  done
done
cat $out
