#!/bin/bash
# Scalar-data-cache counters of the trace kernel for build / BVH-order variants (development aid, GPU box):
#   tools/sqc_variants.sh <tag> <spec> ...      spec = <build_exp dir>[:ENV=VALUE[,ENV=VALUE...]]
# one rocprofv3 --pmc pass per spec (counters only, nothing else traced) around tools/kbench.py dragon 3
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/${tag}_sqc.log
: > $out
for spec in "$@"; do
  dir=${spec%%:*}
  unset RAYHS_BVH_ORDER
  [ "$spec" != "$dir" ] && for kv in $(echo "${spec#*:}" | tr ',' ' '); do export "$kv"; done
  export RAYHS_LIB=$PWD/build_exp/$dir/librayhs_hip.so
  d=gpurun_out/${tag}_sqc_$(echo $spec | tr ':=,' '___')
  rm -rf $d
  rocprofv3 --pmc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_TC_DATA_READ_REQ --output-format csv -d $d -- python3 tools/kbench.py ${SQC_WORKLOAD:-dragon} 3 > $d.txt 2>&1
  python3 - "$spec" $d >> $out <<'PY'
import csv, glob, sys, collections
spec, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Kernel_Name"].startswith("rayhs_trace") and "count" not in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v[-3:]) / len(v[-3:]) for k, v in acc.items()}   # the three timed dragon frames (the parity frames come first)
if m:
    req = m["SQC_DCACHE_REQ"]
    print(f"[{spec}] requests {req/1e6:.1f} M  hits {m['SQC_DCACHE_HITS']/1e6:.1f} M  true misses {m['SQC_DCACHE_MISSES']/1e6:.2f} M  in-flight duplicates {m['SQC_DCACHE_MISSES_DUPLICATE']/1e6:.2f} M  "
          f"hit rate {1 - (m['SQC_DCACHE_MISSES'] + m['SQC_DCACHE_MISSES_DUPLICATE'])/req:.3f}  L2 read requests {m.get('SQC_TC_DATA_READ_REQ', 0)/1e6:.2f} M")
else:
    print(f"[{spec}] no counter rows")
PY
done
cat $out
