#!/bin/bash
# PC sampling of the trace kernel (development aid, GPU box): tools/pc_sample.sh <tag> [method] [interval] [workload]
# rocprofv3's beta PC sampling, nothing else traced.  Output: gpurun_out/<tag>_pcs/ (csv) + a per-instruction histogram.
export TMPDIR=/tmp
tag=$1; method=${2:-stochastic}; interval=${3:-65536}; wl=${4:-dragon}
unit=cycles; [ "$method" = host_trap ] && unit=time
d=gpurun_out/${tag}_pcs
rm -rf $d
export RAYHS_LIB=$PWD/build_exp/${PCS_LIB:-cur}/librayhs_hip.so
timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $unit --pc-sampling-method $method --pc-sampling-interval $interval \
  --output-format csv -d $d -- python3 tools/kbench.py $wl 30 > ${d}.txt 2>&1
echo "pc sampling ($method, $unit, $interval) rc=$?"
tail -3 ${d}.txt
ls -la $d/*/ 2>/dev/null | head
for f in $d/*/*pc_sampling*.csv; do [ -f "$f" ] && { head -3 "$f"; wc -l "$f"; }; done
