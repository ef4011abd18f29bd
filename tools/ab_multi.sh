#!/bin/bash
# alternating A/B/C... of library builds on one workload (GPU box): tools/ab_multi.sh <tag> <workload> <rounds> <dir>...
export TMPDIR=/tmp
tag=$1; wl=$2; n=$3; shift 3
for i in $(seq $n); do for d in "$@"; do
  RAYHS_LIB=$PWD/build_exp/$d/librayhs_hip.so timeout -k 10 200 python tools/kbench.py $wl 15 2>&1 | tail -1 | sed 's/alg_GB.*//' | tee -a gpurun_out/${tag}_ab.log
done; done
