#!/bin/bash
# A/B of library builds on the dragon frame only, alternating (GPU box): tools/ab_dragon.sh <tag> <dirA> <dirB> [rounds]
export TMPDIR=/tmp
tag=$1; a=$2; b=$3; n=${4:-4}
for i in $(seq $n); do for d in $a $b; do
  RAYHS_LIB=$PWD/build_exp/$d/librayhs_hip.so timeout -k 10 200 python tools/kbench.py dragon 15 2>&1 | tail -1 | sed 's/alg_GB.*//' | tee -a gpurun_out/${tag}_ab.log
done; done
