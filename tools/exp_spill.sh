#!/bin/bash
# experiment (GPU box): what a register-limited build costs at the SAME occupancy (4 workgroups per CU)
# usage: tools/exp_spill.sh <tag> <lib dir under build_exp> ...
export TMPDIR=/tmp
tag=$1; shift
out=gpurun_out/${tag}_spill.log
: > $out
for d in "$@"; do
  for wl in dragon dragon_low; do
    echo "== $d $wl (held to 4 blocks/CU)" >> $out
    RAYHS_BLOCKS_PER_CU=4 RAYHS_DEBUG_COUNTERS=1 RAYHS_LIB=$PWD/build_exp/$d/librayhs_hip.so timeout -k 10 200 python tools/kbench.py $wl 9 2>&1 | grep -E "rayhs launch|kernel_ms" | tail -2 >> $out
  done
done
cat $out
