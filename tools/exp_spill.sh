#!/bin/bash
# (round 5: the RAYHS_* knobs this recipe sets exist only in -DRAYHS_DEV_KNOBS builds: build its libraries with tools/build_exp.sh <name> [flags], and use build_exp/cur -- tools/build_exp.sh cur -- where it says "default lib")
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
