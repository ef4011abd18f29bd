#!/bin/bash
# (round 5: the RAYHS_* knobs this recipe sets exist only in -DRAYHS_DEV_KNOBS builds: build its libraries with tools/build_exp.sh <name> [flags], and use build_exp/cur -- tools/build_exp.sh cur -- where it says "default lib")
# experiment (GPU box): five workgroups per CU against four, same BVH (depth-limited so that its stack fits 32 KiB of LDS)
# usage: tools/exp_w5.sh <tag> <lib dir under build_exp> ...
set -o pipefail
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out
out=gpurun_out/${tag}_w5.log
: > $out
run() { # label, env..., lib
  label=$1; shift
  for wl in dragon dragon_low cornell; do
    echo "== $label $wl" >> $out
    env "$@" RAYHS_DEBUG_COUNTERS=1 timeout -k 10 200 python tools/kbench.py $wl 9 2>&1 | grep -E "rayhs launch|kernel_ms" | tail -2 >> $out
  done
}
run "default lib, default tree" X=1
run "default lib, depth-limited tree, 12 ctx slots, cap 15" RAYHS_BVH_DEPTH_LIMIT=16 RAYHS_STACK_CAP=15 RAYHS_TERM_IN_WORKSPACE=1
for d in "$@"; do
  run "$d, depth-limited tree, 12 ctx slots, cap 15" RAYHS_BVH_DEPTH_LIMIT=16 RAYHS_STACK_CAP=15 RAYHS_TERM_IN_WORKSPACE=1 RAYHS_LIB=$PWD/build_exp/$d/librayhs_hip.so
  run "$d, same but held to 4 blocks/CU" RAYHS_BVH_DEPTH_LIMIT=16 RAYHS_STACK_CAP=15 RAYHS_TERM_IN_WORKSPACE=1 RAYHS_BLOCKS_PER_CU=4 RAYHS_LIB=$PWD/build_exp/$d/librayhs_hip.so
done
cat $out
