#!/bin/bash
# (round 5: the RAYHS_* knobs this recipe sets exist only in -DRAYHS_DEV_KNOBS builds: build its libraries with tools/build_exp.sh <name> [flags], and use build_exp/cur -- tools/build_exp.sh cur -- where it says "default lib")
# experiment (GPU box): a 96-VGPR build at 5 workgroups per CU against the shipped one; what each ingredient costs
# usage: tools/exp_w5b.sh <tag> <shipped lib dir> <96-VGPR lib dir>
export TMPDIR=/tmp
tag=$1; base=$2; w5=$3
out=gpurun_out/${tag}_w5b.log
: > $out
run() { label=$1; shift
  for wl in dragon dragon_low; do
    echo "== $label $wl" >> $out
    env "$@" RAYHS_DEBUG_COUNTERS=1 timeout -k 10 200 python tools/kbench.py $wl 9 2>&1 | grep -E "rayhs launch|kernel_ms" | tail -2 | sed 's/alg_GB.*//' >> $out
  done
}
T="RAYHS_TERM_IN_WORKSPACE=1 RAYHS_BVH_DEPTH_LIMIT=13 RAYHS_STACK_CAP=13"
run "A shipped" RAYHS_LIB=$PWD/build_exp/$base/librayhs_hip.so
run "B shipped, light term in the workspace" RAYHS_TERM_IN_WORKSPACE=1 RAYHS_LIB=$PWD/build_exp/$base/librayhs_hip.so
run "C 96 VGPRs at 4 per CU" RAYHS_BLOCKS_PER_CU=4 RAYHS_LIB=$PWD/build_exp/$w5/librayhs_hip.so
run "E shipped, depth-limited tree + 13-entry stack + term in workspace (4 per CU)" $T RAYHS_LIB=$PWD/build_exp/$base/librayhs_hip.so
run "D4 96 VGPRs, same knobs, held to 4 per CU" $T RAYHS_BLOCKS_PER_CU=4 RAYHS_LIB=$PWD/build_exp/$w5/librayhs_hip.so
run "D5 96 VGPRs, same knobs, 5 per CU" $T RAYHS_LIB=$PWD/build_exp/$w5/librayhs_hip.so
run "A shipped (again)" RAYHS_LIB=$PWD/build_exp/$base/librayhs_hip.so
cat $out
