#!/bin/bash
# (round 5: the RAYHS_* knobs this recipe sets exist only in -DRAYHS_DEV_KNOBS builds: build its libraries with tools/build_exp.sh <name> [flags], and use build_exp/cur -- tools/build_exp.sh cur -- where it says "default lib")
# experiment (GPU box): is a fifth workgroup per CU resident, and what does it buy?  (-DRAYHS_PROFILE_TIME builds: counters[6] = waves that obtained work)
export TMPDIR=/tmp
tag=$1
out=gpurun_out/${tag}_w5pt.log
: > $out
for wl in dragon dragon_low; do
for knobs in "RAYHS_BVH_LEAF=7 RAYHS_BVH_DEPTH_LIMIT=13 RAYHS_STACK_CAP=13" "RAYHS_BVH_LEAF=7 RAYHS_BVH_DEPTH_LIMIT=13 RAYHS_STACK_CAP=12" "RAYHS_BVH_LEAF=7 RAYHS_BVH_DEPTH_LIMIT=13 RAYHS_STACK_CAP=10"; do
for cfg in "w5pt 5" "w5pt 4"; do
  set -- $cfg
  echo "== $wl $1 at $2 blocks/CU, $knobs" >> $out
  env $knobs RAYHS_TERM_IN_WORKSPACE=1 RAYHS_DEBUG_COUNTERS=1 RAYHS_BLOCKS_PER_CU=$2 RAYHS_LIB=$PWD/build_exp/$1/librayhs_hip.so timeout -k 10 200 python tools/kbench.py $wl 5 > /tmp/o.txt 2>&1
  grep -E "rayhs launch" /tmp/o.txt | tail -1 >> $out
  grep -E "kernel_ms" /tmp/o.txt | tail -1 | sed 's/Mrays.*//' >> $out
  grep "rayhs adv" /tmp/o.txt | tail -1 | sed 's/.*counters\[6..15\]: \([0-9]*\) \([0-9]*\).*/waves with work \1, chunks \2/' >> $out
done; done; done
cat $out
