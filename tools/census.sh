#!/bin/bash
# Loop counters and the per-lane-walk census of the -DRAYHS_PROFILE_LOOPS build (build_exp/binpl: tools/build_exp.sh binpl -DRAYHS_PROFILE_LOOPS),
# phase times of the -DRAYHS_PROFILE_TIME build (build_exp/binpt) -- GPU box:   tools/census.sh <tag>
export TMPDIR=/tmp
tag=$1
for wl in dragon dragon_low; do
  RAYHS_LIB=$PWD/build_exp/binpl/librayhs_hip.so RAYHS_DEBUG_COUNTERS=1 timeout -k 10 200 python3 tools/kbench.py $wl 2 2>&1 | grep -E "^rayhs (adv|sections|traversal|packet|perlane)" | tail -5 > gpurun_out/${tag}_loop_counters_$wl.txt
  cat gpurun_out/${tag}_loop_counters_$wl.txt
  RAYHS_LIB=$PWD/build_exp/binpt/librayhs_hip.so RAYHS_DEBUG_COUNTERS=1 timeout -k 10 200 python3 tools/kbench.py $wl 3 2>&1 | python3 tools/phase_times.py > gpurun_out/${tag}_phase_times_$wl.txt
  cat gpurun_out/${tag}_phase_times_$wl.txt
done
