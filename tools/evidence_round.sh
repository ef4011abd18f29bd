#!/bin/bash
# Round evidence besides the per-workload profiles (GPU box):  gpurun -- 'bash tools/evidence_round.sh r02'
# first-frame vs steady state, stripe balance, static estimate vs measured tile costs, phase times and section
# utilisation of the development builds, end-to-end CLI timings.  Outputs under gpurun_out/<tag>_*; copy to profiles/<tag>/.
set -o pipefail
tag=${1:-r04}
out=gpurun_out
export TMPDIR=/tmp
mkdir -p $out
python3 tools/first_frame.py > $out/${tag}_first_frame.jsonl 2> $out/${tag}_first_frame.err
python3 tools/stripe_balance.py dragon dragon_low dragon4k > $out/${tag}_stripe_balance.jsonl 2> $out/${tag}_stripe_balance.err
RAYHS_LIB=$PWD/build_exp/cur/librayhs_hip.so python3 tools/tile_costs.py > $out/${tag}_static_costs.md 2> $out/${tag}_static_costs.err   # (the chunk-time dump exists only in -DRAYHS_DEV_KNOBS builds)
if [ -f build_exp/binpt/librayhs_hip.so ]; then
  for wl in dragon dragon4k; do
    RAYHS_LIB=$PWD/build_exp/binpt/librayhs_hip.so RAYHS_DEBUG_COUNTERS=1 python3 tools/kbench.py $wl 3 2>&1 | python3 tools/phase_times.py > $out/${tag}_phase_times_$wl.txt
  done
fi
if [ -f build_exp/binpl/librayhs_hip.so ]; then
  RAYHS_LIB=$PWD/build_exp/binpl/librayhs_hip.so RAYHS_DEBUG_COUNTERS=1 python3 tools/kbench.py dragon 2 2>&1 | grep -E "^rayhs (adv|sections|traversal|packet|perlane)" | tail -5 > $out/${tag}_loop_counters_dragon.txt
fi
: > $out/${tag}_cli_timing.txt
for args in "--spp=16 scenes/dragon_1080p.json" "--spp=4 scenes/dragon_low_1080p.json" "--spp=1 scenes/cornell_1080p_d0.json" "--spp=16 --devices=0,0 scenes/dragon_1080p.json"; do
  for i in 1 2 3; do
    echo "rayhs --timing -o/tmp/x.ppm $args" >> $out/${tag}_cli_timing.txt
    rayhs_amd/bin/rayhs --timing -o/tmp/x.ppm $args 2>> $out/${tag}_cli_timing.txt > /dev/null
  done
done
python3 tools/time_scene_create.py > $out/${tag}_scene_create.txt 2>&1
python3 tools/overlap_frames.py scenes/dragon_1080p.json 16 8 2 > $out/${tag}_overlap_frames.txt 2>&1
tail -3 $out/${tag}_first_frame.jsonl; tail -3 $out/${tag}_cli_timing.txt
