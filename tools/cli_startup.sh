#!/bin/bash
# where a one-shot `rayhs -oX scene.json` spends its time (GPU box): tools/cli_startup.sh <tag> [reps]
export TMPDIR=/tmp
tag=${1:-r04}; reps=${2:-7}
out=gpurun_out/${tag}_cli_startup.txt
: > $out
for args in "--spp=16 scenes/dragon_1080p.json" "--spp=1 scenes/cornell_1080p_d0.json"; do
  for i in $(seq $reps); do
    echo "rayhs --timing -o/tmp/x.ppm $args" >> $out
    RAYHS_DEBUG_TIMING=1 rayhs_amd/bin/rayhs --timing -o/tmp/x.ppm $args 2>> $out > /dev/null
  done
done
cat $out
