#!/bin/bash
# Rank 0's share of the dragon frame (configs[3]) at 2 / 4 / 8 ranks with 2 / 3 / 4 frames in flight, on ONE GPU (development aid):
# what bench.py's IN_FLIGHT buys per rank count.  tools/overlap_sweep.sh > gpurun_out/<tag>_overlap_sweep.txt
export TMPDIR=/tmp
for world in 1 2 4 8; do
  for depth in 2 3 4; do
    echo "== world $world, $depth frames in flight"
    timeout -k 10 120 python3 tools/overlap_frames.py scenes/dragon_1080p.json 16 $world $depth 2>&1 | grep "ms per frame"
  done
done
