#!/bin/bash
# bring-up of an experimental library on the GPU box: goldens + timing first (short timeouts, stop at the first failure), then the GPU
# test-suite against it:   tools/bringup_exp.sh <build_exp dir> <tag>
export TMPDIR=/tmp
lib=$1; tag=$2
export RAYHS_LIB=$PWD/build_exp/$lib/librayhs_hip.so
for wl in cornell dragon_low dragon; do
  timeout -k 10 120 python tools/kbench.py $wl 5 > gpurun_out/${tag}_kb_$wl.log 2>&1
  tail -2 gpurun_out/${tag}_kb_$wl.log
  grep -q "parity=OK" gpurun_out/${tag}_kb_$wl.log || { echo "STOP: $wl failed"; exit 1; }
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/${tag}_tests.log
exit $rc
