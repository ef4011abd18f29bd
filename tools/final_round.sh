#!/bin/bash
# Everything the round's profiles/ are made from, on one GPU box (one workload's profile per call keeps a call under its time limit):
#   gpurun -- 'bash tools/final_round.sh r05 tests'            the -m gpu suite on the shipped library
#   gpurun -- 'bash tools/final_round.sh r05 <workload>'       tools/profile_round.sh for cornell | dragon_low | dragon | dragon4k
#   gpurun -- 'bash tools/final_round.sh r05 evidence'         tools/evidence_round.sh + wave ends + census
export TMPDIR=/tmp
tag=$1; what=$2
case $what in
  tests) timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_gpu_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/${tag}_gpu_tests.log;;
  evidence)
    bash tools/evidence_round.sh $tag
    for wl in dragon dragon_low cornell; do
      RAYHS_WAVE_ENDS_PRODUCTION=1 RAYHS_LIB=$PWD/build_exp/we/librayhs_hip.so timeout -k 10 120 python3 tools/wave_ends.py $wl >> gpurun_out/${tag}_wave_ends.txt 2>&1
    done
    tail -8 gpurun_out/${tag}_wave_ends.txt;;
  *) bash tools/profile_round.sh $tag $what;;
esac
