#!/bin/bash
# one GPU-box visit (development aid): tools/gpu_batch.sh <tag> [tests] [bench] [kbench libs...]
set -o pipefail
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out
for what in "$@"; do
  case $what in
    tests) timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/${tag}_tests.log; tail -4 gpurun_out/${tag}_tests.log;;
    fasttests) timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not 4k and not random and not whole" > gpurun_out/${tag}_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/${tag}_tests.log; tail -4 gpurun_out/${tag}_tests.log;;
    bench) timeout -k 10 300 python bench.py --steps 20 --warmup 3 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$?"; cat gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err;;
    ptime) RAYHS_LIB=$PWD/build_exp/binpt/librayhs_hip.so RAYHS_DEBUG_COUNTERS=1 timeout -k 10 200 python tools/kbench.py dragon 3 > gpurun_out/${tag}_ptime.log 2>&1; grep "rayhs adv" gpurun_out/${tag}_ptime.log | tail -1;;
    ploops) RAYHS_LIB=$PWD/build_exp/binpl/librayhs_hip.so RAYHS_DEBUG_COUNTERS=1 timeout -k 10 200 python tools/kbench.py dragon 3 > gpurun_out/${tag}_ploops.log 2>&1; grep "rayhs adv" gpurun_out/${tag}_ploops.log | tail -1;;
    balance) timeout -k 10 800 python tools/stripe_balance.py dragon dragon_low dragon4k > gpurun_out/${tag}_stripe_balance.jsonl 2> gpurun_out/${tag}_stripe_balance.err; cat gpurun_out/${tag}_stripe_balance.jsonl; tail -3 gpurun_out/${tag}_stripe_balance.err;;
    firstframe) timeout -k 10 600 python tools/first_frame.py > gpurun_out/${tag}_first_frame.jsonl 2> gpurun_out/${tag}_first_frame.err; cat gpurun_out/${tag}_first_frame.jsonl; tail -3 gpurun_out/${tag}_first_frame.err;;
    default) for wl in dragon dragon_low cornell dragon4k; do timeout -k 10 200 python tools/kbench.py $wl 9 2>&1 | tail -1 | tee -a gpurun_out/${tag}_kbench.log; done;;
    *) for wl in dragon dragon_low cornell; do RAYHS_LIB=$PWD/build_exp/$what/librayhs_hip.so timeout -k 10 200 python tools/kbench.py $wl 9 2>&1 | tail -1 | tee -a gpurun_out/${tag}_kbench.log; done;;
  esac
done
