#!/bin/bash
# A/B of experimental builds on one GPU box (development aid): tools/exp_batch.sh <tag> <spec> ...
#   spec = <build_exp dir>[:ENV=VALUE[,ENV=VALUE...]]   e.g.  base  pre  base:RAYHS_BVH_ORDER=bfs
# every spec is timed on dragon, dragon_low and cornell (tools/kbench.py, best / median of 9), twice round the list
export TMPDIR=/tmp
tag=$1; shift
mkdir -p gpurun_out
out=gpurun_out/${tag}_exp.log
: > $out
for round in 1 2; do
  for spec in "$@"; do
    dir=${spec%%:*}; envs=""
    [ "$spec" != "$dir" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
    for wl in ${WORKLOADS:-dragon dragon_low cornell}; do
      echo -n "[$spec] " >> $out
      env $envs RAYHS_LIB=$PWD/build_exp/$dir/librayhs_hip.so timeout -k 10 200 python tools/kbench.py $wl 9 2>&1 | tail -1 | sed 's/^[^ ]* //; s/alg_GB.*nodes/nodes/' >> $out
    done
  done
done
cat $out
