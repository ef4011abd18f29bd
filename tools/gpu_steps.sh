#!/bin/bash
# Run a list of GPU steps on one box visit (development aid):  tools/gpu_steps.sh <tag> <<'STEPS' ... one shell command per line ... STEPS
# Every step runs under its own `timeout -k 10 <seconds>` (first word of the line), logs to gpurun_out/<tag>_stepNN.log, and a step that
# TIMES OUT or is KILLED ends the visit (no further GPU step after a hung one); any other failure is reported and the visit goes on.
export TMPDIR=/tmp
tag=$1
mkdir -p gpurun_out
n=0
while IFS= read -r line; do
  [ -z "$line" ] && continue
  n=$((n + 1))
  secs=${line%% *}; cmd=${line#* }
  log=gpurun_out/${tag}_step$(printf %02d $n).log
  echo "=== step $n (${secs}s): $cmd" | tee $log
  timeout -k 10 $secs bash -c "$cmd" >> $log 2>&1
  rc=$?
  echo "=== step $n rc=$rc" | tee -a $log
  tail -n 12 $log
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $n timed out or was killed: ending the visit"; exit 1; fi
done
exit 0
