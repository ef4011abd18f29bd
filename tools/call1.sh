set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 300 build_exp/micro/ir > gpurun_out/r05_issue_rates.txt 2>&1; echo "ir rc=$?"
tail -4 gpurun_out/r05_issue_rates.txt
bash tools/gpu_batch.sh r05a tests bench default
