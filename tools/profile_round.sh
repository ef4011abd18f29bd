#!/bin/bash
# Round evidence on the GPU box:  gpurun -- 'bash tools/profile_round.sh r01'
# then, back home:                python tools/summarize_profiles.py r01 dragon
# One bench line, the rocprofv3 --kernel-trace --stats summary of the same command, and four PMC passes
# (counters in their own runs, no other trace domain -- see MI355X_MICROARCH.md, HBM/rocprofv3 section).
set -o pipefail
tag=${1:-r01}
out=gpurun_out
export TMPDIR=/tmp
rm -rf $out/${tag}_trace $out/${tag}_pmc_* $out/bench_${tag}.* $out/${tag}_trace_bench.json
mkdir -p $out
timeout -k 10 400 python3 bench.py --steps 30 --warmup 5 > $out/bench_${tag}.json 2> $out/bench_${tag}.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace -- python3 bench.py --steps 30 --warmup 5 --cpu-seconds 0 > $out/${tag}_trace_bench.json 2>/dev/null || exit 2
pmc() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $out/${tag}_pmc_$name -- python3 bench.py --steps 3 --warmup 1 --cpu-seconds 0 > /dev/null 2>&1; }
pmc fetch FETCH_SIZE || exit 3
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 4
pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU || exit 5
pmc sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_FLAT GRBM_GUI_ACTIVE || exit 6
cat $out/bench_${tag}.json
