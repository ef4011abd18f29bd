#!/bin/bash
# Round evidence on the GPU box:   gpurun -- 'bash tools/profile_round.sh r02 dragon'      (one workload per call)
# then, back home:                 python tools/summarize_profiles.py r02 dragon
# Per workload: one bench line, the rocprofv3 --kernel-trace --stats summary of the same command (ONE frame in flight,
# one stream: the --stats rows of every kernel are then its real duration), and four PMC passes (counters in their own
# runs, no other trace domain -- see MI355X_MICROARCH.md, HBM/rocprofv3 section).
set -o pipefail
tag=${1:-r04}
wl=${2:-dragon}
out=gpurun_out
pre=$out/${tag}_${wl}
export TMPDIR=/tmp
export RAYHS_BENCH_IN_FLIGHT=1
# a run that is itself under rocprofv3 must never start a profiler of its own (bench.py's live counter passes are for un-profiled runs on a
# kernel source no committed profile matches -- the normal case while a round is being profiled): summarize_profiles.py derives the roofs from
# the passes below anyway
export RAYHS_BENCH_NO_LIVE_PMC=1
rm -rf ${pre}_trace ${pre}_pmc_* ${pre}_bench.* ${pre}_trace_bench.json
mkdir -p $out
steps=30; [ "$wl" = dragon4k ] && steps=8
python3 -c "import bench; print(bench.kernel_source_sha256())" > ${pre}_kernel_source.sha256   # (over the code of trace_kernel.hip, device_scene.h, short_div.h, abi.cpp: comments stripped)
# (the one un-profiled run of this script may count its instructions itself: its children are the only profiler sessions around it)
RAYHS_BENCH_NO_LIVE_PMC= timeout -k 10 500 python3 bench.py --workload $wl --steps $steps --warmup 3 > ${pre}_bench.json 2> ${pre}_bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d ${pre}_trace -- python3 bench.py --workload $wl --steps $steps --warmup 3 --cpu-seconds 0 --no-4k > ${pre}_trace_bench.json 2>/dev/null || exit 2
pmc() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d ${pre}_pmc_$name -- python3 bench.py --workload $wl --steps 3 --warmup 1 --cpu-seconds 0 --no-4k > /dev/null 2>&1; }
pmc fetch FETCH_SIZE || exit 3
pmc write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 4
pmc sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU || exit 5
pmc sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_FLAT GRBM_GUI_ACTIVE || exit 6
# instruction mix: what the vector pipe is really asked for (fp64 arithmetic runs at half rate, transcendentals at an eighth) and the branches
pmc mix SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES || exit 7
# round 5: what the vector pipe is asked for BESIDES fp64 (fp32 arithmetic occupies a SIMD for 2 cycles per wave64 instruction, integer /
# compare / select / convert for 4: profiles/r05/micro/issue_rates.txt), and the scalar unit's own cycle counters
pmc mix2 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU || echo "pmc pass mix2 failed"
pmc sca SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH GRBM_GUI_ACTIVE || echo "pmc pass sca failed"
# scalar data cache (the packet walk fetches its nodes and triangles through it) and instruction cache; optional passes: a
# counter this rocprofv3 build does not know must not lose the rest of the profile
pmc sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE || echo "pmc pass sqc failed"
pmc sqc2 SQC_DCACHE_REQ_READ_1 SQC_DCACHE_REQ_READ_2 SQC_DCACHE_REQ_READ_4 SQC_DCACHE_REQ_READ_8 SQC_DCACHE_REQ_READ_16 SQC_TC_DATA_READ_REQ SQC_TC_INST_REQ SQC_TC_STALL || echo "pmc pass sqc2 failed"
RAYHS_DEBUG_COUNTERS=1 python3 tools/kbench.py $wl 1 2>&1 | grep "rayhs launch" | tail -1 > ${pre}_launch.txt
cat ${pre}_bench.json
