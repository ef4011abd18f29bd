#!/bin/bash
# Extra PMC passes on the trace kernel (GPU box): memory-pipe latencies, instruction fetch, instruction mix.
#   gpurun -- 'bash tools/pmc_extra.sh r02 dragon'   ->  gpurun_out/<tag>_<wl>_pmcx_*/ ; python tools/pmc_extra_summary.py <tag> <wl>
set -o pipefail
export TMPDIR=/tmp
tag=${1:-r04}; wl=${2:-dragon}
pre=gpurun_out/${tag}_${wl}_pmcx
rm -rf ${pre}_*
pmc() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d ${pre}_$name -- python3 tools/kbench.py $wl 4 > /dev/null 2>&1 || exit 3; }
pmc lat SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pmc vmem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC
pmc ifetch SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64
pmc mix SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES
echo done
