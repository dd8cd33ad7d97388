#!/bin/bash
# Collects rocprofv3 PMC counters for the trace kernels on the GPU box, one
# counter group per pass (TCC slots: FETCH_SIZE and WRITE_SIZE cannot share a
# pass; PMC is never combined with sys/hip/hsa tracing on this pool).
#   usage: tools/pmc_collect.sh <tag> [workload [num [field]]]   (outputs under gpurun_out/pmc_<tag>/)
set -u
TAG=${1:-r01}
WL=${2:-dblgauss_c2}
NUM=${3:-1024}
FLD=${4:-0}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/ab_bench.py --reps 1 --launches 3 --workload $WL --num $NUM --field $FLD"
pass() {   # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o $name -- $CMD > $OUT/$name.log 2>&1
  echo "pass $name rc=$?"
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM
pass sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
# instruction supply and scalar side of the Newton instances (round 5 audit); a pass whose
# counter names this ROCm does not know fails on its own and leaves the others alone
pass sq3 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES
pass sq4 SQ_IFETCH SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pass fetch FETCH_SIZE
pass write WRITE_SIZE
find $OUT -name "*.csv" | head -20
