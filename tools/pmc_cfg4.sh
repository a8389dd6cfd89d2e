#!/bin/bash
# Counter passes for BASELINE config 4 at its own size (tools/cfg4_run.py <spp> <closed|open>), each set in its own run, never with tracing.
# Usage: tools/pmc_cfg4.sh <tag> [closed|open] [spp=256]     -> gpurun_out/prof_<tag>/{stats,pmc_*}
set -u
TAG=$1
MODE=${2:-closed}
SPP=${3:-256}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
run() { local name=$1; shift; rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 tools/cfg4_run.py $SPP $MODE 2 > $OUT/$name.log 2>&1; }
run stats --kernel-trace --stats && \
run pmc_sq --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY && \
run pmc_sq2 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH GRBM_GUI_ACTIVE && \
run pmc_sq3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM && \
run pmc_write --pmc WRITE_SIZE && \
run pmc_fetch --pmc FETCH_SIZE
echo "pmc_cfg4.sh rc=$?"
tail -2 $OUT/stats.log
