#!/bin/bash
# rocprofv3 stats + VALU counters for BASELINE config 4 at 16 spp (tools/cfg4_run.py).  Usage: tools/profile_cfg4.sh <tag> [closed|open]
# (PT_TOOL_VARIANT=<n> in the environment selects a kernel variant)
set -u
TAG=$1
MODE=${2:-closed}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT   # rocprofv3 names its files by pid: leftovers of an earlier run would be averaged in
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 tools/cfg4_run.py 16 $MODE 2 > $OUT/stats.log 2>&1 && \
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 tools/cfg4_run.py 16 $MODE 2 > $OUT/pmc_sq.log 2>&1 && \
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 tools/cfg4_run.py 16 $MODE 2 > $OUT/pmc_sq2.log 2>&1
echo "profile_cfg4.sh rc=$?"
cat $OUT/stats.log | tail -2
