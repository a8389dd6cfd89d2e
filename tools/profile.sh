#!/bin/bash
# Profile bench.py on the GPU box: kernel-trace stats, then PMC passes (each in its own run,
# never combined with tracing).  Usage: tools/profile.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{stats,pmc_sq,pmc_write,pmc_fetch}; summarise with
# tools/summarise_profile.py and copy the summary into profiles/.
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT   # rocprofv3 names its files by pid: leftovers of an earlier run would be averaged in
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --no-alt-rng --no-other-configs $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py $ARGS > $OUT/stats.log 2>&1 && \
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/pmc_sq -- python3 bench.py $ARGS > $OUT/pmc_sq.log 2>&1 && \
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- python3 bench.py $ARGS > $OUT/pmc_sq2.log 2>&1 && \
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/pmc_mix1 -- python3 bench.py $ARGS > $OUT/pmc_mix1.log 2>&1 && \
rocprofv3 --pmc SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH SQ_INSTS_VALU SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_SMEM --output-format csv -d $OUT/pmc_mix2 -- python3 bench.py $ARGS > $OUT/pmc_mix2.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/pmc_write.log 2>&1 && \
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/pmc_fetch.log 2>&1
echo "profile.sh rc=$?"
grep -h '"metric"' $OUT/stats.log | cut -c1-200
