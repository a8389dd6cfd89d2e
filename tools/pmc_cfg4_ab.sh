#!/bin/bash
# PMC comparison of alternative builds on BASELINE config 4 at 16 spp (tools/cfg4_run.py, variant 13).
# Usage: tools/pmc_cfg4_ab.sh <closed|open> <alt name or main>...   -> gpurun_out/pmc_cfg4_<name>_<mode>/
set -u
MODE=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for NAME in "$@"; do
  OUT=gpurun_out/pmc_cfg4_${NAME}_${MODE}
  rm -rf $OUT && mkdir -p $OUT
  if [ "$NAME" = main ]; then unset PT_LIB_ALT; else export PT_LIB_ALT=$NAME; fi
  export PT_TOOL_VARIANT=13
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/a -- python3 tools/cfg4_run.py 16 $MODE 2 > $OUT/a.log 2>&1 && \
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/b -- python3 tools/cfg4_run.py 16 $MODE 2 > $OUT/b.log 2>&1 && \
  rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/c -- python3 tools/cfg4_run.py 16 $MODE 2 > $OUT/c.log 2>&1
  echo "$NAME rc=$?"; tail -1 $OUT/a.log
done
python3 - "$MODE" "$@" <<'PY'
import csv, glob, sys
mode = sys.argv[1]
for name in sys.argv[2:]:
    tot = {}
    for f in glob.glob(f"gpurun_out/pmc_cfg4_{name}_{mode}/*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "pixel_kernel" not in row["Kernel_Name"]: continue
            tot.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    # per launch: rows are per dispatch (x XCD?) -- sum over rows / launches(2)
    print(name, mode, {k: round(sum(v) / 2 / 1e6, 2) for k, v in sorted(tot.items())}, "(millions per launch)")
PY
