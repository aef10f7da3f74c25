#!/bin/bash
# rocprofv3 counter passes over every kernel case of scripts/kernel_rooflines.py (run on the GPU box from the repo root):
#   scripts/pmc_kernels.sh <out-subdir under gpurun_out> [--only substr]
# Separate passes, --kernel-trace only (no other trace domain next to --pmc).  FETCH_SIZE and WRITE_SIZE do not fit one pass
# (TCC slots); SQ / GRBM counters share one.  Summary: <out>/kernel_pmc.json (scripts/pmc_summarize.py).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $(dirname $OUT) $OUT
cd /tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python $R/scripts/kernel_rooflines.py --eager --iters 3 "$@" > $OUT.p$i.log 2>&1
done
python $R/scripts/pmc_summarize.py $OUT
