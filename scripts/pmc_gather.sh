#!/bin/bash
# rocprofv3 PMC passes over the stand-alone gather (run on the GPU box from the repo root):
#   scripts/pmc_gather.sh <out-subdir> [flat]
# Counters are collected in separate passes with --kernel-trace only (no other trace domain).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
cd /tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" \
         "TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" \
         "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python $R/scripts/gather_only.py 3 $2 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for d in sorted(glob.glob('$OUT/p*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(d)):
        if 'triplane_sample' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
with open('$OUT/summary.txt', 'w') as f:
    for k, v in acc.items():
        f.write(f'{k:28s} {sum(v) / len(v):16.0f}  (n={len(v)})\n')
print(open('$OUT/summary.txt').read())
PY
