#!/bin/bash
# rocprofv3 PMC passes over the stand-alone convolution timings (run on the GPU box from the repo root):
#   scripts/pmc_modconv.sh <out-subdir>
# Per kernel template: average counter values per dispatch.  Separate passes, --kernel-trace only.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1
cd /tmp
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python $R/scripts/modconv_only.py 2 > /dev/null 2>&1
done
python - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob('$OUT/p*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(d)):
        m = re.search(r'modconv_kernel<[^>]*>', r['Kernel_Name'])
        if m:
            acc[m.group(0) + ' grid=' + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
with open('$OUT/summary.txt', 'w') as f:
    for k, cs in acc.items():
        f.write(k + '\n')
        for c, v in cs.items():
            f.write(f'    {c:28s} {sum(v) / len(v):16.0f}  (n={len(v)})\n')
print(open('$OUT/summary.txt').read())
PY
