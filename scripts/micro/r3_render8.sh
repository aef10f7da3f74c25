#!/bin/bash
# PMC passes over the fused renderer (split arithmetic) — SQ activity + L1 (TA / TCP) counters
cd /root/repo
export IDE3D_CONV_ARITH=6 TMPDIR=/tmp
bash scripts/pmc_kernels.sh r3_render_pmc --only render_rays > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_render_pmc/kernel_pmc.json'))
for k,v in d.items():
    if 'render' in k: print(k, json.dumps(v))
PY
cd /tmp
timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --kernel-trace --output-format csv -d /root/repo/gpurun_out/r3_render_pmc/l1 -- python /root/repo/scripts/kernel_rooflines.py --eager --iters 3 --only render_rays > /root/repo/gpurun_out/r3_render_pmc/l1.log 2>&1
tail -5 /root/repo/gpurun_out/r3_render_pmc/l1.log
python - <<'PY'
import csv, glob, collections
for f in glob.glob('/root/repo/gpurun_out/r3_render_pmc/l1/**/*counter_collection.csv', recursive=True):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'render_rays' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(k, sum(v)/len(v), len(v))
PY
