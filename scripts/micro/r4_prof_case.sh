#!/bin/bash
# rocprofv3 kernel stats of one kernel_rooflines case:  r4_prof_case.sh "<--only substring>" [env ...]
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD; export TMPDIR=/tmp
ONLY=$1; shift
O=$R/gpurun_out/prof_case; rm -rf $O; mkdir -p $O
( cd /tmp && env "$@" timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o t -- python $R/scripts/kernel_rooflines.py --iters 10 --eager --only "$ONLY" > $O/out.txt 2>&1 )
python - <<PY
import csv,glob
f=glob.glob('$O/p/**/*kernel_stats.csv', recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:14]:
    n=r['Name'].replace('ide3d::','').replace('(anonymous namespace)::','').replace('void ','')
    print(f"{float(r['AverageNs'])/1e3:9.1f} us avg {int(r['Calls']):5d} calls  {n[:100]}")
PY
