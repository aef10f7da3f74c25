#!/bin/bash
# rocprofv3 kernel stats of the bench step for several library builds / arithmetics:  r4_prof_ab.sh <arith> lib_a lib_b ...
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
export TMPDIR=/tmp
A=$1; shift
for L in "$@"; do
  export IDE3D_HIP_LIB=$R/ide-3d_amd/$L/libide3d_hip.so
  O=$R/gpurun_out/prof_${L}_$A
  rm -rf $O; mkdir -p $O
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o t -- python $R/bench.py --steps 20 --warmup 5 --conv-arith $A --no-cpu-baseline --no-roofline --no-roofline-extra --no-arith-sweep --no-dropin --no-parity > $O/bench.json 2> $O/err.txt )
  cp $(find $O/p -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
  python scripts/step_breakdown.py $O/p $O/step_breakdown.json > /dev/null 2>&1
  rm -rf $O/p
  echo "== $L $A: $(python -c "import json;print(json.load(open('$O/bench.json'))['value'])")"
  python - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats.csv')))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:28]:
    n=r['Name'].replace('ide3d::','').replace('(anonymous namespace)::','').replace('void ','')
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:9.1f} us  {n[:110]}")
PY
done
