#!/bin/bash
# round 4: what the RCCL path costs per step (world = 1 through IDE3D_BENCH_FORCE_DIST): kernel trace of both runs
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD
export TMPDIR=/tmp
for D in 0 1; do
  O=$R/gpurun_out/prof_dist$D; rm -rf $O; mkdir -p $O
  ( cd /tmp && IDE3D_BENCH_FORCE_DIST=$([ $D = 1 ] && echo 1) rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-roofline-extra --no-arith-sweep --no-dropin --no-parity > $O/bench.json 2> $O/err.txt )
  cp $(find $O/p -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
  python scripts/step_timeline.py $O/p > $O/step_timeline.txt 2>/dev/null
  rm -rf $O/p
  echo "== dist=$D: $(python -c "import json;d=json.load(open('$O/bench.json'));print(d['value'], d['ms_per_step'])")"
  python - <<PY
import csv
rows=list(csv.DictReader(open('$O/kernel_stats.csv')))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:40]:
    n=r['Name']
    if 'ide3d' in n and 'frame' not in n: continue
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:9.1f} us  {n[:100]}")
PY
  tail -12 $O/step_timeline.txt
done
