#!/bin/bash
# kernel timelines of one batch-1 image (bench_dropin.py) and one batch-4 step (bench.py) under the three forms of the low-resolution group
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O
for mode in ${MODES:-phases persistent}; do
  if [ $mode = nogroup ]; then export IDE3D_NO_LOWRES_GROUP=1; else unset IDE3D_NO_LOWRES_GROUP; fi
  if [ $mode = phases ]; then export IDE3D_LOWRES_PERSISTENT=0; else export IDE3D_LOWRES_PERSISTENT=1; fi
  ( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/prof_b1_$mode -o t -- python $R/scripts/bench_dropin.py 30 > $O/dropin_$mode.json 2> $O/dropin_$mode.err )
  python scripts/step_timeline.py $O/prof_b1_$mode > $O/b1_timeline_$mode.txt 2>/dev/null
  ( cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $O/prof_b4_$mode -o t -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline-extra --no-arith-sweep --no-roofline --no-dropin > $O/bench_$mode.json 2> $O/bench_$mode.err )
  python scripts/step_timeline.py $O/prof_b4_$mode > $O/b4_timeline_$mode.txt 2>/dev/null
  rm -rf $O/prof_b1_$mode $O/prof_b4_$mode
done
