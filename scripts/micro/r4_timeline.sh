#!/bin/bash
# step timeline of the default bench under rocprofv3 (one timed step, every launch):  r4_timeline.sh [env ...]
cd ${GRAFT_REPO_ROOT:-.}
R=$PWD; export TMPDIR=/tmp
O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
( cd /tmp && env "$@" timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/p -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-extra --no-arith-sweep > $O/out.txt 2>&1 )
python scripts/step_timeline.py $O/p > $O/step_timeline.txt 2>/dev/null
rm -rf $O/p
head -45 $O/step_timeline.txt | cut -c1-150
