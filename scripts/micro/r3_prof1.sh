export TMPDIR=/tmp
mkdir -p gpurun_out/r3_prof1
cd /tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3_prof1 -o $ARITH --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --blocks 2 --no-cpu-baseline --no-roofline --no-roofline-extra --no-arith-sweep --no-parity --graph 0 --conv-arith $ARITH > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r3_prof1/err.log
cd $GRAFT_REPO_ROOT
find gpurun_out/r3_prof1 -name "*kernel_stats*" | head; f=$(find gpurun_out/r3_prof1 -name "*kernel_stats.csv" | head -1); head -30 "$f" | cut -c1-220
