export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ARITH in f16x3 bf16x3; do
  mkdir -p $R/gpurun_out/r3_prof2/$ARITH
  cd /tmp
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3_prof2/$ARITH -o t --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --blocks 2 --no-cpu-baseline --no-roofline --no-roofline-extra --no-arith-sweep --no-parity --graph 0 --conv-arith $ARITH > /dev/null 2> $R/gpurun_out/r3_prof2/$ARITH/err.log
  cd $R
  python scripts/step_breakdown.py gpurun_out/r3_prof2/$ARITH gpurun_out/r3_prof2/$ARITH/breakdown.json > /dev/null 2>&1
  rm -f $(find gpurun_out/r3_prof2/$ARITH -name "*kernel_trace.csv")
done
ls -R gpurun_out/r3_prof2 | head -30
