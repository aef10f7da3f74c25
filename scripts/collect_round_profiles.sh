#!/bin/bash
# Everything profiles/roundN holds, produced in ONE gpurun call (run from the repo root on the GPU box):
#   scripts/collect_round_profiles.sh <out-subdir under gpurun_out>
# rocprofv3 counter passes are separate from the trace pass (--kernel-trace only next to --pmc).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
cp gpurun_out/bench_full.json $O/bench_default_full.json 2>/dev/null
for A in fp32 f16x3; do python bench.py --steps 20 --warmup 5 --conv-arith $A --no-arith-sweep --no-roofline-extra --no-cpu-baseline --no-live-pmc > $O/bench_$A.json 2>/dev/null; done
python scripts/kernel_rooflines.py --iters 20 --json $O/kernel_rooflines.json > $O/kernel_rooflines.txt 2>&1
# BASELINE configs 3, 4 (one GPU), 5 and the encoder loop, in the library-default arithmetic (bf16x6 since round 4)
python scripts/bench_video.py 120 > $O/bench_video.json 2> $O/bench_video.err
python scripts/bench_config4.py > $O/bench_config4_1gpu.json 2> $O/bench_config4.err
python scripts/bench_shapes.py > $O/bench_shapes.json 2> $O/bench_shapes.err
# the reference's gen_images.py loop verbatim (batch 1): G.synthesis replaying its own captured hipGraph, then the same loop with eager launches
python scripts/bench_dropin.py 90 > $O/bench_dropin.json 2> $O/bench_dropin.err
python scripts/bench_dropin.py 90 --eager >> $O/bench_dropin.json 2>> $O/bench_dropin.err
python scripts/bench_encoder.py > $O/bench_encoder.json 2> $O/bench_encoder.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o t -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-extra --no-arith-sweep --no-live-pmc > $O/bench_under_rocprof.json 2> $O/prof_bench.err )
python scripts/step_breakdown.py $O/prof_bench $O/bench_step_breakdown.json > /dev/null 2>&1
python scripts/step_timeline.py $O/prof_bench > $O/step_timeline.txt 2>/dev/null
cp $(find $O/prof_bench -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv 2>/dev/null
rm -rf $O/prof_bench
bash scripts/pmc_kernels.sh $1/pmc_kernels > $O/pmc_kernels.log 2>&1
cp $O/pmc_kernels/kernel_pmc.json $O/kernel_pmc.json 2>/dev/null
python scripts/cut_pmc.py $O/kernel_pmc.json $O > /dev/null 2>&1
rm -rf $O/pmc_kernels/p? $O/pmc_kernels.p?.log
# round 6: the low-resolution block group — per-layer launches / one launch per phase / one persistent launch, same box, same binary — as bench
# lines and as kernel timelines of one batch-1 image and one batch-4 step; the GPU's clocks / power (idle here; under load: `gpu_state` in the bench line)
bash scripts/micro/r6_lowres_ab.sh $1/lowres_ab > $O/lowres_ab.txt 2>&1
MODES="nogroup phases persistent" bash scripts/micro/r6_lowres_trace.sh $1/lowres_trace > /dev/null 2>&1
rocm-smi --showclocks --showpower --showproductname > $O/rocm_smi_idle.txt 2>&1
ls -la $O
