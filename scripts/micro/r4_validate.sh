#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_conv_arith.py -q -x -k "arithmetics_vs_float64 or strip_plan or epilogue_and_split_k or y_amax" 2>&1 | tail -2
for rep in 1 2; do for E in A=1 IDE3D_MODCONV_SP_MINCIN=33; do
  echo -n "$E: "; env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dropin 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['frames_per_s'] for k,v in d['by_conv_arithmetic'].items()})"
done; done
