#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_conv_arith.py -q -x -k "arithmetics_vs_float64 or strip_plan or epilogue_and_split_k or y_amax" 2>&1 | tail -2
run() { local only="$1"; shift; echo "== [$only] $*"; env "$@" timeout 120 python scripts/kernel_rooflines.py --iters 20 --only "$only" 2>&1 | grep -E "bf16x6|f16x3" | cut -c1-120; }
run "transposed 3x3 512->512 in@16" A=1
run "transposed 3x3 512->512 in@16" IDE3D_MODCONV_NO_STRIP_SPLIT=1
for rep in 1 2; do for E in A=1 IDE3D_MODCONV_NO_STRIP_SPLIT=1; do
  echo -n "$E: "; env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dropin 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], {k:v['frames_per_s'] for k,v in d['by_conv_arithmetic'].items()})"
done; done
