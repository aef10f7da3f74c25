#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_conv_arith.py -q -x -k "arithmetics_vs_float64 or epilogue_and_split_k or y_amax" 2>&1 | tail -3
for E in A=1 IDE3D_MODCONV_NO_PH32=1; do
  echo "== $E"; env $E timeout 200 python scripts/kernel_rooflines.py --iters 20 --only "modconv 3x3 64->64" 2>&1 | grep -E "bf16x6|f16x3" | cut -c1-120
done
for rep in 1 2; do for E in A=1 IDE3D_MODCONV_NO_PH32=1; do
  echo -n "$E: "; env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dropin 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
