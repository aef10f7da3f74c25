#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_conv_arith.py -q -x -k "per_image_heads or resident_weight" 2>&1 | tail -2
bash scripts/micro/r4_timeline.sh | grep -E "head_small|step span"
for rep in 1 2; do for E in A=1 IDE3D_HEAD_NO_SMALL=1; do
  echo -n "$E: "; env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-dropin 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done; done
