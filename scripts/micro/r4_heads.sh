#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_conv_arith.py -q -x -k "resident_weight_heads or per_image_heads" 2>&1 | tail -5
echo "== heads resident"; timeout 120 python scripts/kernel_rooflines.py --iters 30 --only "dual head" 2>&1 | grep -E "dual head"
echo "== heads old";      IDE3D_HEAD_NO_RESIDENT=1 timeout 120 python scripts/kernel_rooflines.py --iters 30 --only "dual head" 2>&1 | grep -E "dual head"
echo "== conv @32";       timeout 120 python scripts/kernel_rooflines.py --iters 30 --only "modconv 3x3 512->512 @32" 2>&1 | grep -E "bf16x6|f16x3"
echo "== bench";          timeout 300 python bench.py --steps 20 --warmup 5 --no-dropin 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('by_conv_arithmetic'))"
