#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "style" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_bench_config.py -m gpu -x -q 2>&1 | tail -3
run() { timeout 300 python bench.py --steps 30 --warmup 5 --no-arith-sweep --no-roofline-extra --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['value'],1), d['parity_ok'], d['parity']['max_rel_err'], d['native_launches'])"; }
for rep in 1 2; do
echo "== batched styles"; run
echo "== per-layer styles (IDE3D_NO_STYLE_BATCH=1)"; IDE3D_NO_STYLE_BATCH=1 run
done
