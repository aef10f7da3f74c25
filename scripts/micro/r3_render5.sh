#!/bin/bash
cd /root/repo
for a in 1 16; do
  echo "== IDE3D_CONV_ARITH=$a"
  IDE3D_CONV_ARITH=$a timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_bench_config.py -m gpu -x -q 2>&1 | tail -3
done
timeout 300 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/bench_render5.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_render5.json'))
print(d['value'], d['value_fp32_exact'], d['parity'], {k:v['frames_per_s'] for k,v in d['by_conv_arithmetic'].items()})
PY
