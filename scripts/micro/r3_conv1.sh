export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_conv_arith.py -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r3_conv1_tests.log
cat gpurun_out/r3_conv1_tests.log
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-extra > gpurun_out/r3_conv1_bench.json 2> gpurun_out/r3_conv1_bench.err; tail -3 gpurun_out/r3_conv1_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_conv1_bench.json').read().strip().splitlines()[-1])
print('value', d['value'], 'ms', d['ms_per_step'], 'parity', d.get('parity_ok'), d.get('parity',{}).get('max_rel_err'))
for k,v in d.get('by_conv_arithmetic',{}).items(): print(k, round(v['frames_per_s'],1), v.get('parity_ok'), v.get('parity_max_rel_err'))
print('roofline', d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('avg_launch_us'))
PY
