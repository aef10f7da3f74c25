#!/bin/bash
# A/B of the low-resolution block group (csrc/lowres.hip): per-layer path / one launch per phase / persistent launch, same box, same binary.
# Usage (repo root, GPU box): bash scripts/micro/r6_lowres_ab.sh <out dir under gpurun_out>
O=gpurun_out/$1; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline-extra --no-arith-sweep --no-roofline"
for rep in 1 2; do
IDE3D_NO_LOWRES_GROUP=1 $B > $O/ab_nogroup_$rep.json 2>/dev/null
IDE3D_LOWRES_PERSISTENT=0 $B > $O/ab_phases_$rep.json 2>/dev/null
IDE3D_LOWRES_PERSISTENT=1 $B > $O/ab_persistent_$rep.json 2>/dev/null
done
python - <<PY
import json, glob
for f in sorted(glob.glob('$O/ab_*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], 'value', d['value'], 'ms', d['ms_per_step'], 'dropin_b1', d.get('dropin_b1', {}).get('frames_per_s'), 'eager', d.get('dropin_b1', {}).get('eager_launches_frames_per_s'), 'parity', d.get('parity_ok'))
    except Exception as e:
        print(f, 'unreadable', e)
PY
