#!/bin/bash
cd /root/repo
export IDE3D_CONV_ARITH=6 TMPDIR=/tmp
bash scripts/pmc_kernels.sh r3_render_pmc2 --only "$1" > /dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3_render_pmc2/kernel_pmc.json'))
keys=['kernel_cycles','SQ_INSTS_VALU','SQ_INSTS_MFMA','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_ACTIVE_INST_VALU','SQ_WAVE_CYCLES','SQ_WAIT_INST_ANY','SQ_WAIT_ANY','SQ_ACTIVE_INST_ANY','sq_active_inst_valu_frac_of_wave_cycles','mfma_busy_frac','SQ_WAVES']
for k,v in d.items():
    print(k, {q:v.get(q) for q in keys})
PY
