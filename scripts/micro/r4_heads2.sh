#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_conv_arith.py -q -x -k "resident_weight_heads or per_image_heads" 2>&1 | tail -3
bash scripts/micro/r4_heads_dbg.sh lib_hs0 lib
