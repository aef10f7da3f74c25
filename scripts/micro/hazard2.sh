#!/bin/bash
# Round 4: runs scripts/micro/_bin/hazard2 (built from pk_mfma_hazard2.cpp) on this box.  $1 = launches per pair, $2 victims, $3 neighbours.
cd ${GRAFT_REPO_ROOT:-.}
timeout 600 scripts/micro/_bin/hazard2 ${1:-2000} ${2:-BMRS} ${3:-0123456789ab}
