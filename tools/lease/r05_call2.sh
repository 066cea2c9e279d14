#!/bin/bash
# round-5 GPU call 2: overlap with one conv work-group per CU (LDS floor), Infinity-Cache residency probe
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r05c2; mkdir -p $O
timeout 300 python tools/overlap_step_probe.py 4 plain,floor > $O/overlap_floor.txt 2>&1
tail -20 $O/overlap_floor.txt
timeout 120 tools/bin/mall_resident_probe > $O/mall_resident.txt 2>&1
cat $O/mall_resident.txt
