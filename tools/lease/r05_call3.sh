#!/bin/bash
# round-5 GPU call 3: Infinity-Cache-resident fast weights -- GEMV warm/cold probe and the frame A/B
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r05c3; mkdir -p $O
timeout 120 tools/bin/mall_probe > $O/mall_probe.txt 2>&1
cat $O/mall_probe.txt
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
for m in 0 7 3 1 15 23; do
  FMI_RESIDENT=$m timeout 300 $B > $O/bench_res$m.log 2>&1
done
FMI_RESIDENT=0 timeout 300 $B > $O/bench_res0b.log 2>&1
FMI_RESIDENT=7 timeout 300 $B > $O/bench_res7b.log 2>&1
for f in $O/bench_res*.log; do echo -n "$f: "; grep -h '"metric"' $f | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[l.index('{'):]); print(d['value'], d['ms_per_step'], d['breakdown_ms'])
"; done
