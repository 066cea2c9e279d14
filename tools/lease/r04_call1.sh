#!/bin/bash
# round-4 GPU call 1: new decode GEMV (held fragments, epilogue prefetch) -- micro A/B, frame A/B, kernel profile, GPU tests
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r04c1; mkdir -p $O
GEMV_CHECK=1 timeout 300 tools/bin/gemv_bench > $O/gemv_default.txt 2>&1
FMI_GEMV_NOHOLD=1 timeout 300 tools/bin/gemv_bench > $O/gemv_nohold.txt 2>&1
FMI_GEMV_LATE_EPI=1 timeout 300 tools/bin/gemv_bench > $O/gemv_lateepi.txt 2>&1
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
timeout 300 $B > $O/bench_default.log 2>&1
FMI_GEMV_NOHOLD=1 timeout 300 $B > $O/bench_nohold.log 2>&1
FMI_GEMV_LATE_EPI=1 timeout 300 $B > $O/bench_lateepi.log 2>&1
FMI_GEMV_NOHOLD=1 FMI_GEMV_LATE_EPI=1 timeout 300 $B > $O/bench_r03like.log 2>&1
HIP_FORCE_DEV_KERNARG=1 timeout 300 $B > $O/bench_devkernarg.log 2>&1
HIP_FORCE_DEV_KERNARG=0 timeout 300 $B > $O/bench_hostkernarg.log 2>&1
grep -h '"metric"' $O/bench_*.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[l.index('{'):]); print(d['value'], d['ms_per_step'], d.get('roofline', {}).get('frac'), d.get('breakdown_ms'))
" > $O/bench_summary.txt 2>&1
timeout 600 bash tools/make_profiles.sh r04a step > $O/profiles.log 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
tail -5 $O/pytest.log
cat $O/bench_summary.txt
