#!/bin/bash
# round-4 GPU call 2: chunked-loop fix, rolling ring depths, wqkv ring depth -- micro A/B, frame A/B, tests
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r04c2; mkdir -p $O
GEMV_CHECK=1 timeout 300 tools/bin/gemv_bench > $O/gemv_default.txt 2>&1
for d in 2 3 4; do FMI_GEMV_ROLL=$d timeout 300 tools/bin/gemv_bench 2>&1 | grep -A4 "^wo\|^w2" > $O/gemv_roll$d.txt; done
for d in 2 3; do FMI_GEMV_QKV_UNR=$d timeout 300 tools/bin/gemv_bench 2>&1 | grep -A4 "^wqkv" > $O/gemv_qkv$d.txt; done
FMI_GEMV_LATE_EPI=1 timeout 300 tools/bin/gemv_bench 2>&1 | grep -A4 "^wo\|^w2" > $O/gemv_lateepi.txt
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/bench_$name.log 2>&1; grep -h '"metric"' $O/bench_$name.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[l.index('{'):]); print('$name', d['value'], d['ms_per_step'], d['breakdown_ms'])
" >> $O/bench_summary.txt 2>&1; }
run default A=1
run roll2 FMI_GEMV_ROLL=2
run roll3 FMI_GEMV_ROLL=3
run roll4 FMI_GEMV_ROLL=4
run qkv2 FMI_GEMV_QKV_UNR=2
run qkv3 FMI_GEMV_QKV_UNR=3
run lateepi FMI_GEMV_LATE_EPI=1
run default2 A=1
cat $O/bench_summary.txt
timeout 600 bash tools/make_profiles.sh r04b step > $O/profiles.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -15 $O/pytest.log
