#!/bin/bash
# round-5 GPU call 4: sampler with bucket-count candidate selection + register-resident gathers: bit-exactness, stage timing, frame A/B
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r05c4; mkdir -p $O
timeout 600 python -m pytest tests/test_dualar_gpu.py -m gpu -q -x -k "sampler or sampled or full_sequence or ras or batch_equals or teacher_forced" > $O/pytest_sampler.log 2>&1
tail -5 $O/pytest_sampler.log
timeout 300 python -m pytest tests/test_s2_parity_gpu.py -m gpu -q -x -k "full_sequence or ragged or merged or rows_are" > $O/pytest_s2.log 2>&1
tail -5 $O/pytest_s2.log
echo "== short path (default)" > $O/sampler_bench.txt; timeout 120 tools/bin/sampler_bench >> $O/sampler_bench.txt 2>&1
echo "== FMI_SAMPLE_DESCENT=1" >> $O/sampler_bench.txt; FMI_SAMPLE_DESCENT=1 timeout 120 tools/bin/sampler_bench >> $O/sampler_bench.txt 2>&1
cat $O/sampler_bench.txt
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
timeout 300 $B > $O/bench_short.log 2>&1
FMI_SAMPLE_DESCENT=1 timeout 300 $B > $O/bench_descent.log 2>&1
timeout 300 $B > $O/bench_short2.log 2>&1
for f in $O/bench_*.log; do echo -n "$f: "; grep -h '"metric"' $f | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[l.index('{'):]); print(d['value'], d['ms_per_step'], d['breakdown_ms'])
"; done
