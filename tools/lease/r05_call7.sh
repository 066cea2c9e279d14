#!/bin/bash
# round-5 GPU call 7: sampler stages and the frame A/B (bucket counts vs radix descent) from the FINAL binary
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r05c7; mkdir -p $O
tools/bin/sampler_bench > /dev/null 2>&1   # (clock warm-up: the first run of a fresh box reads ~1 us high)
{ echo "== bucket-count candidate selection (shipped)"; tools/bin/sampler_bench; echo "== FMI_SAMPLE_DESCENT=1 (round-4 selection: radix descent in every wave)"; FMI_SAMPLE_DESCENT=1 tools/bin/sampler_bench; } > $O/sampler_stages.txt 2>&1
cat $O/sampler_stages.txt
B="python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras"
for i in 1 2; do
  timeout 300 $B > $O/bench_short$i.log 2>&1
  FMI_SAMPLE_DESCENT=1 timeout 300 $B > $O/bench_descent$i.log 2>&1
done
for f in $O/bench_*.log; do echo -n "$(basename $f): "; grep -h '"metric"' $f | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[l.index('{'):]); print(d['value'], d['ms_per_step'], d['breakdown_ms']['decode_frame_avg'])
"; done
