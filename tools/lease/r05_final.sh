#!/bin/bash
# round-5 final check: the whole GPU suite on the final binary, smoke(), the default bench line (extras + CPU baseline)
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r05final; mkdir -p $O
timeout 1300 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -6 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -4 $O/smoke.log
timeout 900 python bench.py --steps 5 --warmup 1 > $O/bench_default.log 2>&1
grep -h '"metric"' $O/bench_default.log | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l[l.index('{'):]); print(json.dumps({k: d[k] for k in ('value','ms_per_step','breakdown_ms','roofline')}, indent=0)[:1500]); print(json.dumps(d.get('other_configs', {}))[:3000]); print(json.dumps(d.get('cpu_baseline', {}))[:1200]); print(d.get('roofline_prefill')); print(d.get('roofline_codec'))
"
