#!/bin/bash
# round-6 GPU call 19: final tree (split-K for wo too) -- whole suite, profiles with the final Dual-AR sources, default bench
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
O=gpurun_out/r06c19; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout -k 10 1500 bash tools/make_profiles.sh r06 step prefill pmc gemm pmcsq > $O/make_profiles.log 2>&1
tail -2 $O/make_profiles.log
cp gpurun_out/profiles_r06/pmc_traffic.json profiles/pmc_traffic.json
timeout -k 10 900 python bench.py > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['breakdown_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_prefill']['avg_launch_ms'], d['roofline_prefill']['frac'], d['other_configs']['config4_streaming'])"
