#!/bin/bash
# round-5 GPU call 1: the whole GPU suite (new: rank > 0 path, per-handle overflow flag, merge / batch tap tests, S2
# random-weight page-boundary taps), sampler stage timing, overlapped-step probe
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
export TMPDIR=/tmp
O=gpurun_out/r05c1; mkdir -p $O
timeout 1300 python -m pytest tests -m gpu -q --durations=25 > $O/pytest.log 2>&1
tail -45 $O/pytest.log
timeout 120 tools/bin/sampler_bench > $O/sampler_bench.txt 2>&1
cat $O/sampler_bench.txt
timeout 420 python tools/overlap_step_probe.py 4 > $O/overlap_step.txt 2>&1
cat $O/overlap_step.txt | tail -30
