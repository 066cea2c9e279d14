#!/bin/bash
# round-6 GPU call 1: whole GPU suite (new: side-stream replica without host sync, fast chain at all positions on the
# frame loop's path, merged pass at batch 12/16, native 16-row GEMV form), GEMV bench + bit check up to 32 rows,
# start-up ordering stress (ordered vs unordered), default bench
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
O=gpurun_out/r06c1; mkdir -p $O
GEMV_CHECK=1 timeout 300 tools/bin/gemv_bench > $O/gemv_bench.txt 2>&1
tail -70 $O/gemv_bench.txt
timeout 1500 python -m pytest tests -m gpu -q -x --durations=25 > $O/pytest.log 2>&1
tail -45 $O/pytest.log
timeout 600 python tools/startup_order_stress.py --mode ordered --world 4 --iters 10 > $O/stress_ordered_w4.txt 2>&1
grep -E "stress\]|launcher|fault|Error" $O/stress_ordered_w4.txt | grep -v "first step" | tail -15
timeout 600 python tools/startup_order_stress.py --mode unordered --world 4 --iters 10 > $O/stress_unordered_w4.txt 2>&1
grep -E "stress\]|launcher|fault|Error" $O/stress_unordered_w4.txt | grep -v "first step" | tail -15
timeout 600 python bench.py > $O/bench_default.log 2>&1
tail -3 $O/bench_default.log
