#!/bin/bash
# round-6 GPU call 10: validation of the final tree -- whole GPU suite, smoke, default bench, remaining profile sections,
# bench.py --gpus 2 / 4 / 8 in the one-GPU debug mode
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
O=gpurun_out/r06c10; mkdir -p $O
timeout -k 10 1500 python -m pytest tests -m gpu -q --durations=15 > $O/pytest.log 2>&1
tail -22 $O/pytest.log
timeout -k 10 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout -k 10 900 python bench.py > $O/bench_default.log 2>&1
tail -1 $O/bench_default.log | cut -c1-1200
timeout -k 10 1200 bash tools/make_profiles.sh r06 codec stream attn > $O/make_profiles.log 2>&1
tail -3 $O/make_profiles.log
for n in 2 4 8; do
  FMI_BENCH_ONE_GPU=1 timeout -k 10 600 python bench.py --gpus $n --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_onegpu_n$n.log 2>&1
  echo "== one-GPU debug mode, --gpus $n: rc $?; $(grep -c 'Memory access fault' $O/bench_onegpu_n$n.log) faults; $(grep -c '^{\"metric\"' $O/bench_onegpu_n$n.log) JSON line(s)"
  grep "checksums" $O/bench_onegpu_n$n.log | head -1
done
