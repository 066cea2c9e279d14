#!/bin/bash
# round-6 GPU call 2: GEMV forms (row sets restored, 32-row form), engine primitives + layer skeleton, the start-up
# memory fault under rocgdb, Dual-AR GPU tests, bench
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
O=gpurun_out/r06c2; mkdir -p $O
GEMV_CHECK=1 timeout 300 tools/bin/gemv_bench > $O/gemv_bench.txt 2>&1
grep -E "^w|^head|M=(16|24|32) shipped|NO" $O/gemv_bench.txt
timeout 300 tools/bin/engine_probe 200 > $O/engine_probe.txt 2>&1
cat $O/engine_probe.txt
timeout 900 python tools/startup_order_stress.py --mode ordered --world 4 --iters 4 --gdb --timeout 800 > $O/stress_gdb.txt 2>&1
grep -E "stress\] mode|launcher|fault|SIGSEGV|received signal|in .*kernel|#[0-9]+ |Switching|AMDGPU" $O/stress_gdb.txt | head -60
timeout 600 python tools/startup_order_stress.py --mode ordered --world 4 --iters 6 --sync-after-setup > $O/stress_sync_after_setup.txt 2>&1
grep -E "stress\] mode|launcher|fault" $O/stress_sync_after_setup.txt | tail -12
timeout 900 python -m pytest tests/test_s2_parity_gpu.py tests/test_dualar_gpu.py -m gpu -q -x > $O/pytest_dualar.log 2>&1
tail -5 $O/pytest_dualar.log
timeout 600 python bench.py --no-cpu-baseline > $O/bench.log 2>&1
tail -1 $O/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['breakdown_ms'], d['other_configs']['batch16'], d['other_configs']['config3_mixed_lengths_queue16']['audio_sec_per_s'])"
