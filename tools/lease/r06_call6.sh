#!/bin/bash
# round-6 GPU call 6: streaming decoder with kept left context -- codec / stream tests, chunk timings, bench
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
O=gpurun_out/r06c6; mkdir -p $O
timeout -k 10 1200 python -m pytest tests/test_dac_gpu.py tests/test_stream_gpu.py tests/test_rank_gpu.py tests/test_s2_parity_gpu.py tests/test_cli_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
tail -15 $O/pytest.log
timeout -k 10 400 python tools/stream_breakdown.py > $O/stream_breakdown.txt 2>&1
cat $O/stream_breakdown.txt
FMI_DAC_NO_STREAM_HALO=1 STREAM_ONLY=cached timeout -k 10 400 python tools/stream_breakdown.py > $O/stream_breakdown_ctx_recompute.txt 2>&1
cat $O/stream_breakdown_ctx_recompute.txt
timeout -k 10 600 python bench.py --no-cpu-baseline > $O/bench.log 2>&1
tail -1 $O/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['breakdown_ms'], d['roofline_prefill']['avg_launch_ms'], d['other_configs']['config4_streaming'], d['other_configs']['config4_streaming_staggered_arrivals'])"
