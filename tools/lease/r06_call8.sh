#!/bin/bash
# round-6 GPU call 8: k = 7 convs on small grids with two channel groups per step -- threshold sweep on the streamed chunks
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
O=gpurun_out/r06c8; mkdir -p $O
for thr in 0 300 700 1500 4000; do
  echo "== FMI_CONV_K7_SMALL=$thr"
  FMI_CONV_K7_SMALL=$thr timeout -k 10 200 python tools/first_chunk_probe.py 8 20 2>&1 | grep "first chunk"
  FMI_CONV_K7_SMALL=$thr STREAM_ONLY=cached timeout -k 10 300 python tools/stream_breakdown.py 2>&1 | grep -E "frames \[  8| 40, 72|offline decode"
done > $O/k7_sweep.txt 2>&1
cat $O/k7_sweep.txt
timeout -k 10 900 python -m pytest tests/test_dac_gpu.py tests/test_stream_gpu.py -m gpu -q -x > $O/pytest.log 2>&1
tail -3 $O/pytest.log
