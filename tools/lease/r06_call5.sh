#!/bin/bash
# round-6 GPU call 5: profiles of the final Dual-AR binary (kernel traces, PMC traffic, GEMM bench), chunked codec decode
# kernel trace, and LAST the independent-processes fault experiment
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
O=gpurun_out/r06c5; mkdir -p $O
timeout -k 10 1500 bash tools/make_profiles.sh r06 step prefill pmc gemm > $O/make_profiles.log 2>&1
tail -3 $O/make_profiles.log
( cd /tmp && STREAM_ONLY=cached timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_stream -o s -- python $OLDPWD/tools/stream_breakdown.py ) > $O/stream_cached_run.log 2>&1
db=$(find /tmp/prof_stream -name '*_results.db' | head -1)
python tools/rocpd_summary.py "$db" > $O/stream_cached_kernels.txt 2>&1
head -40 $O/stream_cached_kernels.txt
bash tools/multiproc_independent.sh 4 3 $O/multiproc > $O/multiproc.txt 2>&1
cat $O/multiproc.txt
