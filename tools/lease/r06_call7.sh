#!/bin/bash
# round-6 GPU call 7: kernel trace of the first streamed chunk; LAST the 'local' bisect of the multi-rank fault
set -u
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/../..}"
export TMPDIR=/tmp
O=gpurun_out/r06c7; mkdir -p $O
R=$PWD
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_fc -o f -- python $R/tools/first_chunk_probe.py 8 20 ) > $O/first_chunk_run.log 2>&1
grep "first chunk" $O/first_chunk_run.log
db=$(find /tmp/prof_fc -name '*_results.db' | head -1)
python - "$db" > $O/first_chunk_kernels.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='view' or type='table'")]
t = "kernels" if "kernels" in tabs else [x for x in tabs if "kernel" in x.lower()][0]
rows = list(cur.execute(f"select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3 from {t} group by name order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"total kernel time {tot/1e3:.2f} ms over the run; per call of 22 streams: {tot/22/1e3:.3f} ms")
for name, n, s, a, m in rows[:40]:
    print(f"{s/22:9.1f} us/stream  x{n/22:6.1f}  avg {a:8.2f} min {m:8.2f}  {name[:110]}")
PY
head -45 $O/first_chunk_kernels.txt
run() { name=$1; shift; timeout -k 10 460 python tools/startup_order_stress.py --world 4 --iters 6 --timeout 400 "$@" > $O/stress_$name.txt 2>&1
        echo "== $name: $(grep -c 'tokens equal' $O/stress_$name.txt) clean iterations; $(grep -c 'Memory access fault' $O/stress_$name.txt) faults; $(grep 'launcher' $O/stress_$name.txt)"; grep "Memory access fault" $O/stress_$name.txt | head -3; }
run local_a --mode local
run local_b --mode local
