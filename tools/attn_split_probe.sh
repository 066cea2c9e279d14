cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for SP in 1 2 4; do for T in 300 1024 2048; do
  W=/tmp/as_${SP}_$T; rm -rf $W
  ( cd /tmp && FMI_ATTN_SPLIT=$SP rocprofv3 --kernel-trace --stats -d $W -o a -- python $GRAFT_REPO_ROOT/tools/attn_decode_probe.py $T ) > /tmp/as.log 2>&1
  db=$(find $W -name '*_results.db' | head -1)
  echo "split=$SP $(grep '^T=' /tmp/as.log)"
  python - "$db" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
for name, n, avg, mn in cur.execute("select name, count(*), avg(end-start)/1e3, min(end-start)/1e3 from kernels where name like '%attn_decode_fused%' group by name"):
    print(f"   {avg:8.2f} us avg ({mn:.2f} min) x{n}  {name[:70]}")
PY
done; done
