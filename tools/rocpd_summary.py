#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (--kernel-trace --stats) as a per-kernel table.
usage: python tools/rocpd_summary.py gpurun_out/profN/xx_results.db [> profiles/rNN_kernels.txt]"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
        "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    print(f"# {path}: {len(rows)} kernels, total kernel time {tot/1e3:.2f} ms")
    print(f"{'pct':>6} {'calls':>8} {'total_ms':>10} {'avg_us':>10} {'min_us':>9} {'max_us':>10}  name")
    for r in rows:
        print(f"{r[2]/tot*100:6.2f} {r[1]:8d} {r[2]/1e3:10.2f} {r[3]:10.2f} {r[4]:9.2f} {r[5]:10.2f}  {r[0][:120]}")
    print("\n# skinny GEMV and conv kernels by grid (name, grid_wgs_x, grid_y, grid_z, calls, avg_us)")
    q = ("select name, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z, count(*), avg(end-start)/1e3 from kernels "
         "where name like '%linear_skinny%' or name like '%conv_mfma%' group by name, grid_x, grid_y, grid_z order by 6*5 desc limit 40")
    for r in cur.execute(q):
        print(f"{r[5]:10.2f} us x{r[4]:6d}  grid=({r[1]},{r[2]},{r[3]})  {r[0][:100]}")


if __name__ == "__main__":
    main(sys.argv[1])
