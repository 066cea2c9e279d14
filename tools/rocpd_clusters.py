#!/usr/bin/env python3
"""Per-kernel summary of ONE burst of a rocprofv3 --kernel-trace database: kernels are split into bursts wherever the
GPU was idle for more than GAP_MS; prints burst WHICH (negative: from the end).
usage: python tools/rocpd_clusters.py results.db [gap_ms=20] [which=-1]"""
import collections
import sqlite3
import sys


def main(path, gap_ms=20.0, which=-1):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select start, end, name, grid_x/workgroup_x, grid_y/workgroup_y, grid_z/workgroup_z from kernels order by start"))
    bursts, cur_b, last_end = [], [], None
    for r in rows:
        if last_end is not None and r[0] - last_end > gap_ms * 1e6:
            bursts.append(cur_b)
            cur_b = []
        cur_b.append(r)
        last_end = max(last_end or 0, r[1])
    bursts.append(cur_b)
    print(f"# {path}: {len(rows)} kernels in {len(bursts)} bursts (gap > {gap_ms} ms): " +
          ", ".join(f"{len(b)} k / {(b[-1][1] - b[0][0]) / 1e6:.1f} ms" for b in bursts[-8:]))
    b = bursts[which]
    agg = collections.OrderedDict()
    for r in b:
        key = (r[2][:90], r[3], r[4], r[5])
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += (r[1] - r[0]) / 1e3
    tot = sum(a[1] for a in agg.values())
    print(f"# burst {which}: {len(b)} kernels, kernel time {tot / 1e3:.2f} ms, span {(b[-1][1] - b[0][0]) / 1e6:.2f} ms")
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print(f"{a[1] / tot * 100:6.2f}%  {a[0]:4d} x {a[1] / a[0]:9.2f} us = {a[1] / 1e3:7.3f} ms  grid=({key[1]},{key[2]},{key[3]})  {key[0]}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 20.0, int(sys.argv[3]) if len(sys.argv) > 3 else -1)
