#!/usr/bin/env python3
"""Idle gaps on the GPU timeline of a rocprofv3 --kernel-trace database: where no kernel runs for more than MIN_US.
usage: python tools/gpu_gaps.py results.db [min_us=200] [last_ms=0: only the last N ms of the trace]"""
import sqlite3
import sys


def main(path, min_us=200.0, last_ms=0.0):
    cur = sqlite3.connect(path).cursor()
    rows = list(cur.execute("select start, end, name from kernels order by start"))
    if last_ms > 0:
        t_end = max(r[1] for r in rows)
        rows = [r for r in rows if r[0] >= t_end - last_ms * 1e6]
    busy_end, gaps, busy = rows[0][1], [], 0
    prev = rows[0]
    for r in rows[1:]:
        if r[0] > busy_end:
            g = (r[0] - busy_end) / 1e3
            if g >= min_us:
                gaps.append((g, (busy_end - rows[0][0]) / 1e6, prev[2][:60], r[2][:60]))
        if r[1] > busy_end:
            busy_end, prev = r[1], r
    span = (max(r[1] for r in rows) - rows[0][0]) / 1e6
    ksum = sum(r[1] - r[0] for r in rows) / 1e6
    print(f"# {path}: {len(rows)} kernels over {span:.1f} ms, kernel time {ksum:.1f} ms; gaps >= {min_us:.0f} us: {len(gaps)}, "
          f"total {sum(g[0] for g in gaps) / 1e3:.1f} ms")
    for g in gaps:
        print(f"{g[0]:9.1f} us at +{g[1]:9.2f} ms   after {g[2]}   before {g[3]}")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 200.0, float(sys.argv[3]) if len(sys.argv) > 3 else 0.0)
