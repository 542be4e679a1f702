#!/usr/bin/env python3
"""GPU busy time (union of kernel intervals) against the span they cover, from a rocprofv3 rocpd
database: python3 tools/gpu_busy.py results.db [skip_fraction]. Shows how much of a run the GPU
waited for the host or for dependencies between launches."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    rows = db.execute(f"select start, end from {kd} order by start").fetchall()
    rows = rows[int(len(rows) * skip):]
    busy, (cs, ce) = 0, rows[0]
    for s, e in rows[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    span = rows[-1][1] - rows[0][0]
    print(f"dispatches {len(rows)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms  ({100.0 * busy / span:.1f} %)")


if __name__ == "__main__":
    main()
