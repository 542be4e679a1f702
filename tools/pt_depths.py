#!/usr/bin/env python3
"""Launch-by-launch view of a rocprofv3 kernel trace (rocpd database): every dispatch in start order
with its duration and the gap to the previous one, restricted to the last `--last N` dispatches.

    python tools/pt_depths.py results.db [--last 40]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    last = int(sys.argv[sys.argv.index("--last") + 1]) if "--last" in sys.argv else 40
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
    rows = rows[-last:]
    prev_end = rows[0][1]
    t0 = rows[0][1]
    for name, st, en in rows:
        short = name.split("(")[0].replace("void ", "").replace("rtx::", "")
        print(f"{(st - t0) / 1e3:10.1f} us  +{(st - prev_end) / 1e3:7.1f} gap  {(en - st) / 1e3:8.1f} us  {short[:60]}")
        prev_end = max(prev_end, en)
    print(f"span {(max(r[2] for r in rows) - t0) / 1e3:.1f} us, busy {sum(r[2] - r[1] for r in rows) / 1e3:.1f} us")


if __name__ == "__main__":
    main()
