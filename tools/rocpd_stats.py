#!/usr/bin/env python3
"""Per-kernel summary (calls, average/min/max duration, VGPRs, LDS) of a rocprofv3 rocpd
database: python3 tools/rocpd_stats.py results.db [--csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), sum(d.end-d.start), "
                      f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(d.group_segment_size), max(d.private_segment_size), max(d.grid_size_x/d.workgroup_size_x) "
                      f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 6 desc").fetchall()
    total = sum(r[5] for r in rows)
    csv = "--csv" in sys.argv
    if csv:
        print("Name,Calls,TotalNs,AverageNs,Percentage,MinNs,MaxNs,VGPR,AGPR,LDS,Scratch,Blocks")
    for r in rows:
        name = r[0].split("(")[0].replace("void ", "")
        if csv:
            print(f"\"{name}\",{r[1]},{r[5]},{r[2]:.1f},{100*r[5]/total:.2f},{r[3]},{r[4]},{r[6]},{r[7]},{r[8]},{r[9]},{r[10]}")
        else:
            print(f"{name[:44]:44s} {r[1]:5d} avg {r[2]/1e3:9.1f} us  min {r[3]/1e3:9.1f}  max {r[4]/1e3:9.1f}  {100*r[5]/total:5.1f}%  vgpr {r[6]} agpr {r[7]} lds {r[8]} scratch {r[9]} blocks {r[10]}")


if __name__ == "__main__":
    main()
