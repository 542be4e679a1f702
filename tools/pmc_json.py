#!/usr/bin/env python3
"""Summary of the rocprofv3 passes of tools/pmc_collect.sh as JSON: per kernel the mean counter values per
dispatch, the mean duration from the kernel trace, and the figures derived from them.

    python tools/pmc_json.py NAME [--skip N]     reads gpurun_out/pmc_NAME_*/, prints one JSON object

--skip N drops the first N dispatches of every kernel (warm-up launches).

Derived (MI355X: 256 CUs, 1024 SIMDs, 8 XCDs; GRBM_GUI_ACTIVE is summed over the XCDs):
  cycles            = GRBM_GUI_ACTIVE / 8
  valu_busy         = 4 * SQ_ACTIVE_INST_VALU / 1024 / cycles      (a wave64 VALU instruction occupies its SIMD for 4 cycles)
  live_lane_frac    = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU)
  l1_lookups_per_clk_per_cu = TCP_TOTAL_CACHE_ACCESSES_sum / cycles / 256
  l2_hit            = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
  ta_busy           = GRBM_TA_BUSY / GRBM_GUI_ACTIVE
  hbm_bytes         = (2 * FETCH_SIZE + WRITE_SIZE) * 1024       (FETCH_SIZE counts half the bytes of 16-byte-per-lane reads on gfx950)
"""
import collections
import csv
import glob
import json
import os
import sys


def short(name):
    return name.split("(")[0].replace("void ", "").replace("rtx::", "")


def main():
    name = sys.argv[1]
    skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(f"gpurun_out/pmc_{name}_*")):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = collections.defaultdict(lambda: collections.defaultdict(list))
            for row in csv.DictReader(open(f)):
                per[short(row["Kernel_Name"])][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
            for k, cs in per.items():
                for c, v in cs.items():
                    v.sort()
                    acc[k][c] += [x for _, x in v[skip:]]
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            per = collections.defaultdict(list)
            for row in csv.DictReader(open(f)):
                per[short(row["Kernel_Name"])].append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
            for k, v in per.items():
                v.sort()
                acc[k]["duration_ns"] += [x for _, x in v[skip:]]
    out = {}
    for k, cs in acc.items():
        m = {c: sum(v) / len(v) for c, v in cs.items() if v}
        m["dispatches"] = max(len(v) for v in cs.values())
        g = m.get("GRBM_GUI_ACTIVE")
        if g:
            cyc = g / 8.0
            if "SQ_ACTIVE_INST_VALU" in m:
                m["valu_busy"] = round(4 * m["SQ_ACTIVE_INST_VALU"] / 1024 / cyc, 4)
            if "TCP_TOTAL_CACHE_ACCESSES_sum" in m:
                m["l1_lookups_per_clk_per_cu"] = round(m["TCP_TOTAL_CACHE_ACCESSES_sum"] / cyc / 256, 4)
            if "GRBM_TA_BUSY" in m:
                m["ta_busy"] = round(m["GRBM_TA_BUSY"] / g, 4)
        if m.get("SQ_ACTIVE_INST_VALU"):
            m["live_lane_frac"] = round(m.get("SQ_THREAD_CYCLES_VALU", 0) / (64 * m["SQ_ACTIVE_INST_VALU"]), 4)
        if "TCC_HIT_sum" in m and m["TCC_HIT_sum"] + m.get("TCC_MISS_sum", 0) > 0:
            m["l2_hit"] = round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 4)
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            m["hbm_bytes_per_launch"] = int((2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024)
        out[k] = m
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
