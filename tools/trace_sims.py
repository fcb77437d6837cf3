#!/usr/bin/env python3
"""Print every similarity / refresh launch of a rocprofv3 kernel trace (csv) with its start time and duration, and the idle
gaps longer than 0.2 ms: shows how a stage's rate depends on when in a burst it runs.  usage: trace_sims.py <trace dir>"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"]
    if prev is not None and s - prev > 200e3:
        print(f"   -- idle {(s - prev) / 1e6:.2f} ms")
    if "sim_kernel" in n:
        wg = int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"])
        print(f"t={(s - t0) / 1e6:8.2f} ms  {wg:6d} wg  {(e - s) / 1e3:8.1f} us  {1e3 * (e - s) / 1e3 / wg:7.2f} ns/wg")
    prev = max(prev or e, e)
