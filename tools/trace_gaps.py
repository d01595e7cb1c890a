#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace (…_kernel_trace.csv): the last `count` kernels with their duration and the idle gap
since the previous kernel ended.   usage: trace_gaps.py DIR [count]"""
import csv, glob, os, sys
d = sys.argv[1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 80
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-count:]
prev = None
tot_k = tot_gap = 0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    name = r["Kernel_Name"].split("(")[0][:70]
    print(f"{gap:9.2f} us gap  {(e - s) / 1e3:9.2f} us  {name}")
    tot_k += (e - s) / 1e3
    tot_gap += gap
    prev = e
print(f"kernels {tot_k:.1f} us, gaps {tot_gap:.1f} us over {len(rows)} launches")
