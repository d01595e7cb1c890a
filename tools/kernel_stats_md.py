#!/usr/bin/env python3
"""Markdown table of a `rocprofv3 --kernel-trace --stats --output-format csv` run (…_kernel_stats.csv).
usage: kernel_stats_md.py DIR "title" "command" > profiles/NAME.md"""
import csv, glob, os, sys
d, title, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
print(f"# {title}\n\ncommand: `{cmd}`\n")
print("| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    name = r["Name"].split("(")[0][:90]
    print(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | "
          f"{float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |")
