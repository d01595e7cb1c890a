#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage output (stdin) as one line per kernel."""
import re, sys, subprocess
cur = {}
rows = []
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m:
        if "error" in line or "warning" in line:
            print(line.rstrip())
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        if cur: rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if cur: rows.append(cur)
for r in rows:
    try:
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    except Exception:
        name = r["name"]
    name = re.sub(r"\(.*", "", name)[:80]
    print(f'{name:80s} vgpr {r.get("VGPRs","?"):>4} agpr {r.get("AGPRs","?"):>3} sgpr {r.get("TotalSGPRs","?"):>4} '
          f'spill {r.get("VGPRs Spill","?"):>3} scratch {r.get("ScratchSize [bytes/lane]","?"):>4} occ {r.get("Occupancy [waves/SIMD]","?")} lds {r.get("LDS Size [bytes/block]","?")}')
