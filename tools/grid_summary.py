#!/usr/bin/env python3
"""Condense the JSON files written by `python -m fastoptsolver_amd.benchmark` into one table:
per scenario and curve - iterations run, first k with suboptimality <= 1e-5, final suboptimality, seconds."""
import glob, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fastoptsolver_amd import benchmark as bm

src, dst = sys.argv[1], sys.argv[2]
rows = []
for path in sorted(glob.glob(os.path.join(src, "benchmark_*.json"))):
    res = json.load(open(path))
    for panel in bm.PANELS:
        for label in res["curves"][panel]:
            sub = bm.suboptimality(res, panel, label)
            hit = np.nonzero(sub <= 1e-5)[0]
            rows.append(dict(scenario=res["name"], panel=panel, curve=label, iters=int(len(sub)),
                             k_1e5=int(hit[0]) + 1 if len(hit) else None, first=float(sub[0]) if len(sub) else None,
                             last=float(sub[-1]) if len(sub) else None,
                             seconds=round(res["seconds"].get(f"{panel}/{label}", 0.0), 4)))
by_panel = {}
for r in rows:
    by_panel.setdefault(r["panel"], []).append(r)
summary = {p: dict(curves=len(v), median_iters=float(np.median([r["iters"] for r in v])),
                   median_k_1e5=float(np.median([r["k_1e5"] for r in v if r["k_1e5"]])) if any(r["k_1e5"] for r in v) else None,
                   reached_1e5=sum(1 for r in v if r["k_1e5"]), total_seconds=round(sum(r["seconds"] for r in v), 2))
           for p, v in by_panel.items()}
json.dump(dict(summary=summary, rows=rows), open(dst, "w"), indent=0)
print(json.dumps(summary, indent=1, ensure_ascii=False))
