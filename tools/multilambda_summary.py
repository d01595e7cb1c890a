#!/usr/bin/env python3
"""profiles/<tag>_multilambda.md from the rocprofv3 passes of tools/bench_path.py (kernel-trace --stats, --pmc
SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, --pmc FETCH_SIZE) and its own JSON."""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, kt, mf, fe, bj = sys.argv[1:6]
def find(d, pat): return sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))[-1]
stats = {r["Name"]: r for r in csv.DictReader(open(find(kt, "*kernel_stats.csv")))}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in (mf, fe):
    for r in csv.DictReader(open(find(d, "*counter_collection.csv"))):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = [f"# Multi-lambda pass on the matrix cores ({tag})", "",
         "`rocprofv3 --kernel-trace --stats`, `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`, `--pmc FETCH_SIZE` (separate passes) of "
         "`python3 tools/bench_path.py` (cfg2 65536 x 8192 fp32, then the config-5 shard 131072 x 16384 bf16; 1, 2, 4, 8, 16 weights).",
         "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs); fetched bytes = FETCH_SIZE KiB x 1024 x 2 (gfx950).", "",
         "| kernel | calls | avg us | MFMA busy cycles / launch | MfmaUtil | fetched GB / launch | fetched GB/s |", "|---|---|---|---|---|---|---|"]
for name, r in stats.items():
    if "mfma" not in name and "gemv_multi" not in name and "gemv_pair_kernel" not in name:
        continue
    c = acc.get(name, {})
    busy = sum(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / max(len(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), 1)
    gui = sum(c.get("GRBM_GUI_ACTIVE", [0])) / max(len(c.get("GRBM_GUI_ACTIVE", [1])), 1)
    fetch = 2048.0 * sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1)
    avg = float(r["AverageNs"]) / 1e3
    util = busy / (gui / 8 * 1024) if gui else 0.0
    short = name.split("(")[0].replace("void ", "").replace("fos::", "")[:72]
    lines.append(f"| `{short}` | {r['Calls']} | {avg:.1f} | {busy:.4g} | {100 * util:.1f} % | {fetch / 1e9:.3f} | {fetch / avg / 1e3:.0f} |")
b = json.load(open(bj))
lines += ["", "HIP-event timings of the same script (no profiler): microseconds per iteration for all weights, and per weight:", "",
          "| problem | weights | us / iteration | us / weight-iteration | speed-up over one by one |", "|---|---|---|---|---|"]
for prob, res in b.items():
    one = res["1"]["us_per_iteration"]
    for nl, v in res.items():
        lines.append(f"| {prob} | {nl} | {v['us_per_iteration']:.1f} | {v['us_per_lambda_iteration']:.1f} | {one / v['us_per_lambda_iteration']:.2f}x |")
open(os.path.join(ROOT, "profiles", f"{tag}_multilambda.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
