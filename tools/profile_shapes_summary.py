#!/usr/bin/env python3
"""Cut the rocprofv3 traces of tools/profile_shapes.py into per-shape segments and write profiles/<tag>_shapes.md.

    python tools/profile_shapes_summary.py --tag r02 --plan gpurun_out/profile_shapes_plan.json \
        --kernel-trace DIR_KT [--fetch DIR_FETCH --write DIR_WRITE]

Only dispatches of the A-pass kernels are counted (the slab reductions and vector kernels in between are skipped); each
segment of the plan owns the next `launches` of them, the first two are dropped as warm-up.  HBM bytes follow
MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled on gfx950 for 16 B/lane streams."""
import argparse
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PASS_KERNELS = ("gemv_pair_kernel", "gemv_tall_kernel", "gemv_tall_quad_kernel", "gemv_tall_rows_kernel", "gemv_wide_kernel")


def find(d, pattern):
    hits = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
    if not hits:
        raise SystemExit(f"no {pattern} under {d}")
    return hits[-1]


def is_pass(name):
    return any(k in name for k in PASS_KERNELS)


def trace_rows(d):
    rows = list(csv.DictReader(open(find(d, "*kernel_trace.csv"))))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    return [r for r in rows if is_pass(r["Kernel_Name"])]


def counter_rows(d, counter):
    rows = [r for r in csv.DictReader(open(find(d, "*counter_collection.csv"))) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # one row per dispatch and (possibly) per dimension instance: sum per dispatch
    acc, order = {}, []
    for r in rows:
        if not is_pass(r["Kernel_Name"]):
            continue
        k = int(r["Dispatch_Id"])
        if k not in acc:
            acc[k] = 0.0
            order.append(k)
        acc[k] += float(r["Counter_Value"])
    return [acc[k] for k in order]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--plan", required=True)
    ap.add_argument("--kernel-trace", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    a = ap.parse_args()
    plan = json.load(open(a.plan))
    kt = trace_rows(a.kernel_trace)
    fetch = counter_rows(a.fetch, "FETCH_SIZE") if a.fetch else None
    write = counter_rows(a.write, "WRITE_SIZE") if a.write else None
    need = sum(p["launches"] for p in plan)
    assert len(kt) == need, f"trace holds {len(kt)} A-pass dispatches, plan expects {need}"
    lines = [f"# A-pass kernels by shape ({a.tag}): rocprofv3 --kernel-trace" + (" + --pmc FETCH_SIZE / WRITE_SIZE (separate passes)" if fetch else ""),
             "", "command: `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/profile_shapes.py`; "
             "algorithmic bytes per launch = m*n*s_A + 4m + 8n; peak 8000 GB/s; first two launches of a segment dropped.", "",
             "| shape / pass | kernel | VGPRs | LDS B | waves/grid | avg us | algorithmic GB/s | % of 8 TB/s | HBM bytes / algorithmic (PMC) |",
             "|---|---|---|---|---|---|---|---|---|"]
    pos = 0
    out = []
    for p in plan:
        seg = kt[pos:pos + p["launches"]][2:]
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg]
        avg = sum(durs) / len(durs)
        r0 = seg[0]
        name = r0["Kernel_Name"].split("(")[0].replace("void ", "").replace("fos::", "")
        ratio = ""
        if fetch and write and len(fetch) == need and len(write) == need:
            f = fetch[pos:pos + p["launches"]][2:]
            w = write[pos:pos + p["launches"]][2:]
            hbm = 2.0 * 1024.0 * sum(f) / len(f) + 1024.0 * sum(w) / len(w)
            ratio = f"{hbm / p['bytes']:.3f}"
        gbps = p["bytes"] / (avg * 1e-6) / 1e9
        grid = int(r0.get("Grid_Size", r0.get("Grid_Size_X", 0)) or 0)
        wg = int(r0.get("Workgroup_Size", r0.get("Workgroup_Size_X", 0)) or 0)
        lines.append(f"| {p['label']} | `{name[:70]}` | {r0.get('VGPR_Count', r0.get('Arch_VGPR_Count', '?'))} | "
                     f"{r0.get('LDS_Block_Size', '?')} | {grid // 64 if grid else '?'} ({grid // wg if wg else '?'} x {wg}) | "
                     f"{avg:.1f} | {gbps:.0f} | {gbps / 80:.1f} | {ratio} |")
        out.append(dict(label=p["label"], kernel=name, avg_us=avg, gbps=gbps, frac=gbps / 8000.0, traffic_ratio=ratio or None))
        pos += p["launches"]
    path = os.path.join(ROOT, "profiles", f"{a.tag}_shapes.md")
    open(path, "w").write("\n".join(lines) + "\n")
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{a.tag}_shapes.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
