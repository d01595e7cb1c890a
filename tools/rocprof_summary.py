#!/usr/bin/env python3
"""Summarise rocprofv3 output directories into profiles/ (tracked).

    python tools/rocprof_summary.py --tag r01_cfg2 --workload cfg2 \
        --kernel-trace gpurun_out/prof_kt --fetch gpurun_out/prof_fetch --write gpurun_out/prof_write

Inputs are the directories written by
    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 bench.py ...
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -- python3 bench.py ...     (separate pass)
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d DIR -- python3 bench.py ...     (separate pass)
HBM traffic per launch follows MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports exactly half the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled;
WRITE_SIZE is exact for 16 B/lane streaming stores.
"""
import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOMINANT = "fos::gemv_pair_kernel"
DOMINANT_EXTRA = ""


def find(d, pattern):
    hits = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
    if not hits:
        raise SystemExit(f"no {pattern} under {d}")
    return hits[-1]


def counters(d):
    acc = collections.defaultdict(list)
    with open(find(d, "*counter_collection.csv")) as fh:
        for r in csv.DictReader(fh):
            acc[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    return acc


def bench_line(d):
    try:
        with open(os.path.join(d, "bench.json")) as fh:
            return json.loads(fh.read().strip().splitlines()[-1])
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--workload", required=True)
    ap.add_argument("--kernel-trace", required=True)
    ap.add_argument("--fetch", required=True)
    ap.add_argument("--write", required=True)
    ap.add_argument("--command", default="python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline")
    ap.add_argument("--match", default="", help="extra substring the dominant kernel's name must contain "
                                                 "(e.g. 'float, 512' to pick one geometry when a run holds several)")
    a = ap.parse_args()
    global DOMINANT_EXTRA
    DOMINANT_EXTRA = a.match
    out_dir = os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)

    rows = list(csv.DictReader(open(find(a.kernel_trace, "*kernel_stats.csv"))))
    lines = [f"# rocprofv3 --kernel-trace --stats  ({a.tag})", "",
             f"command: `rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- {a.command}`", "",
             "| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    dom = None
    for r in rows:
        name = r["Name"]
        if DOMINANT in name and DOMINANT_EXTRA in name and dom is None:
            dom = r
        short = name.split("(")[0].replace("void ", "")[:90]
        lines.append(f"| `{short}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | "
                     f"{float(r['AverageNs']) / 1e3:.2f} | {float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} | "
                     f"{r['Percentage']} |")
    bl = bench_line(a.kernel_trace)
    if bl:
        lines += ["", f"bench.py line of the same run: value {bl['value']:.1f} {bl['unit']}, ms_per_step "
                  f"{bl['ms_per_step']:.4f}, in-bench HIP-event average of the dominant kernel "
                  f"{bl['roofline']['kernel_avg_us']:.2f} us over {bl['roofline']['kernel_launches_timed']} launches "
                  f"(rocprof average above: {float(dom['AverageNs']) / 1e3:.2f} us over {dom['Calls']} calls, which also "
                  "include the power-iteration and A^T b launches of the setup)."]
    with open(os.path.join(out_dir, f"{a.tag}_kernel_stats.md"), "w") as fh:
        fh.write("\n".join(lines) + "\n")

    fetch, write = counters(a.fetch), counters(a.write)
    plines = [f"# rocprofv3 --pmc passes ({a.tag})", "",
              "separate passes: `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (TCC slots do not fit both)", "",
              "| kernel | launches | FETCH_SIZE KiB (raw avg) | fetched bytes (x2 gfx950 correction) | WRITE_SIZE KiB (avg) | written bytes |",
              "|---|---|---|---|---|---|"]
    traffic = None
    for (k, c), v in sorted(fetch.items()):
        if c != "FETCH_SIZE" or "fos::" not in k:
            continue
        w = write.get((k, "WRITE_SIZE"), [0.0])
        f_avg, w_avg = sum(v) / len(v), sum(w) / len(w)
        fb, wb = 2.0 * f_avg * 1024.0, w_avg * 1024.0
        short = k.split("(")[0].replace("void ", "")[:80]
        plines.append(f"| `{short}` | {len(v)} | {f_avg:.1f} | {fb:.4e} | {w_avg:.1f} | {wb:.4e} |")
        if DOMINANT in k and DOMINANT_EXTRA in k and traffic is None:
            traffic = dict(hbm_bytes_per_launch=fb + wb, fetch_bytes=fb, write_bytes=wb, fetch_size_kib_raw=f_avg,
                           write_size_kib_raw=w_avg, launches=len(v), kernel=short,
                           correction="FETCH_SIZE x2 (gfx950, 16 B/lane coalesced stream), WRITE_SIZE exact; KiB -> bytes")
    with open(os.path.join(out_dir, f"{a.tag}_pmc.md"), "w") as fh:
        fh.write("\n".join(plines) + "\n")

    tpath = os.path.join(out_dir, "pmc_traffic.json")
    try:
        allt = json.load(open(tpath))
    except Exception:
        allt = {}
    if traffic:
        traffic["source"] = f"profiles/{a.tag}_pmc.md"
        allt[a.workload] = traffic
        json.dump(allt, open(tpath, "w"), indent=1)
    print("\n".join(lines[:12]))
    print("\n".join(plines))
    print(json.dumps(traffic))


if __name__ == "__main__":
    main()
