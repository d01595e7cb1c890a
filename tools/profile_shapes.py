#!/usr/bin/env python3
"""One process, many shapes: every kernel family of the A pass (streaming fp32 / bf16 geometries, one-wave-per-row,
row-per-thread, row-per-quad, y-in-LDS wide rows, the fp64-accumulating forms) launched LAUNCHES times each, in a fixed
order, so that a rocprofv3 trace of this command can be cut into per-shape segments by tools/profile_shapes_summary.py.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/profile_shapes.py [--only SUBSTR]
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR -- python3 tools/profile_shapes.py      (separate passes)
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d DIR -- python3 tools/profile_shapes.py

Writes gpurun_out/profile_shapes_plan.json: the order of the segments with the algorithmic bytes of one launch
(m*n*s_A + 4m + 8n for a gradient pass: A once, b, y in, gradient out)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos  # noqa: E402
from fastoptsolver_amd import _core, _lib  # noqa: E402

LAUNCHES = 12
SHAPES = [
    # label, m, n, dtype, passes
    ("cfg2 65536x8192 f32", 65536, 8192, "f32", ("f32", "dd")),
    ("shard 131072x16384 f32", 131072, 16384, "f32", ("f32", "dd")),
    ("shard 131072x16384 bf16", 131072, 16384, "bf16", ("f32", "dd")),
    ("wide 65536x32768 f32 (y in LDS)", 65536, 32768, "f32", ("f32",)),
    ("52224x10240 f32 (five chunks per thread)", 52224, 10240, "f32", ("f32", "dd")),
    ("43520x12288 f32 (three chunks per thread, 1024 threads)", 43520, 12288, "f32", ("f32",)),
    ("104704x5120 f32 (five chunks per thread)", 104704, 5120, "f32", ("f32",)),
    ("43520x24576 bf16 (six chunks per thread)", 43520, 24576, "bf16", ("f32",)),
    ("32768x32768 bf16 (y in LDS)", 32768, 32768, "bf16", ("f32",)),
    ("narrow 1000000x128 f32 (chunk per lane, 32 lanes per row)", 1000000, 128, "f32", ("f32",)),
    ("narrow 2000000x96 f32 (chunk per lane, 32 lanes per row)", 2000000, 96, "f32", ("f32",)),
    ("narrow 500000x512 f32 (one wave per row)", 500000, 512, "f32", ("f32",)),
    ("1048576x1024 f32", 1048576, 1024, "f32", ("f32", "dd")),
    ("tall 2000000x64 f32 (chunk per lane, 16 lanes per row)", 2000000, 64, "f32", ("f32",)),
    ("tall 4000000x16 f32 (row per thread)", 4000000, 16, "f32", ("f32",)),
    ("tall 4000000x32 f32 (chunk per lane, 8 lanes per row)", 4000000, 32, "f32", ("f32",)),
    ("tall 8000000x5 f32 (LDS-staged rows)", 8000000, 5, "f32", ("f32",)),
    ("tall 32000000x5 f32 (LDS-staged rows)", 32000000, 5, "f32", ("f32",)),
    ("262144x8192 bf16", 262144, 8192, "bf16", ("f32", "dd")),
]


def main():
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else ""
    torch.cuda.set_device(0)
    lib = _lib.load()
    plan = []
    for label, m, n, kind, passes in SHAPES:
        if only and only not in label:
            continue
        dt = torch.float32 if kind == "f32" else torch.bfloat16
        A = torch.empty(m, n, device="cuda", dtype=dt)
        for r0 in range(0, m, 1 << 18):
            A[r0:r0 + (1 << 18)] = torch.randn(min(1 << 18, m - r0), n, device="cuda").to(dt)
        b = torch.randn(m, device="cuda")
        prob = fos.prepare(A, b)
        x32 = torch.randn(prob.n_dev, device="cuda")
        x64 = x32.double()
        g32 = torch.empty(prob.n_dev, device="cuda")
        g64 = torch.empty(prob.n_dev + 1, dtype=torch.float64, device="cuda")
        s = 2 if kind == "bf16" else 4
        for ps in passes:
            for _ in range(LAUNCHES):
                if ps == "f32":
                    _lib.check(lib.fos_gemv_pair(prob.h, _core.ptr(x32), 0.0, _core.ptr(g32), None))
                else:
                    _lib.check(lib.fos_gemv_pair_dd(prob.h, _core.ptr(x64), 0.0, _core.ptr(g64)))
            torch.cuda.synchronize()
            plan.append(dict(label=f"{label} [{'fp32 pass' if ps == 'f32' else 'fp64-accumulating pass'}]", m=m, n=n,
                             dtype=kind, launches=LAUNCHES, bytes=m * n * s + 4 * m + 8 * n, plan=prob.plan()))
        del prob, A, b
        torch.cuda.empty_cache()
        time.sleep(2.0)          # returning GiBs to the driver costs the next kernels 3-6 % for seconds (profiles/r03_row_order.md)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "profile_shapes_plan.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(plan, open(out, "w"), indent=1)
    print(f"{len(plan)} segments x {LAUNCHES} launches")


if __name__ == "__main__":
    main()
