#!/usr/bin/env python3
"""The fp64-accumulating pass of the L-BFGS fg (fos_gemv_pair_dd) against the fp32 pass (fos_gemv_pair) on the same
matrix: HIP-event time of the A-pass kernel alone (fos_problem_profile), fraction of the 8 TB/s HBM roofline on the
algorithmic bytes B_fg = m*n*s_A + 4m + 8n, and the whole call (pass + slab reduction)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos  # noqa: E402
from fastoptsolver_amd import _core, _lib  # noqa: E402


def timed_us(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    torch.cuda.set_device(0)
    lib = _lib.load()
    rows = []
    shapes = [(65536, 8192, "f32"), (131072, 16384, "f32"), (131072, 16384, "bf16"), (262144, 4096, "f32"),
              (1048576, 1024, "f32"), (262144, 8192, "bf16")]
    if len(sys.argv) > 1:
        shapes = shapes[: int(sys.argv[1])]
    for m, n, kind in shapes:
        dt = torch.float32 if kind == "f32" else torch.bfloat16
        A = torch.randn(m, n, device="cuda", dtype=torch.float32).to(dt)
        b = torch.randn(m, device="cuda")
        prob = fos.prepare(A, b)
        x32 = torch.randn(n, device="cuda")
        x64 = x32.double()
        g32 = torch.empty(n, device="cuda")
        g64 = torch.empty(n + 1, dtype=torch.float64, device="cuda")
        s = 2 if kind == "bf16" else 4
        b_fg = m * n * s + 4 * m + 8 * n
        res = dict(m=m, n=n, dtype=kind, bytes_fg=b_fg)
        for name, fn in (("f32", lambda: lib.fos_gemv_pair(prob.h, _core.ptr(x32), 0.5, _core.ptr(g32), None)),
                         ("dd", lambda: lib.fos_gemv_pair_dd(prob.h, _core.ptr(x64), 0.5, _core.ptr(g64)))):
            call_us = timed_us(fn, 30)
            prob.profile(1)
            prob.profile_read()
            for _ in range(30):
                fn()
            ms, cnt = prob.profile_read()
            prob.profile(0)
            k_us = ms * 1e3 / cnt
            res[name] = dict(kernel_us=k_us, kernel_frac=b_fg / (k_us * 1e-6) / 8e12, call_us=call_us,
                             call_frac=b_fg / (call_us * 1e-6) / 8e12)
        res["dd_over_f32_kernel_rate"] = res["f32"]["kernel_us"] / res["dd"]["kernel_us"]
        print(json.dumps(res), flush=True)
        rows.append(res)
        del prob, A, b
        torch.cuda.empty_cache()
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench_dd.json")
    json.dump(rows, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
