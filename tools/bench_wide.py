#!/usr/bin/env python3
"""Rows wider than any single-pass kernel: column-blocked streaming passes (A read twice) vs the two-pass kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos
from fastoptsolver_amd import _core, _lib
lib = _lib.load()
torch.cuda.set_device(0)
for m, n, dt in ((32768, 65536, torch.float32), (16384, 131072, torch.float32), (65536, 32768, torch.bfloat16), (32768, 40000, torch.float32)):
    A = torch.randn(m, n, device="cuda").to(dt)
    b = torch.randn(m, device="cuda")
    x = torch.randn(n, device="cuda"); g = torch.empty(n, device="cuda")
    byt = m * n * A.element_size() + 4 * m + 8 * n
    for label, kw in (("column blocks", {}), ("two-pass kernels", dict(no_colblock=True))):
        prob = fos.prepare(A, b, pad=False)
        if kw:
            prob.replan(**kw)
        for _ in range(2):
            lib.fos_gemv_pair(prob.h, _core.ptr(x), 0.0, _core.ptr(g), None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            lib.fos_gemv_pair(prob.h, _core.ptr(x), 0.0, _core.ptr(g), None)
        e1.record(); e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        print(f"{m}x{n} {str(dt)[6:]:9s} {label:18s} plan {prob.plan()['path']}/{prob.plan()['colblock']}: {us:9.1f} us per gradient = "
              f"{byt / us / 1e3 / 80:5.1f} % of 8 TB/s on the single-read bytes", flush=True)
        del prob
    del A, b
    torch.cuda.empty_cache()
