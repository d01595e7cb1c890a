#!/usr/bin/env python3
"""Workgroup-count sweep of the narrow / tall A-pass kernels (fos_problem_tune): HIP-event time per launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos  # noqa: E402
from fastoptsolver_amd import _core, _lib  # noqa: E402

lib = _lib.load()
torch.cuda.set_device(0)
CASES = [
    (1048576, 1024, [(256, 1, 4, w) for w in (256, 512, 1024, 2048)] + [(256, 2, 4, w) for w in (256, 512, 1024)]),
    (1000000, 128, [(64, 1, 4, w) for w in (2048, 4096, 8192, 16384)]),
    (500000, 512, [(64, 2, 4, w) for w in (2048, 4096, 8192)] + [(256, 1, 4, w) for w in (256, 512, 1024, 2048)]),
    (262144, 4096, [(256, 4, 2, w) for w in (256, 512, 768)] + [(512, 4, 1, w) for w in (256, 512)]),
    (2000000, 64, [(0, 0, 0, w) for w in (1024, 2048, 4096, 8192)]),
    (4000000, 32, [(0, 0, 0, w) for w in (1024, 2048, 4096, 8192)]),
    (4000000, 16, [(0, 0, 0, w) for w in (1024, 2048, 4096, 8192)]),
    (8000000, 5, [(0, 0, 0, w) for w in (1024, 2048, 4096, 8192)]),
]
for m, n, geos in CASES:
    A = torch.randn(m, n, device="cuda")
    b = torch.randn(m, device="cuda")
    prob = fos.prepare(A, b)
    x = torch.randn(n, device="cuda")
    g = torch.empty(n, device="cuda")
    byt = m * n * 4 + 4 * m + 8 * n
    for th, k, r, w in geos:
        try:
            prob.tune(th, k, r, w)
        except Exception as exc:
            print(f"{m}x{n} {th}x{k}x{r} wg {w}: {exc}")
            continue
        for _ in range(3):
            lib.fos_gemv_pair(prob.h, _core.ptr(x), 0.0, _core.ptr(g), None)
        prob.profile(1)
        prob.profile_read()
        for _ in range(20):
            lib.fos_gemv_pair(prob.h, _core.ptr(x), 0.0, _core.ptr(g), None)
        ms, cnt = prob.profile_read()
        prob.profile(0)
        us = ms * 1e3 / cnt
        print(f"{m}x{n} {th}x{k}x{r} wg {prob.plan()['workgroups']:5d}: {us:8.1f} us  {byt / us / 1e3 / 80:5.1f} % of 8 TB/s", flush=True)
    del prob, A, b
    torch.cuda.empty_cache()
