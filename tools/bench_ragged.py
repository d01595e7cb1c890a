#!/usr/bin/env python3
"""Ragged tall-skinny pass (the reference's own n = 5): kernel time against m.  A 35 us kernel is mostly fixed cost (launch,
first block's latency, the closing block reduction); t(m) = t0 + bytes / BW separates that from the streaming rate."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastoptsolver_amd as fos  # noqa: E402
from fastoptsolver_amd import _core, _lib  # noqa: E402

lib = _lib.load()
torch.cuda.set_device(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rows = []
for m in (1_000_000, 2_000_000, 4_000_000, 8_000_000, 16_000_000, 32_000_000, 64_000_000, 128_000_000):
    A = torch.randn(m, n, device="cuda")
    b = torch.randn(m, device="cuda")
    prob = fos.prepare(A, b)
    x = torch.randn(n, device="cuda")
    g = torch.empty(n, device="cuda")
    byt = m * n * 4 + 4 * m + 8 * n
    best = None
    for wg in (0, 512, 1024, 2048):
        if wg:
            prob.tune(0, 0, 0, wg)
        for _ in range(3):
            lib.fos_gemv_pair(prob.h, _core.ptr(x), 0.0, _core.ptr(g), None)
        prob.profile(1)
        prob.profile_read()
        for _ in range(20):
            lib.fos_gemv_pair(prob.h, _core.ptr(x), 0.0, _core.ptr(g), None)
        ms, cnt = prob.profile_read()
        prob.profile(0)
        us = ms * 1e3 / cnt
        print(f"{m}x{n} wg {prob.plan()['workgroups']:5d}: {us:8.1f} us  {byt / us / 1e3 / 80:5.1f} % of 8 TB/s", flush=True)
        if wg == 0:
            best = us
    rows.append((byt, best))
    del prob, A, b
    torch.cuda.empty_cache()
B = np.array([r[0] for r in rows], dtype=np.float64)
T = np.array([r[1] for r in rows], dtype=np.float64) * 1e-6
slope, t0 = np.polyfit(B, T, 1)
print(f"fit over the planner's default: t = {t0 * 1e6:.1f} us + bytes / {1 / slope / 1e12:.2f} TB/s  ({1 / slope / 8e12 * 100:.1f} % of 8 TB/s asymptotically)")
